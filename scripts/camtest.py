"""Camera rays (mode 0, 4 Mi) of a Data/ config rendered at 1024^2: reference-tree route against fast-tree route
(MIRT_FAST_ALL_MODES=0/1). usage: python3 scripts/camtest.py Config_Cube128.xml"""
import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import renderer_amd as ra
cfg = sys.argv[1]
scene = ra.HostScene.load_config(os.path.join("/root/repo/Data", cfg))
cam0 = scene.camera()
cam = ra.make_camera(1024, 1024, int(cam0.sight_dist * 1024 // cam0.width), tuple(cam0.position), tuple(cam0.focus), tuple(cam0.up))
dev = ra.Device(device=0); dev.upload(scene.describe())
n = 1024 * 1024 * 4
po = torch.empty((3, n), dtype=torch.float32, device="cuda"); pd = torch.empty_like(po)
dev.generate_rays_dev(cam, 0, 4, 1, po.data_ptr(), pd.data_ptr())
t1 = torch.empty(n, dtype=torch.float32, device="cuda"); i1 = torch.empty(n, dtype=torch.int32, device="cuda")
ts = torch.cuda.Stream(); st = C.c_void_p(ts.cuda_stream); torch.cuda.synchronize()
reps = 10
ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
dev.intersect_dev(po.data_ptr(), pd.data_ptr(), None, n, 0, t1.data_ptr(), i1.data_ptr(), st)
for k in range(reps):
    ev[k].record(ts)
    dev.intersect_dev(po.data_ptr(), pd.data_ptr(), None, n, 0, t1.data_ptr(), i1.data_ptr(), st)
ev[reps].record(ts); torch.cuda.synchronize()
ms = np.median([ev[k].elapsed_time(ev[k + 1]) for k in range(reps)])
print(cfg, os.environ.get("MIRT_FAST_ALL_MODES", "tuned"), "tris", scene.n_tris, "%.3f ms" % ms, "hit", float((i1 >= 0).float().mean()))
