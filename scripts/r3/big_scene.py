"""One-off size check on the GPU box (not part of the test suite): the procedural mesh scene at a subdivision level
above the bench's (level 8: 3.9 M triangles, level 9: 15.7 M, just under the 2^24-slot limit of a leaf link), uploaded,
its captured bounce batch and a camera batch compared ray by ray with the CPU oracle (tests/test_gpu_fullbatch.py's
bar) and timed.
usage: python scripts/r3/big_scene.py LEVEL [rays_log2]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import renderer_amd as ra, renderer_amd.meshgen as meshgen, orc
from test_gpu_fullbatch import compare_full

level = int(sys.argv[1]); n_cap = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
t0 = time.time(); scene = meshgen.mesh_scene(level); t1 = time.time()
print("level %d: %d triangles, host scene + reference tree %.1f s" % (level, scene.n_tris, t1 - t0), flush=True)
dev = ra.Device(device=0); dev.upload(scene.describe()); t2 = time.time()
print("upload (flatten + fast tree + copies) %.1f s, device memory in use %.2f GB" % (
    t2 - t1, (torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 1e9), flush=True)
oracle = orc.OracleScene(scene.raw()); raw = scene.raw()
print("oracle scene %.1f s" % (time.time() - t2), flush=True)
cam = ra.make_camera(1024, 1024, 700)
so = torch.empty((3, n_cap), dtype=torch.float32, device="cuda"); sd = torch.empty_like(so)
sa = torch.empty(n_cap, dtype=torch.int32, device="cuda"); torch.cuda.synchronize()
n_sec = dev.capture_rays_dev(cam, 4, 1, 3, so.data_ptr(), sd.data_ptr(), sa.data_ptr(), n_cap, spp_begin=0, seed=1)
assert n_sec == n_cap
compare_full(dev, oracle, raw, so, sd, sa, n_sec, ra.MODE_NO_SELF, "bounce rays, mode 2", torch, ra)
po = torch.empty((3, 1 << 20), dtype=torch.float32, device="cuda"); pd = torch.empty_like(po)
dev.generate_rays_dev(cam, 0, 1, 1, po.data_ptr(), pd.data_ptr())
compare_full(dev, oracle, raw, po, pd, None, 1 << 20, ra.MODE_UNCONDITIONAL, "camera rays, mode 0", torch, ra)
# timing: a 4 Mi captured batch, median of 5
n_big = 4 << 20
bo = torch.empty((3, n_big), dtype=torch.float32, device="cuda"); bd = torch.empty_like(bo)
ba = torch.empty(n_big, dtype=torch.int32, device="cuda")
t = torch.empty(n_big, dtype=torch.float32, device="cuda"); i = torch.empty(n_big, dtype=torch.int32, device="cuda")
n_big = dev.capture_rays_dev(cam, 4, 1, 3, bo.data_ptr(), bd.data_ptr(), ba.data_ptr(), n_big, spp_begin=0, seed=1)
import ctypes as C
ts = torch.cuda.Stream(); stream = C.c_void_p(ts.cuda_stream); torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
for k in range(7):
    ev[k].record(ts)
    dev.intersect_dev(bo.data_ptr(), bd.data_ptr(), ba.data_ptr(), n_big, 2, t.data_ptr(), i.data_ptr(), stream)
torch.cuda.synchronize()
ms = np.median([ev[k].elapsed_time(ev[k + 1]) for k in range(1, 6)])
print("mode-2 call on %d bounce rays: %.3f ms = %.0f Mrays/s" % (n_big, ms, n_big / ms / 1e3), flush=True)
img, st = dev.render_pt(cam, 16, max_depth=20, seed=1)
t3 = time.time(); img, st = dev.render_pt(cam, 16, max_depth=20, seed=1); t4 = time.time()
print("path tracing 1024x1024x16: %.0f Msamples/s, image mean %.6f" % (16 * 1024 * 1024 / (t4 - t3) / 1e6, float(img.mean())), flush=True)
