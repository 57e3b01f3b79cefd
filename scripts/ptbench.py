"""Path-tracing A/B timing: renders the 1024x1024 Cornell scene on the device and prints Msamples/s.
usage: [MIRT_LIB=...] python3 scripts/ptbench.py [spp] [reps] [config.xml]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import renderer_amd as ra
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = sys.argv[3] if len(sys.argv) > 3 else "Config_Glossy1024.xml"
spl = int(sys.argv[4]) if len(sys.argv) > 4 else 0
mode = sys.argv[5] if len(sys.argv) > 5 else "pt"  # pt | pttest
scene = ra.HostScene.load_config(os.path.join(ROOT, "Data", cfg))
cam = scene.camera(); dev = ra.Device(device=0); dev.upload(scene.describe())
wh = cam.width * cam.height
acc = torch.zeros(wh * 3, dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
best = None
for r in range(reps + 1):
    acc.zero_(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = (dev.render_pttest_dev if mode == "pttest" else dev.render_pt_dev)(cam, acc.data_ptr(), spp, seed=1, spp_per_launch=spl)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if r and (best is None or dt < best[0]):
        best = (dt, st)
dt, st = best
print("%-32s %s spp %d spl %d: %.2f ms wall, gpu %.2f ms, intersect %.2f ms, %.1f Msamples/s, %.1f Mrays/s (+%.1f Mshadow/s), mean %.5f [%s]" % (
    os.path.basename(os.environ.get("MIRT_LIB", "libmirt.so")), cfg, spp, spl, dt * 1e3, st.gpu_ms, st.intersect_ms,
    wh * spp / dt / 1e6, st.n_rays / dt / 1e6, st.n_shadow_rays / dt / 1e6, float(acc.mean().item()) / spp, mode), flush=True)
