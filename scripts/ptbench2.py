"""Experiment: K contexts on ONE GPU render contiguous sample shards concurrently (one host thread each), so that the
latency-bound late bounces of one context's wavefront overlap with another's bulk bounces.
usage: python3 scripts/ptbench2.py [spp] [contexts] [spp_per_launch per context] [config]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import renderer_amd as ra
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
spl = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cfg = sys.argv[4] if len(sys.argv) > 4 else "Config_Glossy1024.xml"
scene = ra.HostScene.load_config(os.path.join(ROOT, "Data", cfg))
cam = scene.camera(); wh = cam.width * cam.height
devs = [ra.Device(device=0) for _ in range(K)]
for d in devs: d.upload(scene.describe())
accs = [torch.zeros(wh * 3, dtype=torch.float32, device="cuda") for _ in range(K)]
torch.cuda.synchronize()
def work(k, stats):
    b, c = ra.shard_samples(spp, K, k)
    stats[k] = devs[k].render_pt_dev(cam, accs[k].data_ptr(), c, spp_begin=b, seed=1, spp_per_launch=spl)
best = None
for r in range(4):
    for a in accs: a.zero_()
    torch.cuda.synchronize()
    stats = [None] * K
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k, stats)) for k in range(K)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if r and (best is None or dt < best): best = dt
tot = sum(accs)
print("contexts %d spp %d spl %d: %.2f ms wall, %.1f Msamples/s, mean %.5f" % (K, spp, spl, best * 1e3, wh * spp / best / 1e6, float(tot.mean().item()) / spp), flush=True)
