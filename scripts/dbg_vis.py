import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import renderer_amd as ra, orc, meshgen
import test_gpu_fuzz as F
seed, n_tri = 11, 300
rng = np.random.default_rng(2000 + seed)
scene, v = F._soup(ra, rng, n_tri, 5)
S = orc.OracleScene(scene.raw())
dev = ra.Device(device=0); dev.upload(scene.describe())
n = 150000
o, d, aux, _ = F._rays(rng, v, n, 5, ra)
q = 57134
oo, dd, aa = np.ascontiguousarray(o[:, q:q+1]), np.ascontiguousarray(d[:, q:q+1]), aux[q:q+1].copy()
for lim in (0.691434, 0.6915, 0.692, 0.7, 1.0, 0.69137):
    L = np.array([lim], np.float32)
    print(os.environ.get("MIRT_VIS_FAST", "fast"), "lim", lim, "gpu", dev.visibility(oo, dd, aa, L, 2), "oracle", S.visibility(oo, dd, aa, L, 2, threads=1))
