"""Developer smoke run on a GPU box: closest-hit parity + image parity against the oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import renderer_amd as ra, orc

s = ra.HostScene.load_config(os.path.join(ROOT, "Data/Config_Glossy128.xml"))
raw = s.raw(); O = orc.OracleScene(raw)
dev = s.upload(); print(dev.info(), flush=True)
cam = s.camera()
rng = np.random.default_rng(0)
n = 200000
# primary-like rays from the camera + random interior rays
o = np.zeros((3, n), np.float32); d = rng.normal(size=(3, n)).astype(np.float32)
o[2, : n // 2] = 2.0; d[2, : n // 2] = -np.abs(d[2, : n // 2]) - 2.0
o[:, n // 2:] = rng.uniform(-0.9, 0.9, size=(3, n - n // 2)).astype(np.float32)
d /= np.linalg.norm(d, axis=0, keepdims=True)
t0 = time.time(); tg, idg = dev.intersect(o, d, None, 0); print("gpu intersect", time.time() - t0, flush=True)
t0 = time.time(); tc, idc = O.intersect(o, d, None, 0); print("cpu intersect", time.time() - t0, flush=True)
print("mode0 tid match", (idg == idc).mean(), "t exact", (tg == tc).mean(), "hits", (idc >= 0).mean())
contact = rng.integers(-1, s.n_objects, n); inside = np.where(rng.random(n) < 0.8, -1, 2)
aux = ra.pack_ids(contact, inside)
tg, idg = dev.intersect(o, d, aux, 2); tc, idc = O.intersect(o, d, aux, 2)
print("mode2 tid match", (idg == idc).mean(), "t exact", (tg == tc).mean(), "hits", (idc >= 0).mean())
last = rng.integers(0, s.n_tris, n).astype(np.int32)
tg, idg = dev.intersect(o, d, last, 1); tc, idc = O.intersect(o, d, last, 1)
print("mode1 tid match", (idg == idc).mean(), "t exact", (tg == tc).mean())
spp = 64
t0 = time.time(); ig, st = dev.render_pt(cam, spp, seed=1); print("gpu render", time.time() - t0, st.gpu_ms, st.intersect_ms, st.n_rays, flush=True)
t0 = time.time(); ic, nr = O.render(cam, spp, seed=1, threads=16); print("cpu render", time.time() - t0, nr, flush=True)
ic2, _ = O.render(cam, spp, seed=2, threads=16)
def rmse(a, b): return float(np.sqrt((np.clip(a, 0, 1) - np.clip(b, 0, 1)) ** 2).sum(-1).mean()) if False else float(np.sqrt(((np.clip(a,0,1)-np.clip(b,0,1))**2).sum(-1).mean()))
print("rmse gpu vs cpu same seed", rmse(ig / spp, ic / spp), "cpu seed1 vs seed2", rmse(ic / spp, ic2 / spp))
print("mean gpu", (ig / spp).mean((0, 1)), "cpu", (ic / spp).mean((0, 1)), "identical pixels", (ig == ic).all(-1).mean())
print("rays gpu", st.n_rays, "cpu", nr)
