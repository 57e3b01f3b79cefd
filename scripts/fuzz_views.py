"""One-off: the repository's own scenes (Data/Config_*128.xml: glossy Cornell box, fog, textures, bump map, balls, cube,
spot light) from random viewpoints inside and outside the room, random image shapes, depth caps and first samples, both
integrators, GPU against the CPU oracle at the same seed. usage: fuzz_views.py first_seed n_seeds [spp]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import renderer_amd as ra, orc


def rmse(a, b):
    return float(np.sqrt(np.nanmean((np.clip(a, 0, 1) - np.clip(b, 0, 1)) ** 2)))


configs = sorted(f for f in os.listdir(os.path.join(ROOT, "Data")) if f.startswith("Config_") and "1024" not in f and "512" not in f)
dev = ra.Device(device=0)
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 16
loaded = {}
worst = 0.0
for seed in range(int(sys.argv[1]), int(sys.argv[1]) + int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    cfg = configs[int(rng.integers(0, len(configs)))]
    if cfg not in loaded:
        s = ra.HostScene.load_config(os.path.join(ROOT, "Data", cfg))
        loaded[cfg] = (s, orc.OracleScene(s.raw()))
    scene, S = loaded[cfg]
    dev.upload(scene.describe())
    base = scene.camera()
    pos = rng.uniform(-0.95, 0.95, 3) if rng.random() < 0.7 else np.asarray(base.position[:]) + rng.normal(size=3) * 0.5
    cam = ra.make_camera(int(rng.integers(8, 80)), int(rng.integers(8, 80)), int(rng.integers(10, 120)), position=tuple(pos.tolist()),
                         focus=tuple(rng.uniform(-0.8, 0.8, 3).tolist()), up=tuple(rng.normal(size=3).tolist()))
    kw = dict(max_depth=int(rng.integers(1, 21)), spp_begin=int(rng.integers(0, 200)))
    sd = int(rng.integers(1, 10000))
    for integ in ("pt", "pttest"):
        if integ == "pt":
            g, st = dev.render_pt(cam, spp, seed=sd, **kw); c, nr = S.render(cam, spp, seed=sd, threads=16, **kw); c2 = S.render(cam, spp, seed=sd + 100, threads=16, **kw)[0]
        else:
            g, st = dev.render_pttest(cam, spp, seed=sd, **kw); c, nr = S.render_pttest(cam, spp, seed=sd, threads=16, **kw)[:2]; c2 = S.render_pttest(cam, spp, seed=sd + 100, threads=16, **kw)[0]
        same, noise = rmse(g / spp, c / spp), rmse(c / spp, c2 / spp)
        ratio = same / max(noise, 1e-30)
        worst = max(worst, ratio if noise > 1e-6 else 0.0)
        print("seed %d %s %dx%d depth %d %s: same-seed rmse %.3g noise %.3g ratio %.3g | identical pixels %.3f | nan gpu %d cpu %d | rays gpu %d cpu %d | mean %.5f vs %.5f" % (
            seed, cfg[7:-4], cam.width, cam.height, kw["max_depth"], integ, same, noise, ratio, (g == c).all(-1).mean(), int(np.isnan(g).sum()), int(np.isnan(c).sum()),
            int(st.n_rays), nr, float(np.nanmean(g / spp)), float(np.nanmean(c / spp))), flush=True)
print("worst same-seed / noise ratio %.3g" % worst)
