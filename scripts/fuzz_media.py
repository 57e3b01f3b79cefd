"""One-off: path-traced images of random scenes with homogeneous media, GPU against the CPU oracle at the same seed
(not part of the test suite). A scene: an emissive quad, optionally an outer fog cube or a floor, and 1 .. 5 closed
objects (cubes of 12 triangles: the small-object shortcut of findInsideObject; displaced icospheres of 80 / 320: the
object-tree walk) that are media (random sigma_s, sigma_a, g, IOR 1 or 1.1 .. 1.5), glass, mirrors-by-glossy or diffuse, at
random places, overlapping or nested as they fall; the camera outside or inside. Both integrators.
usage: [RICH=1|2] [EXTREME=1] fuzz_media.py first_seed n_seeds [spp]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import renderer_amd as ra, orc
from renderer_amd import meshgen


SCALE = [1.0, (0.0, 0.0, 0.0)]  # SCALED=1: the whole scene (and the camera) scaled by 1e-3 .. 1e3 and moved, per seed
RICH = int(os.environ.get("RICH", "0"))  # RICH=1: smooth normals, mirrors, Phong objects, more emitters, 1 280-triangle balls; RICH=2: random image shape, field of view, depth cap, first sample and seed too


def rmse(a, b):
    return float(np.sqrt(np.mean((np.clip(a, 0, 1) - np.clip(b, 0, 1)) ** 2)))


def build(rng):
    return meshgen.random_media_scene(rng, rich=RICH, extreme=bool(os.environ.get("EXTREME")), scale=SCALE[0], offset=SCALE[1])


dev = ra.Device(device=0)
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 24
worst = 0.0
for seed in range(int(sys.argv[1]), int(sys.argv[1]) + int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    if os.environ.get("SCALED"):
        r2 = np.random.default_rng(seed + 10 ** 6)
        SCALE[0] = float(10 ** r2.uniform(-3, 3))
        SCALE[1] = tuple((r2.uniform(-1, 1, 3) * SCALE[0] * 10 ** r2.uniform(-1, 1.3)).tolist()) if r2.random() < 0.5 else (0.0, 0.0, 0.0)
    scene, kinds = build(rng)
    S = orc.OracleScene(scene.raw())
    dev.upload(scene.describe())
    if rng.random() < 0.35:
        pos = tuple(rng.uniform(-1.0, 1.0, 3).tolist())
    else:
        w = rng.normal(size=3); w /= np.linalg.norm(w)
        pos = tuple((w * 3.2).tolist())
    kw = {}
    if RICH >= 2:  # any image shape and field of view, any depth cap, any first sample, any seed
        cam = ra.make_camera(int(rng.integers(8, 96)), int(rng.integers(8, 96)), int(rng.integers(15, 121)), position=tuple((np.asarray(pos) * SCALE[0] + np.asarray(SCALE[1])).tolist()),
                             focus=tuple((rng.uniform(-0.3, 0.3, 3) * SCALE[0] + np.asarray(SCALE[1])).tolist()), up=tuple(rng.normal(size=3).tolist()))
        kw = dict(max_depth=int(rng.integers(1, 21)), spp_begin=int(rng.integers(0, 100)))
        sd = int(rng.integers(1, 10000))
    else:
        cam = ra.make_camera(64, 64, 45, position=pos, focus=tuple(rng.uniform(-0.3, 0.3, 3).tolist()))
        sd = 3
    for integ in ("pt", "pttest"):
        if integ == "pt":
            g, st = dev.render_pt(cam, spp, seed=sd, **kw)
            c, nr = S.render(cam, spp, seed=sd, threads=16, **kw)
            c2, _ = S.render(cam, spp, seed=sd + 100, threads=16, **kw)
        else:
            g, st = dev.render_pttest(cam, spp, seed=sd, **kw)
            c, nr = S.render_pttest(cam, spp, seed=sd, threads=16, **kw)[:2]
            c2 = S.render_pttest(cam, spp, seed=sd + 100, threads=16, **kw)[0]
        same, noise = rmse(g / spp, c / spp), rmse(c / spp, c2 / spp)
        ratio = same / max(noise, 1e-30)
        worst = max(worst, ratio if noise > 1e-6 else 0.0)
        print("seed %d kinds %s tris %d cam %s %s: same-seed rmse %.3g noise %.3g ratio %.3g | identical pixels %.3f | nan gpu %d cpu %d | rays gpu %d cpu %d | mean %.5f vs %.5f" % (
            seed, "".join(str(k) for k in kinds), scene.n_tris, "in " if max(abs(p) for p in pos) <= 1.0 else "out", integ, same, noise, ratio,
            (g == c).all(-1).mean(), int(np.isnan(g).sum()), int(np.isnan(c).sum()), int(st.n_rays), nr, float(np.nanmean(g / spp)), float(np.nanmean(c / spp))), flush=True)
    if os.environ.get("DEPTHSCAN"):  # the path tracer at rising depth caps: where do the two sides part?
        print("   scale %.4g offset %s camera %s" % (SCALE[0], SCALE[1], [round(float(x), 5) for x in cam.position[:]]))
        for k, ob in enumerate(scene.raw()["objects"]):
            print("   object %d type %d ior %g tris %d ss %s sa %s g %g" % (k, ob["type"], ob["ior"], ob["tri_count"], ob["sigma_s"], ob["sigma_a"], ob["g"]))
        for depth in range(1, min(kw.get("max_depth", 20), 12) + 1):
            k2 = dict(kw); k2["max_depth"] = depth
            g, st = dev.render_pt(cam, spp, seed=sd, **k2); c, nr = S.render(cam, spp, seed=sd, threads=16, **k2)
            print("   depth cap %2d: rays gpu %d cpu %d, identical pixels %.3f, mean %.5f vs %.5f" % (depth, int(st.n_rays), nr, (g == c).all(-1).mean(), g.mean() / spp, c.mean() / spp), flush=True)
    S.close(); scene.close()
print("worst same-seed / noise ratio %.3g" % worst)
