"""debug: one scripts/fuzz_media.py scene by seed, ray counts and image differences per depth cap"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
src = open(os.path.join(ROOT, "scripts", "fuzz_media.py")).read().split("dev = ra.Device(device=0)")[0]
ns = {"__file__": os.path.join(ROOT, "scripts", "fuzz_media.py")}
exec(compile(src, "fuzz_media_head", "exec"), ns)
import renderer_amd as ra, orc
dev = ra.Device(device=0)
for seed in [int(x) for x in sys.argv[1].split(",")]:
    rng = np.random.default_rng(seed)
    scene, kinds = ns["build"](rng)
    S = orc.OracleScene(scene.raw()); dev.upload(scene.describe())
    if rng.random() < 0.35:
        pos = tuple(rng.uniform(-1.0, 1.0, 3).tolist())
    else:
        w = rng.normal(size=3); w /= np.linalg.norm(w); pos = tuple((w * 3.2).tolist())
    cam = ra.make_camera(64, 64, 45, position=pos, focus=tuple(rng.uniform(-0.3, 0.3, 3).tolist()))
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    print("seed", seed, "kinds", kinds, "objects:")
    for k, o in enumerate(scene.raw()["objects"]):
        print("  ", k, "type", int(o["type"]), "ior %.3f" % float(o["ior"]), "tris", int(o["tri_count"]), "ss", o["sigma_s"], "sa", o["sigma_a"], "g %.2f" % float(o["g"]))
    for depth in (1, 2, 3, 4, 5, 6, 8, 20):
        g, st = dev.render_pt(cam, spp, seed=3, max_depth=depth); c, nr = S.render(cam, spp, seed=3, threads=16, max_depth=depth)
        diff = np.nonzero(~(g == c).all(-1))
        print("depth %2d rays gpu %8d cpu %8d  differing pixels %4d  mean %.5f vs %.5f  first differing %s" % (
            depth, int(st.n_rays), nr, len(diff[0]), g.mean() / spp, c.mean() / spp,
            [(int(y), int(x), g[y, x].round(4).tolist(), c[y, x].round(4).tolist()) for y, x in zip(diff[0][:2], diff[1][:2])]), flush=True)
    S.close(); scene.close()
