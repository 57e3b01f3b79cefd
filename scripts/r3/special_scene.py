"""one-off: scenes whose vertices hold NaN / inf / 1e38 / denormal values: upload must refuse or work, queries must
equal the oracle's, nothing may fault"""
import os, sys, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import renderer_amd as ra, orc
import test_gpu_fuzz as F
dev = ra.Device(device=0)
for name, val in (("nan", np.nan), ("inf", np.inf), ("-inf", -np.inf), ("1e38", 1e38), ("3e38", 3e38), ("denormal", 1e-42), ("1e20", 1e20)):
    rng = np.random.default_rng(1)
    scene, v = F._soup(ra, rng, 300, 3)
    raw = scene.raw(); verts = raw["verts"].copy(); verts[7, 1] = val; verts[100, 0] = val
    s2 = ra.HostScene.from_arrays(verts, raw["tris"], raw["tri_obj"], raw["objects"], raw["tri_vn"])
    try:
        dev.upload(s2.describe())
    except ra.MirtError as e:
        print(name, "upload refused:", e, flush=True); s2.close(); scene.close(); continue
    S = orc.OracleScene(s2.raw())
    o, d, aux, last = F._rays(rng, v, 50000, 3, ra, tri_obj=raw["tri_obj"])
    line = name + ":"
    with np.errstate(all="ignore"):
        for mode, a in ((0, None), (1, last), (2, aux)):
            tg, ig = dev.intersect(o, d, a, mode); tc, ic = S.intersect(o, d, a, mode, threads=16)
            nd = int(((ig != ic) | ~((tg == tc) | (np.isnan(tg) & np.isnan(tc)))).sum())
            line += " m%d differ %d (hit rate %.2f)" % (mode, nd, (ic >= 0).mean())
        img, st = dev.render_pt(ra.make_camera(32, 32, 30, position=(0, 0, 3.2)), 4, seed=1)
        line += " | render ok, nan pixels %d" % int(np.isnan(img).any(-1).sum())
    print(line, flush=True)
    S.close(); s2.close(); scene.close()
print("done")
