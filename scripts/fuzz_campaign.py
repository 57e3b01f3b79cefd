"""One-off differential fuzz campaign on the GPU box (not part of the test suite): many random triangle soups of
random size, scale and position against the CPU oracle, closest hit in the three modes + visibility.
usage: python scripts/fuzz_campaign.py first_seed n_seeds [rays]   (prints one line per seed, details of differing rays;
SEEDS=a,b,c in the environment replaces the range, AXIS=1 adds near-axis-parallel rays,
VIS1=1 adds the visibility query in mode 1,
MESH=1 takes closed meshes inside Cornell walls (renderer_amd/meshgen.py) instead of soups)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import renderer_amd as ra, orc
import test_gpu_fuzz as F

axis_rays = F._axis_rays  # (the generator lives with the tests that use it too)


first, count = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
dev = ra.Device(device=0)
total = bad = 0
t0 = time.time()
seeds = [int(x) for x in os.environ["SEEDS"].split(",")] if os.environ.get("SEEDS") else range(first, first + count)
for seed in seeds:
    rng = np.random.default_rng(seed)
    n_tri = int(10 ** rng.uniform(0.5, 4.3))
    scale = float(10 ** rng.uniform(-3, 3))
    offset = tuple((rng.uniform(-1, 1, 3) * scale * 10 ** rng.uniform(-1, 1.5)).tolist()) if rng.random() < 0.6 else (0.0, 0.0, 0.0)
    n_obj = int(rng.integers(1, 7)) if n_tri >= 40 else 1
    if os.environ.get("MESH"):  # closed, smooth-shaded meshes inside walls instead of a soup: shared edges and corners
        import renderer_amd.meshgen as meshgen
        level = int(rng.integers(1, 6))
        m = meshgen.cornell_with_meshes(level, seed, smooth=bool(rng.integers(0, 2)))
        m["verts"] = (m["verts"].astype(np.float64) * scale + np.asarray(offset)).astype(np.float32)
        scene = ra.HostScene.from_arrays(m["verts"], m["tris"], m["tri_obj"], m["objects"], m["tri_vn"])
        v = m["verts"][m["tris"].astype(np.int64)]
        n_tri, n_obj = len(v), len(m["objects"])
    else:
        scene, v = F._soup(ra, rng, n_tri, n_obj, scale, offset)
    tri_obj = scene.raw()["tri_obj"]
    S = orc.OracleScene(scene.raw())
    dev.upload(scene.describe())
    o, d, aux, last = F._rays(rng, v, n, n_obj, ra, scale, offset, tri_obj)
    if os.environ.get("AXIS"): axis_rays(rng, v, o, d, scale)
    line = "seed %d tris %d scale %.3g offset (%.3g %.3g %.3g):" % ((seed, n_tri, scale) + offset)
    for mode, a in ((0, None), (1, last), (2, aux)):
        tg, ig = dev.intersect(o, d, a, mode)
        tc, ic = S.intersect(o, d, a, mode, threads=16)
        diff = np.nonzero((ig != ic) | ~((tg == tc) | (np.isnan(tg) & np.isnan(tc))))[0]
        line += " m%d %d" % (mode, len(diff))
        total += n; bad += len(diff)
        for q in diff[:3]:
            print("   mode %d ray %d gpu (%d, %r) ref (%d, %r) o %s d %s" % (mode, q, ig[q], float(tg[q]), ic[q], float(tc[q]), o[:, q], d[:, q]))
    tc, ic = S.intersect(o, d, aux, 2, threads=16)
    lim = np.where(ic >= 0, tc, scale).astype(np.float32)
    lim = (lim * rng.choice(np.array([0.5, 1.0 - 1e-6, 1.0, 1.0 + 1e-6, 1.0 + 1e-4, 2.0, 100.0], np.float32), n)).astype(np.float32)
    for mode, a in ((2, aux), (0, None)) + (((1, last),) if os.environ.get("VIS1") else ()):
        vg = dev.visibility(o, d, a, lim, mode)
        vc = S.visibility(o, d, a, lim, mode, threads=16)
        nd = int((vg != vc).sum())
        line += " v%d %d" % (mode, nd)
        for q in np.nonzero(vg != vc)[0][:3]:
            tq, iq = S.intersect(o[:, q:q + 1].copy(), d[:, q:q + 1].copy(), None if a is None else a[q:q + 1].copy(), mode, threads=1)
            print("   visibility mode %d ray %d gpu %d ref %d limit %r closest (%d, %r) o %s d %s" % (mode, q, vg[q], vc[q], float(lim[q]), iq[0], float(tq[0]), o[:, q], d[:, q]))
            if iq[0] >= 0: print("      tri", v[iq[0]].tolist())
        total += n; bad += nd
    print(line, flush=True)
    S.close(); scene.close()
print("campaign: %d differing of %d queries, %.0f s" % (bad, total, time.time() - t0))
