"""One-off: path-traced images of random triangle soups (slivers, degenerate and duplicated triangles, random vertex
normals; one emissive object), GPU against the CPU oracle at the same seed. usage: [TEX=1|2] fuzz_pt.py first_seed n_seeds   (TEX=1: random textures on the materials, TEX=2: bump maps too)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import renderer_amd as ra, orc
from renderer_amd import meshgen
import test_gpu_fuzz as F

def rmse(a, b):
    return float(np.sqrt(np.mean((np.clip(a, 0, 1) - np.clip(b, 0, 1)) ** 2)))

dev = ra.Device(device=0)
for seed in range(int(sys.argv[1]), int(sys.argv[1]) + int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    n_tri = int(10 ** rng.uniform(1.5, 3.7))
    n_obj = 6
    scale = float(10 ** rng.uniform(-2, 2.5)) if seed % 2 else 1.0
    offset = tuple((rng.uniform(-1, 1, 3) * scale * 8).tolist()) if seed % 4 == 3 else (0.0, 0.0, 0.0)
    try:
        scene, v = F._soup(ra, rng, n_tri, n_obj, scale, offset)
    except ra.MirtError as e:  # (a small soup can leave an object without triangles: the loader refuses it)
        print("seed %d skipped: %s" % (seed, e), flush=True)
        continue
    raw = scene.raw()
    objs = raw["objects"].copy()
    kinds = [ra.OBJ_EMISSIVE, ra.OBJ_DIFFUSE, ra.OBJ_GLOSSY, ra.OBJ_REFRACTIVE, ra.OBJ_PHONG, ra.OBJ_DIFFUSE]
    for k in range(n_obj):
        o = meshgen._obj(kinds[k], (8, 8, 8) if k == 0 else tuple(rng.uniform(0.3, 0.9, 3)), coeff=float(rng.uniform(2, 40)), ior=1.5,
                         has_vn=int(rng.integers(0, 2)), color2=tuple(rng.uniform(0.1, 0.5, 3)))
        objs[k] = np.array(o, ra.OBJECT_DTYPE)
    scene.close()
    scene = ra.HostScene.from_arrays(raw["verts"], raw["tris"], raw["tri_obj"], objs, raw["tri_vn"])
    if os.environ.get("TEX"):  # random textures (rgb / gray, 1 .. 16 texels a side, random transforms) on random materials, bump maps too
        texs = []
        for k in range(int(rng.integers(1, 5))):
            h, w = int(rng.integers(1, 17)), int(rng.integers(1, 17))
            tx = rng.uniform(0.05, 0.95, (h, w, 3) if rng.random() < 0.6 else (h, w)).astype(np.float32)
            coord = np.eye(4, dtype=np.float32); color = np.eye(4, dtype=np.float32)
            if rng.random() < 0.5:
                coord[0, 0], coord[1, 1] = rng.uniform(0.3, 4, 2); coord[0, 3], coord[1, 3] = rng.uniform(-1, 1, 2)
            if rng.random() < 0.3:
                color[0, 0] = color[1, 1] = color[2, 2] = rng.uniform(0.5, 1.0)
            texs.append(ra.make_texture(tx, coord, color))
        ot = np.full(n_obj, -1, ra.OBJECT_TEX_DTYPE)
        for k in range(1, n_obj):
            for field, pr in (("diffuse", 0.5), ("glossy_color", 0.4), ("glossy_coeff", 0.25), ("bump", 0.25 if os.environ["TEX"] == "2" else 0.0)):
                if rng.random() < pr:
                    ot[field][k] = int(rng.integers(0, len(texs)))
        vt = rng.uniform(-0.5, 1.5, (len(raw["tris"]), 3, 3)).astype(np.float32)
        vt[..., 2] = 0
        scene.set_textures(vt.reshape(-1, 9), texs, ot)
    S = orc.OracleScene(scene.raw())
    dev.upload(scene.describe())
    cam = ra.make_camera(96, 96, 60, position=(offset[0], offset[1], offset[2] + 3.2 * scale), focus=offset)
    spp = 32
    for integ in ("pt", "pttest"):
        if integ == "pt":
            g, st = dev.render_pt(cam, spp, seed=3)
            c, nr = S.render(cam, spp, seed=3, threads=16)
            c2, _ = S.render(cam, spp, seed=103, threads=16)
        else:
            g, st = dev.render_pttest(cam, spp, seed=3)
            c, nr = S.render_pttest(cam, spp, seed=3, threads=16)[:2]
            c2 = S.render_pttest(cam, spp, seed=103, threads=16)[0]
        same, noise = rmse(g / spp, c / spp), rmse(c / spp, c2 / spp)
        print("seed %d tris %d scale %.3g %s: same-seed rmse %.3g noise %.3g ratio %.3g | identical pixels %.3f | nan gpu %d cpu %d | rays gpu %d cpu %d | mean %s vs %s" % (
            seed, n_tri, scale, integ, same, noise, same / max(noise, 1e-30), (g == c).all(-1).mean(), int(np.isnan(g).sum()), int(np.isnan(c).sum()),
            int(st.n_rays), nr, np.nanmean(g / spp, (0, 1)).round(4), np.nanmean(c / spp, (0, 1)).round(4)), flush=True)
        if int(np.isnan(g).sum()) != int(np.isnan(c).sum()):  # where does the NaN enter? the pixel at rising depth caps
            ys, xs = np.nonzero(np.isnan(g).any(-1) != np.isnan(c).any(-1))
            for depth in range(1, 12):
                if integ == "pt":
                    gd = dev.render_pt(cam, spp, seed=3, max_depth=depth)[0]; cd = S.render(cam, spp, seed=3, threads=16, max_depth=depth)[0]
                else:
                    gd = dev.render_pttest(cam, spp, seed=3, max_depth=depth)[0]; cd = S.render_pttest(cam, spp, seed=3, threads=16, max_depth=depth)[0]
                print("   pixel (%d, %d) depth cap %d: gpu %s cpu %s" % (ys[0], xs[0], depth, gd[ys[0], xs[0]], cd[ys[0], xs[0]]), flush=True)
    S.close(); scene.close()
