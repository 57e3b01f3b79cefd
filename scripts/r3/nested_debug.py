"""debug: nested / disjoint / overlapping media, GPU vs oracle ray counts"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import renderer_amd as ra, orc
from renderer_amd import meshgen

def med(ss, sa, ior=1.0, g=0.0):
    ob = meshgen._obj(ra.OBJ_VPM, has_vn=0, ior=ior)
    ob["sigma_s"] = (ss,) * 3; ob["sigma_a"] = (sa,) * 3; ob["g"] = g; ob["volumetric"] = 1
    return ob

def scene(parts):
    verts, tris, tri_obj, objects, off = [], [], [], [], 0
    for oi, (pv, pf, ob) in enumerate(parts):
        verts.append(pv); tris.append(pf + off); tri_obj.append(np.full(len(pf), oi, np.int32)); objects.append(ob); off += len(pv)
    return ra.HostScene.from_arrays(np.concatenate(verts).astype(np.float32), np.concatenate(tris).astype(np.uint32),
                                    np.concatenate(tri_obj), np.array(objects, ra.OBJECT_DTYPE), None)

lv, lf, _ = meshgen._quad((-0.3, -0.3, -1), (0.3, -0.3, -1), (0.3, 0.3, -1), (-0.3, 0.3, -1), (0, 0, 1))
light = (lv, lf, meshgen._obj(ra.OBJ_EMISSIVE, (40, 40, 40), has_vn=0))
cases = {
    "nested ior1": [light, meshgen._box(1.5) + (med(0.5, 0.1),), meshgen._box(0.4, (0.2, 0.1, 0.0)) + (med(2.0, 0.2),)],
    "nested inner no scatter": [light, meshgen._box(1.5) + (med(0.5, 0.1),), meshgen._box(0.4, (0.2, 0.1, 0.0)) + (med(0.0, 0.2),)],
    "nested outer no scatter": [light, meshgen._box(1.5) + (med(0.0, 0.1),), meshgen._box(0.4, (0.2, 0.1, 0.0)) + (med(2.0, 0.2),)],
    "nested ior": [light, meshgen._box(1.5) + (med(0.5, 0.1),), meshgen._box(0.4, (0.2, 0.1, 0.0)) + (med(2.0, 0.2, ior=1.3),)],
    "disjoint": [light, meshgen._box(0.4, (-0.6, 0, 0)) + (med(1.0, 0.1),), meshgen._box(0.4, (0.6, 0.1, 0.0)) + (med(2.0, 0.2),)],
    "overlap": [light, meshgen._box(0.5, (-0.2, 0, 0)) + (med(1.0, 0.1),), meshgen._box(0.5, (0.3, 0.1, 0.0)) + (med(2.0, 0.2),)],
    "single": [light, meshgen._box(1.5) + (med(0.5, 0.1),)],
}
dev = ra.Device(device=0)
cam = ra.make_camera(64, 64, 45, position=(0.4, 0.3, 3.2), focus=(0, 0, 0))
spp = 16
for name, parts in cases.items():
    s = scene(parts); S = orc.OracleScene(s.raw()); dev.upload(s.describe())
    for integ in ("pt", "pttest"):
        for depth in (1, 2, 3, 20):
            if integ == "pt":
                g, st = dev.render_pt(cam, spp, seed=3, max_depth=depth); c, nr = S.render(cam, spp, seed=3, threads=16, max_depth=depth)
            else:
                g, st = dev.render_pttest(cam, spp, seed=3, max_depth=depth); c, nr = S.render_pttest(cam, spp, seed=3, threads=16, max_depth=depth)[:2]
            print("%-26s %-6s depth %2d rays gpu %8d cpu %8d  identical %.3f  mean %.5f vs %.5f" % (name, integ, depth, int(st.n_rays), nr, (g == c).all(-1).mean(), g.mean() / spp, c.mean() / spp), flush=True)
    S.close(); s.close()
