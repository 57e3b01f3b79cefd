import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import renderer_amd as ra, orc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = ra.HostScene.load_config(os.path.join(ROOT, "Data", "Config_Bump128.xml"))
S = orc.OracleScene(s.raw())
gpu = ra.Device()
gpu.upload(s.describe())
cam = s.camera()
for md in (0, 1, 2, 3, 5):
    g, st = gpu.render_pt(cam, 4, max_depth=md, seed=1)
    c, nr = S.render(cam, 4, max_depth=md, seed=1, threads=16)
    diff = np.abs(g - c).max(-1)
    bad = np.argwhere(diff > 1e-3 * (1 + np.abs(c).max(-1)))
    print("max_depth", md, "rays gpu/cpu", st.n_rays, nr, "pixels differing", len(bad), "first", bad[:8].tolist())
# path of one differing pixel in the oracle
g, st = gpu.render_pt(cam, 1, max_depth=3, seed=1)
c, nr = S.render(cam, 1, max_depth=3, seed=1, threads=16)
bad = np.argwhere(np.abs(g - c).max(-1) > 1e-3 * (1 + np.abs(c).max(-1)))
print("1 spp: rays", st.n_rays, nr, "bad", len(bad))
for y, x in bad[:6]:
    verts, _, n = S.trace_path(cam, int(y) * cam.width + int(x), 0, max_depth=3)
    print("pixel", y, x, "gpu", g[y, x], "cpu", c[y, x])
    for k in range(n):
        print("   v%d" % k, np.array2string(np.asarray(verts[k]), precision=4, max_line_width=200))
