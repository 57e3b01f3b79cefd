"""one-off: arguments nobody should pass, through the C-ABI: each must come back as an error or as an answer, never
as a crash or a hang"""
import os, sys, numpy as np, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import renderer_amd as ra
from renderer_amd import meshgen
dev = ra.Device(device=0)
def attempt(name, f):
    try:
        r = f()
        print("%-44s -> %s" % (name, r), flush=True)
    except ra.MirtError as e:
        print("%-44s -> refused: %s" % (name, str(e)[:110]), flush=True)
    except Exception as e:
        print("%-44s -> python %s: %s" % (name, type(e).__name__, str(e)[:90]), flush=True)

m = meshgen.cornell_with_meshes(1)
scene = ra.HostScene.from_arrays(m["verts"], m["tris"], m["tri_obj"], m["objects"], m["tri_vn"])
dev.upload(scene.describe())
def img_stats(r): return "mean %.4g nan %d rays %d" % (float(np.nanmean(r[0])), int(np.isnan(r[0]).sum()), int(r[1].n_rays))
for name, cam in (("camera position == focus", ra.make_camera(16, 16, 20, position=(0, 0, 0), focus=(0, 0, 0))),
                  ("camera up parallel to the view", ra.make_camera(16, 16, 20, position=(0, 0, 2), focus=(0, 0, 0), up=(0, 0, 1))),
                  ("camera sight distance 0", ra.make_camera(16, 16, 0)),
                  ("camera NaN position", ra.make_camera(16, 16, 20, position=(np.nan, 0, 2))),
                  ("camera 1 x 1", ra.make_camera(1, 1, 20)),
                  ("camera 1 x 4000", ra.make_camera(1, 4000, 20))):
    attempt(name + " pt", lambda: img_stats(dev.render_pt(cam, 3)))
    attempt(name + " pttest", lambda: img_stats(dev.render_pttest(cam, 3)))
for name, kw in (("width 0", dict(w=0, h=16)), ("height -3", dict(w=16, h=-3)), ("width 70000", dict(w=70000, h=70000))):
    def f():
        c = ra.make_camera(16, 16, 20); c.width, c.height = kw["w"], kw["h"]
        p = ra.RenderParams(c, 0, 1, 20, 1, 0); st = ra.RenderStats()
        img = np.zeros((16, 16, 3), np.float32)
        rc = ra.lib().mirt_render_pt(dev.ctx, C.byref(p), img.ctypes.data_as(C.c_void_p), C.byref(st)); return "rc %d" % rc
    attempt("render " + name, f)
cam = ra.make_camera(16, 16, 20)
attempt("spp 0", lambda: img_stats(dev.render_pt(cam, 0)))
attempt("max_depth 0", lambda: img_stats(dev.render_pt(cam, 2, max_depth=0)))
attempt("max_depth 100000", lambda: img_stats(dev.render_pt(cam, 2, max_depth=100000)))
attempt("spp_begin 2^32 - 2, 4 samples", lambda: img_stats(dev.render_pt(cam, 4, spp_begin=(1 << 32) - 2)))
attempt("spp 2^31 on 1 x 1 (refuse or run?) skipped", lambda: "skipped")
attempt("spp_per_launch 2^31", lambda: img_stats(dev.render_pt(cam, 4, spp_per_launch=(1 << 31) - 1)))
o = np.zeros((3, 0), np.float32)
attempt("intersect n = 0", lambda: dev.intersect(o, o, None, 0)[0].shape)
o1 = np.zeros((3, 5), np.float32); d1 = np.ones((3, 5), np.float32)
attempt("intersect mode 7", lambda: dev.intersect(o1, d1, None, 7))
attempt("intersect mode 2 without aux", lambda: dev.intersect(o1, d1, None, 2))
attempt("aux ids far out of range (mode 2)", lambda: dev.intersect(o1, d1, np.full(5, 0x7fffffff, np.int32), 2)[1])
attempt("aux ids -1 packed wrongly (mode 2)", lambda: dev.intersect(o1, d1, np.full(5, -1, np.int32), 2)[1])
attempt("last tid out of range (mode 1)", lambda: dev.intersect(o1, d1, np.full(5, 1 << 30, np.int32), 1)[1])
# empty scene
attempt("scene without triangles", lambda: ra.HostScene.from_arrays(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint32), np.zeros(0, np.int32), np.zeros(0, ra.OBJECT_DTYPE), None))
# object parameters
ob = m["objects"].copy(); ob["type"][8] = 99
attempt("object of unknown type", lambda: (dev.upload(ra.HostScene.from_arrays(m["verts"], m["tris"], m["tri_obj"], ob, m["tri_vn"]).describe()), img_stats(dev.render_pt(cam, 2)))[1])
dev.upload(scene.describe())
# textures
vt = np.zeros((len(m["tris"]), 9), np.float32); ot = np.full(len(m["objects"]), -1, ra.OBJECT_TEX_DTYPE); ot["diffuse"][1] = 5
attempt("texture index out of range", lambda: (scene.set_textures(vt, [ra.make_texture(np.ones((2, 2, 3), np.float32))], ot), dev.upload(scene.describe()), img_stats(dev.render_pt(cam, 2)))[2])
ot["diffuse"][1] = 0
attempt("texture with NaN texels", lambda: (scene.set_textures(vt, [ra.make_texture(np.full((2, 2, 3), np.nan, np.float32))], ot), dev.upload(scene.describe()), img_stats(dev.render_pt(cam, 2)))[2])
attempt("tile out of the image", lambda: dev.render_tile_dev(cam, ra.Tile(10, 10, 16, 16), 0, 1))
print("done", flush=True)
