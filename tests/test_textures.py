"""Image textures (Texture.cpp:5-176, SimpleShape.cpp:274-298, DiffuseMaterial.cpp:23,83,99,
GlossyMaterial.cpp:11-12,66,74,78) on the CPU side: the oracle's lookups against hand-computed values, the
host loader (.pfm, XML, descriptor) and closed-form renders. The reference has no tests or golden outputs
for textures and its OpenCV / GLUT build does not compile here, so above the arithmetic restated from the
source these are "parity unpinned" (DESIGN.md §3)."""
import ctypes as C
import os

import numpy as np
import pytest

import meshgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tex_color(orc, ra, tex, coord):
    arr = ra.texture_structs([tex])
    c = np.asarray(coord, np.float32)
    out = np.zeros(3, np.float32)
    orc.lib().orc_texture_color(C.cast(arr, C.c_void_p), orc._p(c), orc._p(out))
    return out


def test_texture_lookup_known_answers(orc, ra):
    rgb = np.array([[[1, 0, 0], [0, 1, 0]], [[0, 0, 1], [1, 1, 1]]], np.float32)  # rows: file order
    t = ra.make_texture(rgb)
    # texel centres sit at (i + 0.5) / size (Texture.cpp:46-47)
    assert np.array_equal(_tex_color(orc, ra, t, (0.25, 0.25, 0)), rgb[0, 0])
    assert np.array_equal(_tex_color(orc, ra, t, (0.75, 0.25, 0)), rgb[0, 1])
    assert np.array_equal(_tex_color(orc, ra, t, (0.25, 0.75, 0)), rgb[1, 0])
    assert np.array_equal(_tex_color(orc, ra, t, (0.75, 0.75, 0)), rgb[1, 1])
    # half way between the four centres: their mean; between two: their mean
    assert np.allclose(_tex_color(orc, ra, t, (0.5, 0.5, 0)), rgb.mean((0, 1)), atol=1e-7)
    assert np.allclose(_tex_color(orc, ra, t, (0.5, 0.25, 0)), rgb[0].mean(0), atol=1e-7)
    # wrapping, both ways (Texture.cpp:57-65)
    for shift in ((1, 0), (0, 1), (-1, 0), (-3, 2)):
        assert np.array_equal(_tex_color(orc, ra, t, (0.75 + shift[0], 0.25 + shift[1], 0)), rgb[0, 1])
    # across the border the neighbour is the opposite edge's texel
    assert np.allclose(_tex_color(orc, ra, t, (0.0, 0.25, 0)), (rgb[0, 0] + rgb[0, 1]) / 2, atol=1e-7)
    # transforms: coordinates first, colours last (Texture.cpp:45, 79)
    coord = np.diag([2, 2, 1, 1]).astype(np.float32)
    coord[0, 3] = 0.25
    color = np.diag([0.5, 0.5, 0.5, 1]).astype(np.float32)
    color[1, 3] = 0.125
    t2 = ra.make_texture(rgb, coord, color)
    assert np.allclose(_tex_color(orc, ra, t2, (0.25, 0.125, 0)), rgb[0, 1] * 0.5 + (0, 0.125, 0), atol=1e-7)
    # a one-texel texture is the constant colour wherever it is read (Texture::setColor)
    one = ra.make_texture(np.array([[[0.3, 0.6, 0.9]]], np.float32))
    for c in ((0, 0, 0), (0.3, 7.7, 0), (-5.2, 0.01, 3)):
        assert np.allclose(_tex_color(orc, ra, one, c), (0.3, 0.6, 0.9), atol=1e-7)
    # gray textures: the fourth sample repeats (top, left) (Texture.cpp:86), kept as the reference has it
    g = ra.make_texture(np.array([[1, 2], [4, 8]], np.float32))
    assert np.allclose(_tex_color(orc, ra, g, (0.5, 0.5, 0)), (1 + 2 + 4 + 4) / 4.0, atol=1e-6)
    assert np.allclose(_tex_color(orc, ra, g, (0.75, 0.75, 0)), 8.0, atol=0)  # weights (0,0,0,1)... of rt
    assert np.allclose(_tex_color(orc, ra, g, (0.25, 0.75, 0)), 4.0, atol=0)


def _textured_floor(ra, texture, kind=None, floor=None):
    """floor_and_light with texture coordinates (x, z) -> (u, v) over the floor and `texture` bound to it"""
    kind = ra.OBJ_DIFFUSE if kind is None else kind
    m = meshgen.floor_and_light(albedo=0.6, emission=5.0, floor=floor)
    s = ra.HostScene.from_arrays(m["verts"], m["tris"], m["tri_obj"], m["objects"], m["tri_vn"])
    v = m["verts"][m["tris"].astype(np.int64)]                      # (T, 3, 3)
    vt = np.stack([(v[..., 0] + 20) / 40, (v[..., 2] + 20) / 40, np.zeros(v.shape[:2])], -1).reshape(len(v), 9)
    ot = np.full(2, -1, ra.OBJECT_TEX_DTYPE)
    ot["diffuse"][1] = 0
    s.set_textures(vt.astype(np.float32), [texture], ot)
    return s


def test_tex_coord_is_barycentric(orc, ra):
    s = _textured_floor(ra, ra.make_texture(np.full((2, 2, 3), 0.3, np.float32)))
    S = orc.OracleScene(s.raw())
    for p in ((0, -1, 0), (-20, -1, -20), (5, -1, -13), (19.5, -1, 19.5)):
        for tid in (2, 3):  # both floor triangles extrapolate to the same plane map
            out = np.zeros(3, np.float32)
            orc.lib().orc_tex_coord(S.h, tid, orc._p(np.asarray(p, np.float32)), orc._p(out))
            assert np.allclose(out, ((p[0] + 20) / 40, (p[2] + 20) / 40, 0), atol=2e-6), (p, tid, out)
    S.close()
    s.close()


def test_texture_replaces_the_constant_colour(orc, ra):
    """PTtest at maxDepth 1 shows exactly the direct light rho / pi * E * G (tests/test_oracle.py); with a
    uniform texture of 0.3 bound to a floor whose <color> is 0.6 the answer is the one for rho = 0.3, for
    an rgb texture, a gray one, and through a colour transform."""
    cam = ra.make_camera(16, 16, 16, position=(0, 0.5, 2.5), focus=(0, -1, 0))
    t = (-1 - 0.5) / (-1.5 / np.linalg.norm((0, -1.5, -2.5)))
    x = np.array([0, 0.5, 2.5]) + t * np.array([0, -1.5, -2.5]) / np.linalg.norm((0, -1.5, -2.5))
    g = (np.arange(200) + 0.5) / 200 - 0.5
    yx, yz = np.meshgrid(g, g)
    dvec = np.stack([yx - x[0], np.full_like(yx, 1 - x[1]), yz - x[2]], -1)
    r2 = (dvec ** 2).sum(-1)
    geom = (dvec[..., 1] ** 2 / r2 / r2).mean()
    half = np.diag([0.5, 0.5, 0.5, 1]).astype(np.float32)
    for tex in (ra.make_texture(np.full((2, 2, 3), 0.3, np.float32)), ra.make_texture(np.full((3, 5), 0.3, np.float32)),
                ra.make_texture(np.full((4, 4, 3), 0.6, np.float32), None, half)):
        s = _textured_floor(ra, tex)
        S = orc.OracleScene(s.raw())
        img, _, _ = S.render_pttest(cam, 3000, max_depth=1, seed=3, threads=8)
        got = img[7:9, 7:9].mean() / 3000
        expect = 5.0 * 0.3 / np.pi * geom
        assert abs(got - expect) / expect < 0.05, (got, expect)
        S.close()
        s.close()


def _two_tone(ra):
    """four texels (0.2, 0.2, 0.8, 0.8) along u with u' = 8 u - 3.5 = 0.5 + 0.2 x on the floor: between the
    first two texel centres (x in [-1.9, -0.6]) the lookup is 0.2, between the last two (x in [0.6, 1.9]) 0.8"""
    return ra.make_texture(np.array([[[0.2] * 3, [0.2] * 3, [0.8] * 3, [0.8] * 3]], np.float32),
                           np.array([[8, 0, 0, -3.5], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32))


def test_texture_pattern_reaches_the_image(orc, ra):
    """a texture that is dark for x < 0 and bright for x > 0 on the floor: the left and right parts of the
    image (columns that look at x ~ -1 and x ~ +1) differ by the texels' ratio"""
    tex = _two_tone(ra)
    s = _textured_floor(ra, tex)
    S = orc.OracleScene(s.raw())
    cam = ra.make_camera(16, 16, 16, position=(0, 0.5, 2.5), focus=(0, -1, 0))
    img, _, _ = S.render_pttest(cam, 1500, max_depth=1, seed=5, threads=8)
    left, right = img[6:10, 1:4].mean(), img[6:10, 12:15].mean()
    assert 3.6 < right / left < 4.4, (left, right)  # 0.8 / 0.2
    S.close()
    s.close()


def test_glossy_scatter_reads_its_textures_at_the_origin(orc, ra):
    """GlossyMaterial.cpp:11-12: scatter() evaluates colorTex / coeffTex at getTexCoord(tid, outRay.origin)
    before outRay.origin is assigned (:17), i.e. at the world origin, whatever the hit point. With the same
    two-texel texture as above bound as the glossy colour, the path tracer's image therefore shows ONE
    colour across the floor (the texel under the world origin's extrapolated coordinate)."""
    tex = _two_tone(ra)
    floor = meshgen._obj(ra.OBJ_GLOSSY, (0.5, 0.5, 0.5), coeff=1.0)
    s = _textured_floor(ra, tex, floor=floor)
    ot = np.full(2, -1, ra.OBJECT_TEX_DTYPE)
    ot["glossy_color"][1] = 0
    s.set_textures(s.raw()["tri_vt"], [tex], ot)
    S = orc.OracleScene(s.raw())
    cam = ra.make_camera(16, 16, 16, position=(0, 0.5, 2.5), focus=(0, -1, 0))
    img, _ = S.render(cam, 3000, max_depth=2, seed=5, threads=8)
    left, right = img[6:10, 1:4].mean(), img[6:10, 12:15].mean()
    assert 0.75 < right / left < 1.35, (left, right)
    S.close()
    s.close()


def test_textured_config_loads(ra, tmp_path):
    s = ra.HostScene.load_config(os.path.join(ROOT, "Data", "Config_Textured128.xml"))
    r = s.raw()
    assert r["tri_vt"].shape == (s.n_tris, 9)
    ot = r["object_tex"]
    assert len(ot) == s.n_objects == 8 and len(r["textures"]) == 6
    # light + plain walls read none; the Phong ball both lobes; the diffuse ball / floor one; the glossy wall two
    assert [int(o["diffuse"]) >= 0 for o in ot] == [False, True, True, False, False, False, True, False]
    assert [int(o["glossy_color"]) >= 0 for o in ot] == [False, True, False, False, False, False, False, True]
    assert [int(o["glossy_coeff"]) >= 0 for o in ot] == [False, False, False, False, False, False, False, True]
    phong_diffuse = r["textures"][ot["diffuse"][1]]
    assert phong_diffuse["texels"].shape == (8, 8, 3)
    assert np.allclose(phong_diffuse["coord"].diagonal(), (4, 4, 1, 1))      # <coordTransform>
    assert np.allclose(phong_diffuse["color"].diagonal(), (0.5, 0.5, 0.5, 1))  # <colorScale>
    coeff = r["textures"][ot["glossy_coeff"][7]]
    assert coeff["texels"].shape == (4, 4) and 4 <= coeff["texels"].min() and coeff["texels"].max() <= 40
    # objects without texture coordinates read (0,0,0) (SimpleShape.cpp:276-277)
    plain = r["objects"][3]
    assert not r["tri_vt"][plain["tri_begin"]:plain["tri_begin"] + plain["tri_count"]].any()
    # the floor's coordinates follow plane_uv.obj
    floor = r["objects"][6]
    assert set(np.unique(r["tri_vt"][floor["tri_begin"]:floor["tri_begin"] + 2])) == {0.0, 1.0}
    s.close()
    # <colorTransform> sets the COORDINATE transform in the reference (Texture.cpp:154-155), kept;
    # unreadable files are refused
    mats = open(os.path.join(ROOT, "Data", "Materials.xml")).read()
    scene = open(os.path.join(ROOT, "Data", "Scene_Textured.xml")).read()
    for sub in ("objs", "textures"):
        os.symlink(os.path.join(ROOT, "Data", sub), str(tmp_path / sub))
    cfg = open(os.path.join(ROOT, "Data", "Config_Textured128.xml")).read()
    (tmp_path / "Scene_Textured.xml").write_text(scene)
    (tmp_path / "c.xml").write_text(cfg)
    quirk = mats.replace("<path>textures/stripes.pfm</path>\n\t</ColorTexture>",
                         "<path>textures/stripes.pfm</path>\n\t\t<colorTransform>[3, 0, 0, 0][0, 3, 0, 0][0, 0, 1, 0][0, 0, 0, 1]</colorTransform>\n\t</ColorTexture>")
    assert quirk != mats
    (tmp_path / "Materials.xml").write_text(quirk)
    s = ra.HostScene.load_config(str(tmp_path / "c.xml"))
    r = s.raw()
    wall = r["textures"][r["object_tex"]["glossy_color"][7]]
    assert np.allclose(wall["coord"].diagonal(), (3, 3, 1, 1)) and np.allclose(wall["color"], np.eye(4))
    s.close()
    (tmp_path / "Materials.xml").write_text(mats.replace("textures/coeff.pfm", "textures/missing.pfm"))
    with pytest.raises(ra.MirtError, match="cannot open"):
        ra.HostScene.load_config(str(tmp_path / "c.xml"))


def _bumped_floor(ra, texels, coord=None, color=None, vt_scale=1.0):
    """floor_and_light whose floor carries a bump texture (texture coordinates as in _textured_floor)"""
    m = meshgen.floor_and_light(albedo=0.6, emission=5.0)
    s = ra.HostScene.from_arrays(m["verts"], m["tris"], m["tri_obj"], m["objects"], m["tri_vn"])
    v = m["verts"][m["tris"].astype(np.int64)]
    vt = np.stack([(v[..., 0] + 20) / 40, (v[..., 2] + 20) / 40, np.zeros(v.shape[:2])], -1).reshape(len(v), 9) * vt_scale
    ot = np.full(2, -1, ra.OBJECT_TEX_DTYPE)
    ot["bump"][1] = 0
    s.set_textures(vt.astype(np.float32), [ra.make_texture(texels, coord, color)], ot)
    return s, m, vt.astype(np.float32)


def test_bump_normal_known_answers(orc, ra):
    """SceneObject::getWorldNormal's bump branch (SceneObject.cpp:250-283) and Texture::getGrad
    (Texture.cpp:96-147) against the same formulas written out in float64 numpy."""
    # a flat texture leaves the normal alone (zero gradient: angle 0)
    s, m, vt = _bumped_floor(ra, np.full((4, 4), 0.5, np.float32))
    S = orc.OracleScene(s.raw())
    assert np.allclose(S.world_normal(2, (0.3, -1, 0.7)), (0, 1, 0), atol=1e-6)
    S.close()
    s.close()
    # a ramp along u, read with a coordinate transform and a colour scale
    ramp = np.tile(0.1 * np.arange(8, dtype=np.float32), (8, 1))
    coord = np.diag([16.0, 16.0, 1, 1]).astype(np.float32)
    color = np.diag([2.0, 2.0, 2.0, 1]).astype(np.float32)
    s, m, vt = _bumped_floor(ra, ramp, coord, color)
    S = orc.OracleScene(s.raw())
    checked = 0
    for tid in (2, 3):
        vps = m["verts"][m["tris"][tid].astype(np.int64)].astype(np.float64)
        vts = vt[tid].reshape(3, 3).astype(np.float64)
        for p in ((0.3, -1, 0.7), (-1.2, -1, 0.4), (2.0, -1, -1.0), (-0.4, -1, -2.2)):
            u = (p[0] + 20) / 40 * 16 * 8 - 0.5
            if int(np.floor(u)) % 8 == 7:
                continue  # the forward difference would wrap from the last texel to the first
            g = np.array([0.1, 0.0, 0.0])                      # image gradient (right - left, top - bottom)
            var_n = np.linalg.norm(g) * 2.0                      # through the colour transform
            uvg = np.array([g[0] / np.linalg.norm(g) / 16, 0.0, var_n])  # direction through inverse(coordTransform)
            b1, b2 = vps[1] - vps[0], vps[2] - vps[0]
            d1, d2 = vts[1] - vts[0], vts[2] - vts[0]
            k2 = (uvg[0] * d1[1] - uvg[1] * d1[0]) / (d1[1] * d2[0] - d1[0] * d2[1])
            k1 = (uvg[1] - k2 * d2[1]) / d1[1]
            dl = k1 * b1 / np.linalg.norm(b1) + k2 * b2 / np.linalg.norm(b2)
            n = np.array([0.0, 1.0, 0.0])
            dh = n * uvg[2]
            if not np.isfinite(dl).all():
                continue  # this triangle's first edge runs along u (duv1.y = 0): the reference divides by zero
            ang = np.arctan2(np.linalg.norm(dh), np.linalg.norm(dl))
            ax = np.cross(dl, dh)
            ax /= np.linalg.norm(ax)
            rot = n * np.cos(ang) + np.cross(ax, n) * np.sin(ang) + ax * np.dot(ax, n) * (1 - np.cos(ang))
            got = S.world_normal(tid, p)
            assert np.allclose(got, rot, atol=2e-4), (tid, p, got, rot)
            assert abs(np.linalg.norm(got) - 1) < 1e-5 and got[1] < 0.999  # tilted, still unit length
            checked += 1
    assert checked >= 3
    S.close()
    s.close()


def test_bump_config_loads(ra):
    s = ra.HostScene.load_config(os.path.join(ROOT, "Data", "Config_Bump128.xml"))
    r = s.raw()
    ot = r["object_tex"]
    assert [int(o["bump"]) >= 0 for o in ot] == [False, True, True, False, False, False, True, False]
    for k in (1, 2, 6):
        t = r["textures"][ot["bump"][k]]
        assert t["texels"].shape == (16, 16)  # gray after ConfigManager.cpp:161's toGrayScale
    assert np.allclose(r["textures"][ot["bump"][6]]["color"].diagonal(), (3, 3, 3, 1))
    assert np.allclose(r["textures"][ot["bump"][2]]["coord"].diagonal(), (2, 2, 1, 1))
    # the mean of the rgb texel (Texture.cpp:174)
    assert abs(r["textures"][ot["bump"][1]]["texels"][0, 0] - 0.5 * (1 + 0.9 + 1.1) / 3) < 1e-6
    s.close()
