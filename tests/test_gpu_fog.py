"""GPU parity of the homogeneous-medium path (SceneVPMObject.cpp:4-233, Scene::findInsideObject
Scene.cpp:354-390, per-object trees Scene.cpp:287-320) against the CPU oracle: SURVEY.md C5."""
import os

import numpy as np
import pytest

import meshgen
from conftest import rmse

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _parity(dev, oracle, cam, spp, seed=1, same_tol=0.05, ray_tol=2e-3):
    g, st = dev.render_pt(cam, spp, seed=seed)
    c, nr = oracle.render(cam, spp, seed=seed, threads=16)
    c2, _ = oracle.render(cam, spp, seed=seed + 100, threads=16)
    noise = rmse(c / spp, c2 / spp)
    same = rmse(g / spp, c / spp)
    cross = rmse(g / spp, c2 / spp)
    assert same <= same_tol * noise, (same, noise)
    assert cross <= 1.15 * noise, (cross, noise)
    mg, mc = (g / spp).mean((0, 1)), (c / spp).mean((0, 1))
    assert (np.abs(mg - mc) / mc <= 0.01).all(), (mg, mc)
    assert abs(int(st.n_rays) - nr) / nr < ray_tol, (int(st.n_rays), nr)
    return g, c


def test_balls_in_fog_image_parity(gpu, ra, orc):
    """C5: camera inside the fog cube; glass, mirror, glossy and diffuse objects inside it."""
    s = ra.HostScene.load_config(os.path.join(ROOT, "Data", "Config_Fog128.xml"))
    assert s.describe().n_obj_trees == 1
    S = orc.OracleScene(s.raw())
    gpu.upload(s.describe())
    g, c = _parity(gpu, S, s.camera(), 32)
    assert (g == c).all(-1).mean() > 0.3  # many pixels bit-identical
    S.close()
    s.close()


def test_small_object_tree_shortcut_equals_the_tree_walk(gpu, ra, monkeypatch):
    """findInsideObject on a small object (the 12-triangle fog cube) tests the triangles first and only then the boxes
    above the ones that are hit (pt_kernels.h inside_small); MIRT_VOL_SMALL=0 at upload makes it walk the object's tree
    like any larger one. Same streams, same answers: the images are equal bit for bit, both integrators, and so are
    the ray counts; camera inside (BallsInFog) and outside (a fog ball in the Cornell box) the medium."""
    s = ra.HostScene.load_config(os.path.join(ROOT, "Data", "Config_Fog128.xml"))
    cam = s.camera()
    out = {}
    for small in ("1", "0"):
        monkeypatch.setenv("MIRT_VOL_SMALL", small)
        gpu.upload(s.describe())
        out[small] = (gpu.render_pt(cam, 24, seed=3), gpu.render_pttest(cam, 8, seed=3))
    monkeypatch.delenv("MIRT_VOL_SMALL")
    for k in (0, 1):
        a, b = out["1"][k], out["0"][k]
        assert np.array_equal(a[0], b[0]), float(np.abs(a[0] - b[0]).max())
        assert int(a[1].n_rays) == int(b[1].n_rays)
    assert out["1"][0][0].mean() > 0.01
    s.close()


def test_medium_known_answers_on_gpu(gpu, ra):
    """the closed forms of tests/test_oracle.py::test_medium_known_answers, on the device"""
    cam = ra.make_camera(32, 32, 400, position=(0, 0, 1), focus=(0, 0, -1))
    for sa, expect in ((0.5, 50 * np.exp(-0.5 * 2.0)), (1e-6, 50.0)):
        s = meshgen.fog_box_scene((0, 0, 0), (sa, sa, sa))
        gpu.upload(s.describe())
        spp = 256
        img, _ = gpu.render_pt(cam, spp, seed=2)
        got = (img / spp)[8:24, 8:24].mean()
        assert abs(got - expect) / expect < 0.02, (sa, got, expect)
        s.close()


def test_anisotropic_medium_and_camera_outside(gpu, ra, orc):
    """Henyey-Greenstein g = 0.6, IOR 1.2 boundary (Fresnel on the way in and out), camera OUTSIDE
    the medium looking in through the boundary, a diffuse floor inside."""
    fv, ff, _ = meshgen._quad((-1.2, -0.8, -1.2), (1.2, -0.8, -1.2), (1.2, -0.8, 1.2), (-1.2, -0.8, 1.2), (0, 1, 0))
    floor = (fv, ff, meshgen._obj(ra.OBJ_DIFFUSE, (0.7, 0.6, 0.5), has_vn=0))
    s = meshgen.fog_box_scene((0.5, 0.4, 0.3), (0.05, 0.1, 0.15), ior=1.2, g=0.6, extra=[floor])
    S = orc.OracleScene(s.raw())
    gpu.upload(s.describe())
    cam = ra.make_camera(48, 48, 60, position=(0.3, 0.2, 3.0), focus=(0, -0.2, -1))
    _parity(gpu, S, cam, 64)
    S.close()
    s.close()


def test_render_refuses_medium_without_its_tree(gpu, ra):
    s = meshgen.fog_box_scene((0.3, 0.3, 0.3), (0.2, 0.2, 0.2))
    d = s.describe()
    d.n_obj_trees = 0
    gpu.upload(d)  # closest-hit queries do not need the object trees
    o = np.array([[0.0], [0.0], [1.0]], np.float32)
    dd = np.array([[0.0], [0.0], [-1.0]], np.float32)
    t, i = gpu.intersect(o, dd)
    assert i[0] >= 0 and abs(t[0] - 2.0) < 1e-6
    with pytest.raises(RuntimeError, match="own tree"):
        gpu.render_pt(ra.make_camera(8, 8, 10), 1)
    s.close()
