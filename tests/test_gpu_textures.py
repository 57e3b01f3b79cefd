"""GPU parity of image textures (Texture.cpp:43-94 through DiffuseMaterial / GlossyMaterial)
against the CPU oracle, with the bars of tests/test_gpu_pt.py and tests/test_gpu_pttest.py: same RNG streams
-> images equal to a small fraction of the noise, ray counts equal."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import rmse
from test_gpu_pt import _parity as pt_parity
from test_gpu_pttest import _parity as pttest_parity
from test_textures import _textured_floor, _two_tone

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_textured_room_parity(gpu, ra, orc):
    """Data/Config_Textured128.xml: checker floor and ball (Diffuse <Texture>, with <coordTransform>), striped
    glossy wall with a coefficient texture, a Phong ball with both lobes textured and <colorScale>"""
    s = ra.HostScene.load_config(os.path.join(ROOT, "Data", "Config_Textured128.xml"))
    S = orc.OracleScene(s.raw())
    gpu.upload(s.describe())
    g, c = pt_parity(gpu, S, s.camera(), 32)
    assert np.isclose(g, c, rtol=1e-4, atol=1e-5).all(-1).mean() > 0.5  # most pixels equal to rounding
    g, c = pttest_parity(gpu, S, s.camera(), 16)
    assert np.isclose(g, c, rtol=1e-4, atol=1e-5).all(-1).mean() > 0.5
    S.close()
    s.close()


def test_textured_floor_from_arrays(gpu, ra, orc):
    cam = ra.make_camera(32, 32, 32, position=(0, 0.5, 2.5), focus=(0, -1, 0))
    for tex in (_two_tone(ra), ra.make_texture(np.array([[0.2, 0.7], [0.5, 0.3]], np.float32))):
        s = _textured_floor(ra, tex)
        S = orc.OracleScene(s.raw())
        gpu.upload(s.describe())
        g, st = gpu.render_pttest(cam, 64, max_depth=3, seed=2)
        c, nr, _ = S.render_pttest(cam, 64, max_depth=3, seed=2, threads=16)
        assert int(st.n_rays) == nr
        assert np.isclose(g, c, rtol=1e-4, atol=1e-5).all(-1).mean() > 0.9
        left, right = g[12:20, 2:8].mean(), g[12:20, 24:30].mean()
        if tex["texels"].ndim == 3:
            assert 3.3 < right / left < 4.6, (left, right)
        S.close()
        s.close()


def test_descriptor_without_texture_fields(gpu, ra, orc):
    """struct_size versions the descriptor: a caller built against the form that ends at obj_trees still
    uploads, and renders the constant colours"""
    s = _textured_floor(ra, _two_tone(ra))
    d = s.describe()
    assert d.n_textures == 1
    d.struct_size = ra.SceneDesc.tri_vt.offset
    gpu.upload(d)
    cam = ra.make_camera(16, 16, 16, position=(0, 0.5, 2.5), focus=(0, -1, 0))
    a, _ = gpu.render_pttest(cam, 32, max_depth=1, seed=2)
    m = s.raw()
    plain = ra.HostScene.from_arrays(m["verts"], m["tris"], m["tri_obj"], m["objects"], m["tri_vn"])
    gpu.upload(plain.describe())
    b, _ = gpu.render_pttest(cam, 32, max_depth=1, seed=2)
    assert np.array_equal(a, b)
    # bad texture descriptions are refused on the host, before anything reaches the GPU
    d = s.describe()
    bad = ra.texture_structs([ra.make_texture(np.zeros((2, 2, 3), np.float32))])
    bad[0].channels = 2
    d.textures = C.cast(bad, C.c_void_p)
    with pytest.raises(ra.MirtError):
        gpu.upload(d)
    d = s.describe()
    ot = np.full(2, 5, ra.OBJECT_TEX_DTYPE)
    d.object_tex = ot.ctypes.data
    with pytest.raises(ra.MirtError):
        gpu.upload(d)
    plain.close()
    s.close()


def test_bump_room_parity(gpu, ra, orc):
    """Data/Config_Bump128.xml: bump-mapped mirror and glass balls and a bump-mapped diffuse floor. The
    perturbed normal (SceneObject.cpp:250-283) feeds the hit predicate, the frames, the cosine terms and
    the Fresnel split. The device's atan2 / sin / cos are not glibc's, so more paths than usual differ in
    the last bits and a few take another branch: the bar on equal-stream images is looser here."""
    s = ra.HostScene.load_config(os.path.join(ROOT, "Data", "Config_Bump128.xml"))
    S = orc.OracleScene(s.raw())
    gpu.upload(s.describe())
    # closest hits under the predicate, rays from inside the room
    import meshgen
    o, d, aux = meshgen.random_rays(200000, 77, s.n_objects, inside_obj=2)
    tg, ig = gpu.intersect(o, d, aux, ra.MODE_NO_SELF)
    tc, ic = S.intersect(o, d, aux, ra.MODE_NO_SELF, threads=16)
    assert (ig == ic).mean() >= 0.9999, (ig != ic).sum()
    both = (ig == ic) & (ic >= 0)
    assert np.array_equal(tg[both], tc[both])
    lim = np.where(tc > 0, tc * np.random.default_rng(3).uniform(0.5, 1.5, len(tc)), 1.0).astype(np.float32)
    vg = gpu.visibility(o, d, aux, lim, ra.MODE_NO_SELF)
    vc = S.visibility(o, d, aux, lim, ra.MODE_NO_SELF, threads=16)
    assert (vg == vc).mean() >= 0.9999
    pt_parity(gpu, S, s.camera(), 32, same_tol=0.25)
    pttest_parity(gpu, S, s.camera(), 16, same_tol=0.25)
    # instrumented runs are not built for bump scenes
    import torch
    n = 1024
    od = torch.as_tensor(o[:, :n].copy()).cuda()
    dd = torch.as_tensor(d[:, :n].copy()).cuda()
    ad = torch.as_tensor(aux[:n].copy()).cuda()
    t = torch.empty(n, device="cuda")
    i = torch.empty(n, dtype=torch.int32, device="cuda")
    with pytest.raises(ra.MirtError, match="bump"):
        gpu.intersect_count_dev(od.data_ptr(), dd.data_ptr(), ad.data_ptr(), n, ra.MODE_NO_SELF, t.data_ptr(), i.data_ptr())
    S.close()
    s.close()
