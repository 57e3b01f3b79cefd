"""GPU parity tests of the light-connection integrator (PathTracerTest.cpp:7-135)
against the CPU oracle's restatement. Same bars as the path tracer (tests/test_gpu_pt.py): at equal spp
and the SAME RNG streams the images agree to a small fraction of the noise; against another seed the
RMSE is the noise's; closest-hit and shadow-ray counts agree; and, as the reference's authors used it,
the converged PTtest image agrees with the converged PathTracer image."""
import os

import numpy as np
import pytest

from conftest import rmse

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _parity(dev, oracle, cam, spp, seed=1, same_tol=0.03):
    g, st = dev.render_pttest(cam, spp, seed=seed)
    c, nr, ns = oracle.render_pttest(cam, spp, seed=seed, threads=16)
    c2, _, _ = oracle.render_pttest(cam, spp, seed=seed + 100, threads=16)
    noise = rmse(c / spp, c2 / spp)
    same = rmse(g / spp, c / spp)
    cross = rmse(g / spp, c2 / spp)
    assert same <= same_tol * noise, (same, noise)
    assert cross <= 1.15 * noise, (cross, noise)
    mg, mc = (g / spp).mean((0, 1)), (c / spp).mean((0, 1))
    assert (np.abs(mg - mc) / mc <= 0.01).all(), (mg, mc)
    assert abs(int(st.n_rays) - nr) / nr < 1e-4
    # the oracle counts a connection per RANDOM vertex; the device skips the shadow ray when the
    # BSDF / cosine terms are already zero, so it issues at most as many
    assert 0.5 * ns < st.n_shadow_rays <= ns, (st.n_shadow_rays, ns)
    return g, c


def test_cornell_glossy_pttest_parity(gpu, cornell, cornell_oracle):
    gpu.upload(cornell.describe())
    g, c = _parity(gpu, cornell_oracle, cornell.camera(), 32)
    assert np.isclose(g, c, rtol=1e-4, atol=1e-5).all(-1).mean() > 0.5  # most pixels equal to rounding


def test_other_scenes_pttest_parity(gpu, ra, orc):
    for cfg in ("Config_Balls2_128.xml", "Config_Cube128.xml", "Config_SpotLight128.xml"):
        s = ra.HostScene.load_config(os.path.join(ROOT, "Data", cfg))
        S = orc.OracleScene(s.raw())
        gpu.upload(s.describe())
        _parity(gpu, S, s.camera(), 16)
        S.close()
        s.close()


def test_pttest_agrees_with_path_tracer(gpu, cornell):
    """the reference's own cross-check: both integrators converge to the same image. PathTracer
    divides by the vignetting factor and carries the camera cosine, PTtest does neither; the two
    cancel at pixel centres, so the images agree up to noise."""
    gpu.upload(cornell.describe())
    cam = cornell.camera()
    a, _ = gpu.render_pt(cam, 1024, seed=5)
    b, _ = gpu.render_pttest(cam, 256, seed=6)
    a, b = a / 1024, b / 256
    # block means (16x16 pixels) beat the per-pixel noise down; clamping a noisy pixel biases it, and the
    # two integrators have very different noise in the caustic under the glass ball, hence the slack
    blk = lambda x: np.clip(x, 0, 1).reshape(8, 16, 8, 16, 3).mean((1, 3))
    diff = np.abs(blk(a) - blk(b))
    assert diff.max() < 0.2 and np.median(diff) < 0.01, (diff.max(), np.median(diff))
    assert np.abs(np.clip(a, 0, 1).mean((0, 1)) - np.clip(b, 0, 1).mean((0, 1))).max() < 0.01
    # and the light connections are what they are for: less noise per sample
    a2, _ = gpu.render_pt(cam, 256, seed=7)
    b2, _ = gpu.render_pttest(cam, 256, seed=8)
    a3, _ = gpu.render_pt(cam, 256, seed=9)
    assert rmse(b / 1, b2 / 256) < 0.7 * rmse(a2 / 256, a3 / 256)


def test_pttest_determinism_sharding_depth(gpu, cornell, cornell_oracle, ra):
    gpu.upload(cornell.describe())
    cam = cornell.camera()
    a, _ = gpu.render_pttest(cam, 12, seed=3)
    b, _ = gpu.render_pttest(cam, 12, seed=3)
    assert np.array_equal(a, b)
    lo, _ = gpu.render_pttest(cam, 5, spp_begin=0, seed=3)
    hi, _ = gpu.render_pttest(cam, 7, spp_begin=5, seed=3)
    assert np.abs((lo + hi) - a).max() <= 1e-5 * max(1.0, a.max())
    c, _ = gpu.render_pttest(cam, 12, seed=3, spp_per_launch=5)
    assert np.abs(c - a).max() <= 1e-5 * max(1.0, a.max())
    for md in (0, 1, 2):  # the depth cap drops the capped vertex with its connection (MCRenderer.cpp:181-191)
        g, st = gpu.render_pttest(cam, 8, max_depth=md, seed=2)
        c, nr, ns = cornell_oracle.render_pttest(cam, 8, max_depth=md, seed=2, threads=16)
        assert rmse(g / 8, c / 8) < 5e-3 and abs(int(st.n_rays) - nr) / nr < 1e-3
    g, st = gpu.render_pttest(cam, 0, seed=1)
    assert st.n_rays == 0 and not g.any()


def test_pttest_in_media_parity(gpu, ra, orc):
    """PathTracerTest inside a homogeneous medium, as the reference computes it (PathTracerTest.cpp:20-135 with
    SceneVPMObject.cpp:239-267): connections from scattering events through the phase function, transmittance along
    the eye segments, originProb never divided out. Same-stream parity with the oracle's restatement on BallsInFog."""
    s = ra.HostScene.load_config(os.path.join(ROOT, "Data", "Config_Fog128.xml"))
    S = orc.OracleScene(s.raw())
    gpu.upload(s.describe())
    g, c = _parity(gpu, S, s.camera(), 24, same_tol=0.05)
    assert g.sum() > 0
    # determinism and sample sharding hold there too
    a, _ = gpu.render_pttest(s.camera(), 6, seed=3)
    lo, _ = gpu.render_pttest(s.camera(), 2, spp_begin=0, seed=3)
    hi, _ = gpu.render_pttest(s.camera(), 4, spp_begin=2, seed=3)
    assert np.abs((lo + hi) - a).max() <= 1e-5 * max(1.0, a.max())
    S.close()
    s.close()


def test_pttest_through_host_surface(ra, cornell_oracle, tmp_path):
    """Renderer::render with <type>PTtest</type> -> PathTracerTestHIP::renderPixels -> mirt_render_pttest"""
    s = ra.HostScene.load_config(os.path.join(ROOT, "Data", "Config_Glossy128_PTtest.xml"))
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        img, st = s.render(spp=8, seed=1)
    finally:
        os.chdir(cwd)
    c, _, _ = cornell_oracle.render_pttest(s.camera(), 8, seed=1, threads=16)
    assert rmse(img, c / 8) < 5e-3 and st.n_shadow_rays > 0
    assert os.path.exists(tmp_path / "cornell_pttest.pfm")
    s.close()


def test_phong_object_parity(gpu, ra, orc):
    """a Phong (glossy + diffuse) displaced sphere in the mesh scene: both integrators against the oracle"""
    import meshgen
    from test_gpu_pt import _parity as _pt_parity

    m = meshgen.cornell_with_meshes(3)
    m["objects"]["type"][6] = ra.OBJ_PHONG
    m["objects"]["color"][6] = (0.5, 0.4, 0.2)
    m["objects"]["color2"][6] = (0.4, 0.4, 0.4)
    m["objects"]["coeff"][6] = 8.0
    m["objects"]["type"][8] = ra.OBJ_DIFFUSE  # (the glass sphere amplifies ulp differences: test_gpu_pt.py)
    m["objects"]["color"][8] = (0.6, 0.6, 0.6)
    s = ra.HostScene.from_arrays(m["verts"], m["tris"], m["tri_obj"], m["objects"], m["tri_vn"])
    S = orc.OracleScene(s.raw())
    gpu.upload(s.describe())
    cam = ra.make_camera(96, 96, 65)
    _pt_parity(gpu, S, cam, 16)
    _parity(gpu, S, cam, 16)
    S.close()
    s.close()
