"""The math layer against the REFERENCE ITSELF: tests/golden/ref_math.npz holds inputs and outputs of
the reference's own header-only code (nvVector.h, nvMatrix.h, LocalFrame.h, HGPhaseFunc.h), compiled
in place by oracle/ref_math.cpp and recorded by scripts/gen_ref_math_golden.py. The oracle's
restatements and the product's host math must reproduce every output bit for bit (NaNs in the same
places). When oracle/_ref/libref_math.so is present (build container) the fixture is also re-derived
live, so a stale fixture cannot pass."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "ref_math.npz"))
f32 = np.float32


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def same(a, b):
    a, b = np.asarray(a, f32), np.asarray(b, f32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32) | (np.isnan(a) * 0xFFFFFFFF).astype(np.uint32),
                                                 b.view(np.uint32) | (np.isnan(b) * 0xFFFFFFFF).astype(np.uint32))


def rows(fn, n, *cols, out=3):
    res = np.zeros((n, out), f32)
    for i in range(n):
        fn(*[P(np.ascontiguousarray(c[i])) if isinstance(c[i], np.ndarray) else C.c_float(float(c[i])) for c in cols],
           P(res[i]))
    return res


def test_oracle_frames_and_phase(orc):
    L = orc.lib()
    L.orc_hg_eval.restype = C.c_float
    n = len(G["lf_n"])
    s, t = np.zeros((n, 3), f32), np.zeros((n, 3), f32)
    for i in range(n):
        L.orc_local_frame(P(G["lf_n"][i].copy()), P(s[i]), P(t[i]))
    assert same(s, G["lf_s"]) and same(t, G["lf_t"])
    assert same(rows(L.orc_to_world, n, G["lf_n"], G["lf_a"]), G["lf_to_world"])
    hv = np.array([L.orc_hg_eval(C.c_float(float(G["hg_g"][i])), P(G["hg_in"][i].copy()), P(G["hg_out"][i].copy()))
                   for i in range(n)], f32)
    assert same(hv, G["hg_value"])


def test_oracle_normals_rotation_vectors(orc):
    L = orc.lib()
    L.orc_dot.restype = C.c_float
    L.orc_length.restype = C.c_float
    n = len(G["tri_p"])
    p, cn = G["tri_p"], G["tri_cn"]
    assert same(rows(L.orc_face_normal, n, p[:, 0], p[:, 1], p[:, 2]), G["tri_face_normal"])
    assert same(rows(L.orc_interp_normal, n, p[:, 0], p[:, 1], p[:, 2], cn[:, 0], cn[:, 1], cn[:, 2], G["tri_pos"]),
                G["tri_shading_normal"])
    assert same(rows(L.orc_rotate, n, G["rot_axis"], G["rot_angle"], G["rot_v"]), G["rot_out"])
    assert same(rows(L.orc_normalize, n, G["vec_a"]), G["vec_normalize"])
    assert same(rows(L.orc_div, n, G["vec_a"], G["vec_d"]), G["vec_div"])
    assert same([L.orc_dot(P(G["vec_a"][i].copy()), P(G["vec_b"][i].copy())) for i in range(n)], G["vec_dot"])
    assert same([L.orc_length(P(G["vec_a"][i].copy())) for i in range(n)], G["vec_length"])
    # the zero cases really are in the fixture
    assert (G["vec_normalize"][:3] == 0).all() and same(G["vec_div"][:16], G["vec_a"][:16])


def test_oracle_matrix_ops(orc):
    """the matrix arithmetic of the oracle's texture path (coordinate / colour transforms, the inverse
    Texture::getGrad needs) against the reference's nvMatrix.h"""
    L = orc.lib()
    L.orc_mat_inverse.argtypes = [C.c_void_p] * 2
    L.orc_mat_inverse.restype = None
    L.orc_mat_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    L.orc_mat_apply.restype = None
    n = len(G["xf_m"])
    inv = np.zeros((n, 16), f32)
    wv = np.zeros((n, 3), f32)
    for i in range(n):
        L.orc_mat_inverse(P(G["xf_m"][i].copy()), P(inv[i]))
        L.orc_mat_apply(P(G["xf_m"][i].copy()), P(G["xf_v"][i].copy()), 1.0, P(wv[i]))
    assert same(wv, G["xf_world_vertex"])
    # a singular matrix: the reference returns identity when a row is all zero or the last pivot is 0, and
    # otherwise whatever its elimination leaves; parity is required wherever its result is finite
    ok = np.isfinite(G["xf_inverse"]).all(1)
    assert ok.mean() > 0.9 and same(inv[ok], G["xf_inverse"][ok])


def test_host_transforms(ra):
    """the product's host math (vecmath.h) that flattens a scene: world vertices and the normal matrix"""
    L = ra.lib()
    n = len(G["xf_m"])
    assert same(rows(L.mirth_world_vertex, n, G["xf_m"], G["xf_v"]), G["xf_world_vertex"])
    got = rows(L.mirth_normal_transform, n, G["xf_m"], G["xf_vn"])
    # singular matrices: the reference's inverse() yields whatever its elimination leaves (inf / nan);
    # scene transforms are invertible, so parity is required on the invertible ones and only
    # "not finite or equal" on the rest
    ok = np.isfinite(G["xf_normal"]).all(1) & np.isfinite(G["xf_inverse"]).all(1)
    assert ok.mean() > 0.9
    assert same(got[ok], G["xf_normal"][ok])


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref_math.so")),
                    reason="oracle/_ref is built only where /root/reference exists")
def test_fixture_is_what_the_reference_computes_now():
    R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_math.so"))
    n = len(G["lf_n"])
    s, t = np.zeros((n, 3), f32), np.zeros((n, 3), f32)
    for i in range(n):
        R.ref_local_frame(P(G["lf_n"][i].copy()), P(s[i]), P(t[i]))
    assert same(s, G["lf_s"]) and same(t, G["lf_t"])
    assert same(rows(R.ref_normal_transform, len(G["xf_m"]), G["xf_m"], G["xf_vn"]), G["xf_normal"])
    assert same(rows(R.ref_rotate, n, G["rot_axis"], G["rot_angle"], G["rot_v"]), G["rot_out"])
