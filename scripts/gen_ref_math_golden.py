#!/usr/bin/env python3
"""Writes tests/golden/ref_math.npz: inputs and outputs of the reference's own header-only math
(oracle/_ref/libref_math.so = oracle/ref_math.cpp compiled against /root/reference in place).
Run in the build container after `make -C oracle ref`; the .npz is data (fp32 inputs + outputs),
no reference text. Deterministic: seeded numpy generator."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_math.so"))
L.ref_hg_eval.restype = C.c_float
L.ref_dot.restype = C.c_float
L.ref_length.restype = C.c_float
f32 = np.float32


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def unit(v):
    return (v / np.linalg.norm(v, axis=-1, keepdims=True)).astype(f32)


def transforms(rng, n):
    """object transforms as the scene files use them: rotations, reflections, non-uniform scales,
    translations, products of those, and a few singular / near-singular matrices"""
    out = np.zeros((n, 4, 4), f32)
    for i in range(n):
        k = i % 8
        M = np.eye(4)
        if k in (0, 4, 5, 6):
            a = unit(rng.normal(size=3)).astype(np.float64)
            th = rng.uniform(-np.pi, np.pi)
            K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
            M[:3, :3] = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
        if k in (1, 4, 6):
            M[:3, :3] = M[:3, :3] @ np.diag(rng.choice([-1.0, 1.0], 3) * rng.uniform(0.05, 3.0, 3))
        if k in (2, 5, 6):
            M[:3, 3] = rng.uniform(-2, 2, 3)
        if k == 3:  # the walls of the Cornell scenes: axis swaps with sign flips
            perm = rng.permutation(3)
            M[:3, :3] = np.eye(3)[perm] * rng.choice([-1.0, 1.0], 3)[:, None]
            M[:3, 3] = rng.choice([-1.0, 0.0, 1.0], 3)
        if k == 7:
            M[:3, :3] = rng.normal(size=(3, 3))
            if i % 16 == 7:
                M[2, :3] = M[0, :3]  # singular
            if i % 32 == 15:
                M[1, :] = 0  # zero row
        out[i] = M.astype(f32)
    return out


def main():
    rng = np.random.default_rng(20241)
    n = 512
    g = {}
    M = transforms(rng, n)
    v = rng.uniform(-2, 2, (n, 3)).astype(f32)
    vn = unit(rng.normal(size=(n, 3)))
    wv, nt, inv = np.zeros((n, 3), f32), np.zeros((n, 3), f32), np.zeros((n, 16), f32)
    for i in range(n):
        m = np.ascontiguousarray(M[i].reshape(16))
        L.ref_world_vertex(P(m), P(v[i]), P(wv[i]))
        L.ref_normal_transform(P(m), P(vn[i]), P(nt[i]))
        L.ref_inverse(P(m), P(inv[i]))
    g.update(xf_m=M.reshape(n, 16), xf_v=v, xf_vn=vn, xf_world_vertex=wv, xf_normal=nt, xf_inverse=inv)

    # local frames: random normals plus the |n.z| > 0.99 switch and axis-aligned normals
    nn = unit(rng.normal(size=(n, 3)))
    special = np.array([[0, 0, 1], [0, 0, -1], [0, 1, 0], [1, 0, 0], [0, -1, 0], [-1, 0, 0], [0.1, 0.0995, 0.99],
                        [0.14, 0.0, 0.9901], [0.1410, 0.0, 0.98999], [0, 0, 0]], f32)
    nn[:len(special)] = special
    a = rng.uniform(-1, 1, (n, 3)).astype(f32)
    fs, ft, tw = (np.zeros((n, 3), f32) for _ in range(3))
    for i in range(n):
        L.ref_local_frame(P(nn[i]), P(fs[i]), P(ft[i]))
        L.ref_to_world(P(nn[i]), P(a[i]), P(tw[i]))
    g.update(lf_n=nn, lf_a=a, lf_s=fs, lf_t=ft, lf_to_world=tw)

    # Henyey-Greenstein
    hg_g = rng.uniform(-0.95, 0.95, n).astype(f32)
    hg_g[:4] = (0.0, 0.9, -0.9, 1e-6)
    hin, hout = unit(rng.normal(size=(n, 3))), unit(rng.normal(size=(n, 3)))
    hout[4:8] = hin[4:8]
    hout[8:12] = -hin[8:12]
    hv = np.array([L.ref_hg_eval(C.c_float(float(hg_g[i])), P(hin[i]), P(hout[i])) for i in range(n)], f32)
    g.update(hg_g=hg_g, hg_in=hin, hg_out=hout, hg_value=hv)

    # face normals and interpolated shading normals
    p = rng.uniform(-1, 1, (n, 3, 3)).astype(f32)
    p[:8, 2] = p[:8, 1]  # degenerate triangles: zero cross product, normalize() leaves it
    cn = (unit(rng.normal(size=(n, 3, 3))) * rng.uniform(0.2, 3, (n, 3, 1))).astype(f32)
    w = rng.dirichlet((1, 1, 1), n)
    pos = (w[:, :, None] * p.astype(np.float64)).sum(1).astype(f32)
    pos[8:40] += rng.normal(scale=0.05, size=(32, 3)).astype(f32)  # off-triangle points extrapolate
    fn, sn = np.zeros((n, 3), f32), np.zeros((n, 3), f32)
    for i in range(n):
        L.ref_face_normal(P(p[i, 0]), P(p[i, 1]), P(p[i, 2]), P(fn[i]))
        c = np.ascontiguousarray(cn[i])
        L.ref_interp_normal(P(p[i, 0]), P(p[i, 1]), P(p[i, 2]), P(c[0]), P(c[1]), P(c[2]), P(pos[i]), P(sn[i]))
    g.update(tri_p=p, tri_cn=cn, tri_pos=pos, tri_face_normal=fn, tri_shading_normal=sn)

    # rotation about an axis (refraction directions)
    ax = unit(rng.normal(size=(n, 3)))
    ang = rng.uniform(0, np.pi, n).astype(f32)
    rv = unit(rng.normal(size=(n, 3)))
    ro = np.zeros((n, 3), f32)
    for i in range(n):
        L.ref_rotate(P(ax[i]), C.c_float(float(ang[i])), P(rv[i]), P(ro[i]))
    g.update(rot_axis=ax, rot_angle=ang, rot_v=rv, rot_out=ro)

    # vector primitives incl. the zero cases
    vv = rng.uniform(-3, 3, (n, 3)).astype(f32)
    vv[:3] = 0
    vv[3] = (1e-30, 0, 0)
    vb = rng.uniform(-3, 3, (n, 3)).astype(f32)
    dv = rng.uniform(-2, 2, n).astype(f32)
    dv[:16] = 0
    nrm, dvo = np.zeros((n, 3), f32), np.zeros((n, 3), f32)
    dots, lens = np.zeros(n, f32), np.zeros(n, f32)
    for i in range(n):
        L.ref_normalize(P(vv[i]), P(nrm[i]))
        L.ref_div(P(vv[i]), C.c_float(float(dv[i])), P(dvo[i]))
        dots[i] = L.ref_dot(P(vv[i]), P(vb[i]))
        lens[i] = L.ref_length(P(vv[i]))
    g.update(vec_a=vv, vec_b=vb, vec_d=dv, vec_normalize=nrm, vec_div=dvo, vec_dot=dots, vec_length=lens)

    out = os.path.join(ROOT, "tests", "golden", "ref_math.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
