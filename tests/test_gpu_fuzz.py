"""Differential fuzz of the closest-hit path: random triangle soups (intersecting, sliver, axis-aligned,
duplicated and zero-area triangles; objects with and without vertex normals) against the CPU oracle, ids and t
bit for bit in all three modes. Rays: random, aimed at triangle vertices / edge midpoints / centroids, and
starting on triangles (as bounce rays do)."""
import numpy as np
import pytest

import meshgen

pytestmark = pytest.mark.gpu


def _soup(ra, rng, n_tri, n_obj, scale=1.0, offset=(0.0, 0.0, 0.0)):
    kind = rng.integers(0, 5, n_tri)
    c = rng.uniform(-0.9, 0.9, (n_tri, 1, 3))
    size = 10.0 ** rng.uniform(-2.5, -0.3, (n_tri, 1, 1))
    v = c + size * rng.normal(size=(n_tri, 3, 3))
    sl = kind == 1  # slivers: third corner almost on the first edge
    v[sl, 2] = v[sl, 0] + (v[sl, 1] - v[sl, 0]) * rng.uniform(0, 1, (sl.sum(), 1)) + 1e-4 * rng.normal(size=(sl.sum(), 3))
    ax = kind == 2  # axis-aligned: one coordinate shared by the three corners, on a coarse grid
    a = rng.integers(0, 3, n_tri)
    for k in np.nonzero(ax)[0]:
        v[k, :, a[k]] = np.round(v[k, 0, a[k]] * 8) / 8
    dg = kind == 3  # degenerate: two equal corners
    v[dg, 2] = v[dg, 1]
    dup = np.nonzero(kind == 4)[0]  # duplicates of another triangle
    v[dup] = v[rng.integers(0, n_tri, len(dup))]
    v = (np.clip(v, -1.0, 1.0) * scale + np.asarray(offset)).astype(np.float32)
    verts = v.reshape(-1, 3)
    tris = np.arange(3 * n_tri, dtype=np.uint32).reshape(n_tri, 3)
    tri_obj = np.sort(rng.integers(0, n_obj, n_tri)).astype(np.int32)
    types = [ra.OBJ_DIFFUSE, ra.OBJ_GLOSSY, ra.OBJ_REFRACTIVE]
    objects = np.array([meshgen._obj(types[k % 3], (0.7, 0.6, 0.5), coeff=5.0, ior=1.5, has_vn=int(k % 2)) for k in range(n_obj)],
                       ra.OBJECT_DTYPE)
    vn = rng.normal(size=(n_tri, 9)).astype(np.float32)
    return ra.HostScene.from_arrays(verts, tris, tri_obj, objects, vn), v


def _rays(rng, v, n, n_obj, ra, scale=1.0, offset=(0.0, 0.0, 0.0), tri_obj=None):
    T = len(v)
    o = rng.uniform(-1.2, 1.2, (n, 3)) * scale + np.asarray(offset)
    k = rng.integers(0, T, n)
    w = rng.dirichlet((1, 1, 1), n)
    pick = rng.integers(0, 4, n)
    w[pick == 0] = (1, 0, 0)            # at a corner
    w[pick == 1] = (0.5, 0.5, 0)        # at an edge midpoint
    target = (v[k].astype(np.float64) * w[:, :, None]).sum(1)
    d = target - o
    rnd = rng.random(n) < 0.25
    d[rnd] = rng.normal(size=(rnd.sum(), 3))
    on = rng.random(n) < 0.3            # start on a triangle, leave in a random direction
    k2 = rng.integers(0, T, n)
    w2 = rng.dirichlet((1, 1, 1), n)
    o[on] = (v[k2].astype(np.float64) * w2[:, :, None]).sum(1)[on]
    d[on] = rng.normal(size=(on.sum(), 3))
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)
    contact = rng.integers(-1, n_obj, n)
    if tri_obj is not None:  # half of the rays that start on a triangle know it as their contact object, as bounce rays do
        own = on & (rng.random(n) < 0.5)
        contact[own] = tri_obj[k2[own]]
    inside = np.where(rng.random(n) < 0.7, -1, rng.integers(-1, n_obj, n))
    return (np.ascontiguousarray(o.T, np.float32), np.ascontiguousarray(d.T, np.float32),
            ra.pack_ids(contact, inside), k2.astype(np.int32))


def _axis_rays(rng, v, o, d, scale):
    """scripts/fuzz_campaign.py AXIS=1: half of the rays replaced by rays that barely move along one or two axes (direction components of
    1e-8 .. 1e-2) and pass through a corner, an edge midpoint or an interior point of a triangle"""
    n = o.shape[1]
    sel = np.nonzero(rng.random(n) < 0.5)[0]
    m = len(sel)
    k = rng.integers(0, len(v), m)
    w = rng.dirichlet((1, 1, 1), m)
    pick = rng.integers(0, 4, m)
    w[pick == 0] = (1, 0, 0)
    w[pick == 1] = (0.5, 0.5, 0)
    target = (v[k].astype(np.float64) * w[:, :, None]).sum(1)
    dd = rng.normal(size=(m, 3))
    dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    small = rng.random((m, 3)) < 0.45
    small[small.all(1), 0] = False  # (at most two small components)
    dd = np.where(small, dd * 10.0 ** rng.uniform(-8, -2, (m, 3)), dd)
    dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    dist = rng.uniform(0.05, 3.0, (m, 1)) * scale
    o[:, sel] = (target - dd * dist).T.astype(np.float32)
    d[:, sel] = dd.T.astype(np.float32)


@pytest.mark.parametrize("seed,n_tri", [(1, 60), (2, 400), (3, 2500), (4, 12000), (5, 1000), (6, 5000)])
def test_triangle_soup_against_oracle(gpu, ra, orc, seed, n_tri):
    rng = np.random.default_rng(1000 + seed)
    n_obj = 5
    scene, v = _soup(ra, rng, n_tri, n_obj)
    S = orc.OracleScene(scene.raw())
    gpu.upload(scene.describe())
    n = 150000
    o, d, aux, last = _rays(rng, v, n, n_obj, ra)
    for mode, a in ((ra.MODE_UNCONDITIONAL, None), (ra.MODE_LAST_TID, last), (ra.MODE_NO_SELF, aux)):
        tg, ig = gpu.intersect(o, d, a, mode)
        tc, ic = S.intersect(o, d, a, mode, threads=16)
        diff = np.nonzero((ig != ic) | ~((tg == tc) | (np.isnan(tg) & np.isnan(tc))))[0]
        print("soup seed %d mode %d: %d tris, hit rate %.2f, %d of %d rays differ" % (seed, mode, n_tri, (ic >= 0).mean(), len(diff), n))
        # the tolerated divergences of DESIGN.md 4.1 only: the traversal prunes boxes that start beyond the closest
        # accepted t (with a slack), and the reference's own t = orthoDist / (n.d) is far off for a grazing hit or
        # for a sliver (its fp32 plane normal is inaccurate); and rays that lie in a triangle's plane
        assert len(diff) <= 2e-4 * n, (mode, len(diff), diff[:10], ig[diff[:10]], ic[diff[:10]])
        for q in diff:
            ok = False
            for t_id, t_rep in ((ig[q], tg[q]), (ic[q], tc[q])):
                if t_id < 0:
                    continue
                w = v[t_id].astype(np.float64)
                m = np.cross(w[1] - w[0], w[2] - w[1])
                nm = np.linalg.norm(m)
                if nm == 0:
                    ok = True
                    continue
                m /= nm
                dq, oq = d[:, q].astype(np.float64), o[:, q].astype(np.float64)
                nd = float(m @ dq)
                if abs(nd) < 1e-3:
                    ok = True  # grazing / coplanar
                    continue
                t_true = float(m @ (w[2] - oq)) / nd
                if abs(t_true - float(t_rep)) > 1e-4 * abs(t_true) + 1e-5:
                    ok = True  # the reference's t is off by more than the prune slack (sliver)
            assert ok, (mode, q, ig[q], ic[q], tg[q], tc[q])
    S.close()
    scene.close()


@pytest.mark.parametrize("seed,n_tri", [(11, 300), (12, 4000)])
def test_triangle_soup_visibility_against_oracle(gpu, ra, orc, seed, n_tri):
    """the any-hit query on the same soups: limits at, just before and just behind hits, and far beyond"""
    rng = np.random.default_rng(2000 + seed)
    n_obj = 5
    scene, v = _soup(ra, rng, n_tri, n_obj)
    S = orc.OracleScene(scene.raw())
    gpu.upload(scene.describe())
    n = 150000
    o, d, aux, _ = _rays(rng, v, n, n_obj, ra)
    tc, ic = S.intersect(o, d, aux, ra.MODE_NO_SELF, threads=16)
    lim = np.where(ic >= 0, tc, 1.0).astype(np.float32)
    scale = rng.choice(np.array([0.5, 1.0 - 1e-6, 1.0, 1.0 + 1e-6, 1.0 + 1e-4, 2.0, 100.0], np.float32), n)
    lim = (lim * scale).astype(np.float32)
    for mode, a in ((ra.MODE_NO_SELF, aux), (ra.MODE_UNCONDITIONAL, None)):
        vg = gpu.visibility(o, d, a, lim, mode)
        vc = S.visibility(o, d, a, lim, mode, threads=16)
        diff = np.nonzero(vg != vc)[0]
        print("soup visibility seed %d mode %d: visible %.2f, %d of %d differ" % (seed, mode, vc.mean(), len(diff), n))
        assert len(diff) <= 1e-4 * n, (mode, len(diff), diff[:10])
        # tolerated: an occluder whose own t is off by more than the prune slack (a sliver's fp32 plane normal, a
        # grazing hit) AND lies within that error of the limit, so that its box starts beyond the limit
        tq, iq = S.intersect(o[:, diff], d[:, diff], None if a is None else a[diff], mode, threads=4)
        for j, q in enumerate(diff):
            assert iq[j] >= 0 and abs(float(tq[j]) - float(lim[q])) <= 1e-3 * float(lim[q]), (mode, q, tq[j], lim[q])
            w = v[iq[j]].astype(np.float64)
            m = np.cross(w[1] - w[0], w[2] - w[1])
            m /= np.linalg.norm(m)
            nd = float(m @ d[:, q].astype(np.float64))
            t_true = float(m @ (w[2] - o[:, q].astype(np.float64))) / nd
            assert abs(nd) < 1e-3 or abs(t_true - float(tq[j])) > 1e-4 * abs(t_true) + 1e-5, (mode, q, t_true, tq[j])
    S.close()
    scene.close()


@pytest.mark.parametrize("seed", [8072, 8094, 8227, 8267, 8301, 8358])
def test_corner_rays_that_barely_move_along_an_axis(gpu, ra, orc, seed):
    """scripts/fuzz_campaign.py's soups of these seeds each hold a ray with one direction component of 1e-6 .. 1e-4
    aimed at a corner of a well-shaped triangle: the reference's edge signs accept the hit although the ray's line
    passes the corner ~1e-6 outside, which along that axis is 1e-2 .. 0.3 in t IN FRONT of every box that lists the
    triangle, and the exact kernels' prune (mode 0 takes them: the limit sits right behind the hit) skipped those
    boxes until it got exact_prune_slack (mirt_dev.h). Bit-equal flags in mode 0 now; the fast route (mode 2) always
    carried the ray's margin."""
    n = 150000
    rng = np.random.default_rng(seed)  # (the campaign's recipe, draw for draw)
    n_tri = int(10 ** rng.uniform(0.5, 4.3))
    scale = float(10 ** rng.uniform(-3, 3))
    offset = tuple((rng.uniform(-1, 1, 3) * scale * 10 ** rng.uniform(-1, 1.5)).tolist()) if rng.random() < 0.6 else (0.0, 0.0, 0.0)
    n_obj = int(rng.integers(1, 7)) if n_tri >= 40 else 1
    scene, v = _soup(ra, rng, n_tri, n_obj, scale, offset)
    S = orc.OracleScene(scene.raw())
    gpu.upload(scene.describe())
    o, d, aux, last = _rays(rng, v, n, n_obj, ra, scale, offset, scene.raw()["tri_obj"])
    for mode, a in ((0, None), (1, last), (2, aux)):  # (the campaign's draws in between: none, these are queries)
        tg, ig = gpu.intersect(o, d, a, mode)
        tc, ic = S.intersect(o, d, a, mode, threads=16)
        assert np.array_equal(ig, ic) and np.array_equal(tg, tc, equal_nan=True), mode
    tc, ic = S.intersect(o, d, aux, 2, threads=16)
    lim = np.where(ic >= 0, tc, scale).astype(np.float32)
    lim = (lim * rng.choice(np.array([0.5, 1.0 - 1e-6, 1.0, 1.0 + 1e-6, 1.0 + 1e-4, 2.0, 100.0], np.float32), n)).astype(np.float32)
    for mode, a in ((2, aux), (0, None)):
        vg = gpu.visibility(o, d, a, lim, mode)
        vc = S.visibility(o, d, a, lim, mode, threads=16)
        assert np.array_equal(vg, vc), (mode, np.nonzero(vg != vc)[0][:5])
    S.close()
    scene.close()


@pytest.mark.parametrize("seed,n_tri,scale,offset", [(21, 3000, 1.0, (0, 0, 0)), (22, 3000, 25.0, (300.0, -120.0, 75.0)),
                                                    (23, 800, 1e-3, (0.0, 0.0, 0.0))])
def test_soup_bounce_and_edge_parallel_rays(gpu, ra, orc, seed, n_tri, scale, offset):
    """(a) bounce-like rays (origin on a triangle, that triangle's object as the contact object) on soups that are
    scaled and moved away from the origin (the error model's margins scale with the coordinates); (b) rays running
    along a triangle's edge direction, tilted by 1e-7 .. 1e-2 rad and passing 1e-6 .. 1e-3 (in units of the scene)
    beside that edge: the reference's edge sign for that edge is then rounding noise over a reach that grows as
    1 / tilt, the one situation the widened leaf boxes do not bound. Ids must agree bit for bit in (a)
    and on all but a small share of (b), where every differing ray must be such an edge-parallel one by construction."""
    rng = np.random.default_rng(3000 + seed)
    n_obj = 5
    scene, v = _soup(ra, rng, n_tri, n_obj, scale, offset)
    tri_obj = scene.raw()["tri_obj"]
    S = orc.OracleScene(scene.raw())
    gpu.upload(scene.describe())
    n = 150000
    o, d, aux, last = _rays(rng, v, n, n_obj, ra, scale, offset, tri_obj)
    for mode, a in ((ra.MODE_UNCONDITIONAL, None), (ra.MODE_NO_SELF, aux)):
        tg, ig = gpu.intersect(o, d, a, mode)
        tc, ic = S.intersect(o, d, a, mode, threads=16)
        nd = int(((ig != ic) | ~((tg == tc) | (np.isnan(tg) & np.isnan(tc)))).sum())
        print("soup seed %d scale %g mode %d: hit rate %.2f, %d of %d rays differ" % (seed, scale, mode, (ic >= 0).mean(), nd, n))
        assert nd <= 2e-5 * n, (mode, nd)
    # (b)
    T = len(v)
    k = rng.integers(0, T, n)
    e = rng.integers(0, 3, n)
    va = v[k, e].astype(np.float64)
    vb = v[k, (e + 1) % 3].astype(np.float64)
    vc = v[k, (e + 2) % 3].astype(np.float64)
    ed = vb - va
    el = np.linalg.norm(ed, axis=1, keepdims=True)
    ok = el[:, 0] > 0
    ed = np.where(ok[:, None], ed / np.maximum(el, 1e-30), (1.0, 0.0, 0.0))
    out = va - vc
    out -= ed * (out * ed).sum(1, keepdims=True)  # in the triangle's plane, pointing away from the third corner
    out /= np.maximum(np.linalg.norm(out, axis=1, keepdims=True), 1e-30)
    nrm = np.cross(ed, out)
    side = scale * 10.0 ** rng.uniform(-6, -3, (n, 1)) * rng.choice([-1.0, 1.0], (n, 1))
    lift = scale * 10.0 ** rng.uniform(-6, -3, (n, 1)) * rng.choice([-1.0, 0.0, 1.0], (n, 1))
    tilt = 10.0 ** rng.uniform(-7, -2, (n, 1))
    dd = ed + tilt * (np.cos(rng.uniform(0, 6.28, (n, 1))) * out + np.sin(rng.uniform(0, 6.28, (n, 1))) * nrm)
    dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    oo = va + side * out + lift * nrm - dd * (scale * rng.uniform(0.2, 1.0, (n, 1)))
    o2, d2 = np.ascontiguousarray(oo.T, np.float32), np.ascontiguousarray(dd.T, np.float32)
    aux2 = ra.pack_ids(np.full(n, -1), np.full(n, -1))
    for mode, a in ((ra.MODE_UNCONDITIONAL, None), (ra.MODE_NO_SELF, aux2)):
        tg, ig = gpu.intersect(o2, d2, a, mode)
        tc, ic = S.intersect(o2, d2, a, mode, threads=16)
        nd = int((ig != ic).sum())
        print("soup seed %d scale %g mode %d, edge-parallel rays: %d of %d differ" % (seed, scale, mode, nd, n))
        assert nd <= 5e-4 * n, (mode, nd)
    S.close()
    scene.close()


@pytest.mark.parametrize("seed,n_tri,far", [(31, 2500, 6.0), (32, 2500, 40.0), (33, 800, 1000.0)])
def test_far_origin_rays_against_oracle(gpu, ra, orc, seed, n_tri, far):
    """Rays that start `far` scene sizes outside the scene, aimed at corners, edge midpoints and interiors of the soup's
    triangles (slivers included). The reach of the reference's edge-sign rounding beyond a triangle's outline grows
    with |v - o|, and the fast tree's single-triangle boxes are padded for origins within 4 x the scene's bound only
    (mirt.hip tri_fast_box): such rays must not take the fast route (RayPre::near) and the answers must stay the
    oracle's, bit for bit, in all three modes."""
    rng = np.random.default_rng(4000 + seed)
    n_obj = 5
    scene, v = _soup(ra, rng, n_tri, n_obj)
    S = orc.OracleScene(scene.raw())
    gpu.upload(scene.describe())
    n = 60000
    T = len(v)
    k = rng.integers(0, T, n)
    w = rng.dirichlet((1, 1, 1), n)
    pick = rng.integers(0, 4, n)
    w[pick == 0] = (1, 0, 0)
    w[pick == 1] = (0.5, 0.5, 0)
    target = (v[k].astype(np.float64) * w[:, :, None]).sum(1)
    target += (rng.random((n, 1)) < 0.3) * 1e-5 * rng.normal(size=(n, 3))  # some just beside the corner / edge
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    oo = u * far * rng.uniform(0.8, 1.2, (n, 1))
    dd = target - oo
    dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    o, d = np.ascontiguousarray(oo.T, np.float32), np.ascontiguousarray(dd.T, np.float32)
    aux = ra.pack_ids(rng.integers(-1, n_obj, n), np.full(n, -1))
    for mode, a in ((ra.MODE_UNCONDITIONAL, None), (ra.MODE_LAST_TID, k.astype(np.int32)), (ra.MODE_NO_SELF, aux)):
        tg, ig = gpu.intersect(o, d, a, mode)
        tc, ic = S.intersect(o, d, a, mode, threads=16)
        nd = int(((ig != ic) | ~((tg == tc) | (np.isnan(tg) & np.isnan(tc)))).sum())
        print("far origins x%g seed %d mode %d: hit rate %.2f, %d of %d rays differ" % (far, seed, mode, (ic >= 0).mean(), nd, n))
        assert nd == 0, (mode, nd)
    S.close()
    scene.close()


@pytest.mark.parametrize("seed,n_tri", [(41, 50), (42, 3000)])
def test_special_value_rays(gpu, ra, orc, seed, n_tri):
    """Rays nobody should send and somebody will: NaN, +-inf, +-0, denormal and 1e38 components in origins and
    directions, the zero direction, unnormalised directions. No fault, no hang, and the reference's answer (its
    arithmetic on those values, as the oracle computes it) in every mode, closest hit and visibility."""
    rng = np.random.default_rng(seed)
    n_obj = 4
    scene, v = _soup(ra, rng, n_tri, n_obj)
    S = orc.OracleScene(scene.raw())
    gpu.upload(scene.describe())
    n = 60000
    o, d, aux, last = _rays(rng, v, n, n_obj, ra, tri_obj=scene.raw()["tri_obj"])
    special = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-42, -1e-42, 1e38, -1e38, 1e-30, 3.0], np.float32)
    for arr, share in ((o, 0.3), (d, 0.5)):
        for c in range(3):
            hit = rng.random(n) < share / 2
            arr[c, hit] = rng.choice(special, int(hit.sum()))
    d[:, rng.random(n) < 0.02] = 0.0  # the zero direction
    with np.errstate(invalid="ignore", over="ignore"):
        for mode, a in ((0, None), (1, last), (2, aux)):
            tg, ig = gpu.intersect(o, d, a, mode)
            tc, ic = S.intersect(o, d, a, mode, threads=16)
            diff = np.nonzero((ig != ic) | ~((tg == tc) | (np.isnan(tg) & np.isnan(tc))))[0]
            # tolerated, as everywhere: a ray lying IN a triangle's plane (a denormal direction component
            # from an origin on an axis-aligned triangle's plane does that), where the reference's test reports t = 0
            # on triangles the ray does not touch and the pruned walk may not have visited the one it prefers
            left = []
            for q in diff:
                cop = False
                for tid in (ig[q], ic[q]):
                    if tid >= 0:
                        w = v[tid].astype(np.float64)
                        m = np.cross(w[1] - w[0], w[2] - w[1])
                        dq = d[:, q].astype(np.float64)
                        cop |= abs(float(m @ dq)) <= 1e-4 * np.linalg.norm(m) * np.linalg.norm(dq)
                if not (cop and ig[q] >= 0 and ic[q] >= 0):
                    left.append(q)
            assert len(diff) <= 1e-4 * n and not left, (mode, len(diff), [(int(q), int(ig[q]), float(tg[q]), int(ic[q]), float(tc[q]), o[:, q].tolist(), d[:, q].tolist()) for q in left[:4]])
        lim = rng.choice(np.array([0.0, 0.5, 2.0, 1e30, np.inf, np.nan, -1.0], np.float32), n)
        for mode, a in ((2, aux), (0, None), (1, last)):
            vg = gpu.visibility(o, d, a, lim, mode)
            vc = S.visibility(o, d, a, lim, mode, threads=16)
            diff = np.nonzero(vg != vc)[0]
            assert len(diff) == 0, (mode, len(diff), [(int(q), int(vg[q]), int(vc[q]), float(lim[q]), o[:, q].tolist(), d[:, q].tolist()) for q in diff[:4]])
    S.close()
    scene.close()


@pytest.mark.parametrize("value", [np.inf, -np.inf, 3e38, 1e20, 1e-42])
def test_special_value_scene(gpu, ra, orc, value):
    """A scene with corners at +-inf, near FLT_MAX, 1e20 or denormal: it leaves the classifier's error model, so upload
    builds no fast tree (its builder would bin by those values: a host crash before this test existed) and every query
    takes the exact kernels, whose arithmetic on such values is the reference's. (NaN corners are refused by the host
    surface, tests/test_host.py: the reference's tree build is undefined for them.)"""
    rng = np.random.default_rng(1)
    scene, v = _soup(ra, rng, 300, 3)
    raw = scene.raw()
    verts = raw["verts"].copy()
    verts[7, 1] = value
    verts[100, 0] = value
    s2 = ra.HostScene.from_arrays(verts, raw["tris"], raw["tri_obj"], raw["objects"], raw["tri_vn"])
    gpu.upload(s2.describe())
    S = orc.OracleScene(s2.raw())
    o, d, aux, last = _rays(rng, v, 50000, 3, ra, tri_obj=raw["tri_obj"])
    with np.errstate(all="ignore"):
        for mode, a in ((0, None), (1, last), (2, aux)):
            tg, ig = gpu.intersect(o, d, a, mode)
            tc, ic = S.intersect(o, d, a, mode, threads=16)
            assert np.array_equal(ig, ic) and np.array_equal(tg, tc, equal_nan=True), mode
        img, st = gpu.render_pt(ra.make_camera(32, 32, 30, position=(0, 0, 3.2)), 4, seed=1)
        assert int(st.n_rays) > 0
    S.close()
    s2.close()
    scene.close()


@pytest.mark.parametrize("seed,level", [(61, 2), (62, 4)])
def test_mesh_scene_with_near_axis_rays(gpu, ra, orc, seed, level):
    """A slice of scripts/fuzz_campaign.py MESH=1 AXIS=1 inside the suite: closed meshes inside Cornell walls (shared
    edges and corners: exact ties), half of the rays barely moving along one or two axes and passing through corners, edge
    midpoints and interiors: ids and t bit for bit in the three modes; visibility flags equal but for grazing hits."""
    rng = np.random.default_rng(seed)
    m = meshgen.cornell_with_meshes(level, seed, smooth=bool(seed & 1))
    scene = ra.HostScene.from_arrays(m["verts"], m["tris"], m["tri_obj"], m["objects"], m["tri_vn"])
    v = m["verts"][m["tris"].astype(np.int64)]
    n_obj = len(m["objects"])
    S = orc.OracleScene(scene.raw())
    gpu.upload(scene.describe())
    n = 100000
    o, d, aux, last = _rays(rng, v, n, n_obj, ra, tri_obj=m["tri_obj"])
    _axis_rays(rng, v, o, d, 1.0)
    for mode, a in ((0, None), (1, last), (2, aux)):
        tg, ig = gpu.intersect(o, d, a, mode)
        tc, ic = S.intersect(o, d, a, mode, threads=16)
        assert np.array_equal(ig, ic) and np.array_equal(tg, tc, equal_nan=True), (mode, int((ig != ic).sum()))
    tc, ic = S.intersect(o, d, aux, 2, threads=16)
    lim = np.where(ic >= 0, tc, 1.0).astype(np.float32)
    lim = (lim * rng.choice(np.array([0.5, 1.0 - 1e-6, 1.0, 1.0 + 1e-6, 1.0 + 1e-4, 2.0, 100.0], np.float32), n)).astype(np.float32)
    for mode, a in ((2, aux), (0, None), (1, last)):
        vg = gpu.visibility(o, d, a, lim, mode)
        vc = S.visibility(o, d, a, lim, mode, threads=16)
        diff = np.nonzero(vg != vc)[0]
        assert len(diff) <= 1e-4 * n
        tq, iq = S.intersect(o[:, diff], d[:, diff], None if a is None else a[diff], mode, threads=4)
        for j, q in enumerate(diff):  # (tolerated: an occluder met at a grazing angle, its reference t just before the limit)
            w = v[iq[j]].astype(np.float64)
            nn = np.cross(w[1] - w[0], w[2] - w[1])
            assert iq[j] >= 0 and abs(float(nn @ d[:, q])) <= 1e-3 * np.linalg.norm(nn), (mode, q)
    S.close()
    scene.close()
