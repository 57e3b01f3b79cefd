"""CPU tests of the oracle itself: hand-derived known answers and invariants. The reference holds
no fixtures for this path and cannot be built here, so these pins are ours
("parity unpinned")."""
import numpy as np
import pytest

import meshgen
from conftest import rmse


def test_box_known_answers(orc):
    lo, hi = (-1, -1, -1), (1, 1, 1)
    # from outside along +x: enters at x=-1 after distance 2
    assert orc.box_intersect(lo, hi, (-3, 0, 0), (1, 0, 0)) == 2.0
    # from inside: the reference returns the EXIT distance (planes behind have t < 0), KDTree.cpp:44
    assert orc.box_intersect(lo, hi, (0, 0, 0), (1, 0, 0)) == 1.0
    # pointing away: miss
    assert orc.box_intersect(lo, hi, (-3, 0, 0), (-1, 0, 0)) == -1.0
    # parallel to the slab and outside it: miss (zero components are skipped, KDTree.cpp:40)
    assert orc.box_intersect(lo, hi, (-3, 2, 0), (1, 0, 0)) == -1.0
    # flat box (an axis-aligned wall): hit through its own plane
    assert orc.box_intersect((-1, 0, -1), (1, 0, 1), (0.25, 2, 0.5), (0, -1, 0)) == 2.0
    # inclusive compares: a ray through the box corner edge still hits
    assert orc.box_intersect(lo, hi, (-3, 1, 1), (1, 0, 0)) == 2.0


def test_triangle_known_answers(orc):
    v0, v1, v2 = (0, 0, 0), (1, 0, 0), (0, 1, 0)
    assert orc.tri_intersect(v0, v1, v2, (0.25, 0.25, 1), (0, 0, -1)) == 1.0
    # two-sided (KDTree.cpp:280-281 accepts either winding)
    assert orc.tri_intersect(v0, v1, v2, (0.25, 0.25, -2), (0, 0, 1)) == 2.0
    # behind the origin
    assert orc.tri_intersect(v0, v1, v2, (0.25, 0.25, 1), (0, 0, 1)) == -1.0
    # outside the triangle
    assert orc.tri_intersect(v0, v1, v2, (0.75, 0.75, 1), (0, 0, -1)) == -1.0
    # parallel to the plane
    assert orc.tri_intersect(v0, v1, v2, (0.25, 0.25, 1), (1, 0, 0)) == -1.0
    # on an edge: inclusive sign tests accept it
    assert orc.tri_intersect(v0, v1, v2, (0.5, 0.0, 1), (0, 0, -1)) == 1.0
    # t == 0 is accepted (origin on the plane)
    assert orc.tri_intersect(v0, v1, v2, (0.25, 0.25, 0), (0, 0, -1)) == 0.0


def test_rng_reference_values(orc):
    def lowbias32(x):
        x &= 0xFFFFFFFF
        x ^= x >> 16
        x = (x * 0x7FEB352D) & 0xFFFFFFFF
        x ^= x >> 15
        x = (x * 0x846CA68B) & 0xFFFFFFFF
        x ^= x >> 16
        return x

    for seed, pixel, sample, dim in [(1, 0, 0, 0), (1, 12345, 7, 3), (99, 2**20 - 1, 4095, 40)]:
        h1 = lowbias32(sample + 0x9E3779B9 * (seed + 1))
        key = lowbias32(pixel ^ h1)
        assert orc.rng_draw(seed, pixel, sample, dim) == lowbias32(key + 0x9E3779B9 * dim)
    # uniformity of the 24-bit floats over dimensions and samples
    u = np.array([orc.rng_draw(1, 17, s, d) >> 8 for s in range(200) for d in range(20)]) / 2.0**24
    assert 0.0 <= u.min() and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 0.02 and abs(u.var() - 1 / 12) < 0.01


def test_camera_ray(orc, ra):
    cam = ra.make_camera(512, 512, 350)
    # Camera.cpp:15-22 at the image centre with u = v = 0: direction is exactly -z
    d = orc.camera_ray(cam, 256 * 512 + 256, 0.0, 0.0)
    assert np.allclose(d, (0, 0, -1))
    # pixel (0,0), u=v=0.5: right*(0.5-256) + up*(256-0.5) + front*350
    d = orc.camera_ray(cam, 0, 0.5, 0.5)
    e = np.array([-255.5, 255.5, -350.0])
    assert np.allclose(d, e / np.linalg.norm(e), atol=1e-6)


def test_hemi_cos_distribution(orc):
    # RandGenerator.cpp:152-168: +Y is the pole; cos(theta) = u2^(1/(k+1))
    rng = np.random.default_rng(3)
    for k in (1.0, 5.0):
        u = rng.random((4000, 2)).astype(np.float32)
        dirs = np.array([orc.hemi_cos(k, a, b) for a, b in u])
        assert np.allclose(np.linalg.norm(dirs, axis=1), 1, atol=1e-5)
        assert (dirs[:, 1] >= 0).all()
        # E[cos] for pdf ~ cos^k over the hemisphere = (k+1)/(k+2)
        assert abs(dirs[:, 1].mean() - (k + 1) / (k + 2)) < 0.01
    assert np.allclose(orc.hemi_cos(1.0, 0.0, 1.0), (0, 1, 0), atol=1e-6)


def test_tree_invariants(cornell_oracle):
    f = cornell_oracle.flatten()
    N, T = cornell_oracle.n_nodes, len(cornell_oracle.tris)
    leaf = f["node_right"] < 0
    assert (f["leaf_begin"][leaf] < f["leaf_end"][leaf]).all()
    # every triangle in exactly one leaf (KDTree.cpp:208-224), slots contiguous in pre-order
    assert sorted(f["slot_tid"].tolist()) == list(range(T))
    assert f["leaf_begin"][leaf][0] == 0 and f["leaf_end"][leaf][-1] == T
    assert (f["leaf_begin"][leaf][1:] == f["leaf_end"][leaf][:-1]).all()
    # node boxes are the tight bounds of their triangles
    v = cornell_oracle.verts[cornell_oracle.tris.astype(np.int64)]  # [T,3,3]
    for i in np.nonzero(leaf)[0][:200]:
        tv = v[f["slot_tid"][f["leaf_begin"][i]:f["leaf_end"][i]]].reshape(-1, 3)
        assert np.array_equal(tv.min(0), f["node_min"][i]) and np.array_equal(tv.max(0), f["node_max"][i])
    assert (f["node_right"][~leaf] > np.nonzero(~leaf)[0] + 1).all() and (f["node_right"] < N).all()


def _brute_force(verts, tris, o, d):
    """independent numpy closest hit over ALL triangles with the reference's triangle test order"""
    f32 = np.float32
    v = verts[tris.astype(np.int64)]  # [T,3,3]
    n = o.shape[1]
    best_t = np.full(n, -1, f32)
    best_id = np.full(n, -1, np.int32)

    def dot(a, b):
        r = (a[..., 0] * b[..., 0]).astype(f32)
        r = (r + (a[..., 1] * b[..., 1]).astype(f32)).astype(f32)
        return (r + (a[..., 2] * b[..., 2]).astype(f32)).astype(f32)

    def cross(l, r):
        return np.stack([(l[..., 1] * r[..., 2]).astype(f32) - (l[..., 2] * r[..., 1]).astype(f32),
                         (l[..., 2] * r[..., 0]).astype(f32) - (l[..., 0] * r[..., 2]).astype(f32),
                         (l[..., 0] * r[..., 1]).astype(f32) - (l[..., 1] * r[..., 0]).astype(f32)], -1).astype(f32)

    O = o.T[:, None, :]
    D = d.T[:, None, :]
    pos = np.ones((n, len(tris)), bool)
    neg = np.ones((n, len(tris)), bool)
    for i in range(3):
        line = (v[:, (i + 1) % 3] - v[:, i])[None]
        r2v = (v[:, i][None] - O).astype(f32)
        s = dot(D, cross(np.broadcast_to(line, r2v.shape), r2v))
        pos &= s >= 0
        neg &= s <= 0
    nrm = cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 1])
    ln = np.sqrt(dot(nrm, nrm)).astype(f32)
    nrm = np.where(ln[:, None] > 0, (nrm / np.where(ln > 0, ln, 1)[:, None]).astype(f32), nrm)
    od = dot(nrm[None], r2v)
    dd = dot(nrm[None], D)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = (od / dd).astype(f32)
    ok = (dd != 0) & ~(q < 0) & (pos | neg)
    q = np.where(ok, q, np.inf)
    best_id = q.argmin(1).astype(np.int32)
    best_t = q[np.arange(n), best_id]
    miss = ~np.isfinite(best_t)
    return np.where(miss, -1, best_t).astype(f32), np.where(miss, -1, best_id)


def test_tree_traversal_equals_brute_force(cornell_oracle):
    """The tree only prunes: away from box-boundary rounding the closest hit equals the closest
    over all triangles (checked with an independent numpy restatement of the triangle test)."""
    o, d, _ = meshgen.random_rays(1500, 11, 8)
    t, tid = cornell_oracle.intersect(o, d, None, 0, threads=4)
    bt, bid = _brute_force(cornell_oracle.verts, cornell_oracle.tris, o, d)
    same = tid == bid
    # differences must be ties (equal t) or box-rounding misses, and rare
    assert same.mean() > 0.995
    diff = ~same
    tie = np.isclose(t[diff], bt[diff], rtol=1e-5, atol=1e-6) | (tid[diff] < 0)
    assert tie.all()
    assert np.array_equal(t[same], bt[same])


def test_ordered_traversal_matches_reference_order(cornell_oracle, mesh20k_oracle):
    """The product's traversal order (near child first, distance pruning, explicit tie rule) must
    return what the reference's both-children order returns."""
    for orc_scene, nobj in ((cornell_oracle, 8), (mesh20k_oracle, 9)):
        o, d, aux = meshgen.random_rays(20000, 5, nobj, inside_obj=2 if nobj == 8 else 8)
        for mode, a in ((0, None), (2, aux)):
            t, tid = orc_scene.intersect(o, d, a, mode, threads=8)
            t2, tid2, nb, nt = orc_scene.intersect_ordered(o, d, a, mode)
            assert np.array_equal(tid, tid2) and np.array_equal(t, t2)
            rb, rt = orc_scene.ref_counts(o[:, :2000], d[:, :2000], None if a is None else a[:2000], mode)
            assert nb / 20000 < rb / 2000  # pruning visits fewer boxes than the reference order


def test_visibility_known_answers(cornell_oracle):
    """MCRenderer::testVisibility (MCRenderer.cpp:115-117): visible iff no legal hit, or the closest
    one is not nearer than the connected vertex by more than EPSILON. Worked from the Cornell
    geometry: floor y=-1, ceiling y=+1, camera-side open."""
    S = cornell_oracle
    o = np.array([[0.9, 0.0, 0.9]] * 6, np.float32).T.copy()
    d = np.array([[0, -1, 0]] * 5 + [[0, 0, 1]], np.float32).T.copy()
    # the floor lies at distance 1 below (0.9, 0, 0.9)
    lim = np.array([0.5, 1.0, 1.00002, 1.000009, 2.0, 5.0], np.float32)
    vis = S.visibility(o, d, None, lim, 0, threads=1)
    assert vis.tolist() == [1, 1, 0, 1, 0, 1]  # last ray leaves through the open front: always visible
    # consistency with the closest hit on random rays, every mode
    o, d, aux = meshgen.random_rays(20000, 17, 8, inside_obj=2)
    for mode, a in ((0, None), (2, aux)):
        t, _ = S.intersect(o, d, a, mode, threads=8)
        lim = (np.where(t >= 0, t, 1) * np.random.default_rng(mode).choice([0.5, 1.0, 1.5], len(t))).astype(np.float32)
        vis = S.visibility(o, d, a, lim, mode, threads=8)
        assert np.array_equal(vis, ((t < 0) | (t - lim >= np.float32(-1e-5))).astype(np.uint8))


def test_predicate_cases(cornell_oracle):
    """NoSelfIntersectionCondition.cpp:12-27 on the floor (object 6, normal +y)."""
    S = cornell_oracle
    tid = int(np.nonzero(S.tri_obj == 6)[0][0])
    down, up = (0, -1, 0), (0, 1, 0)
    o_above, o_below = (0.1, -0.5, 0.1), (0.1, -1.5, 0.1)
    # no contact, not inside anything: front hit from above is legal, back hit from below is not
    assert S.legal(tid, o_above, down, 0.5, -1, -1) is True
    assert S.legal(tid, o_below, up, 0.5, -1, -1) is True  # inside == contact == NULL: only "in && contact==hit"
    # contact = some other object, inside NULL: back faces of everything are ignored
    assert S.legal(tid, o_below, up, 0.5, 3, -1) is False
    assert S.legal(tid, o_above, down, 0.5, 3, -1) is True
    # leaving the floor itself (contact = floor): re-hitting its front side from above is legal
    assert S.legal(tid, o_above, down, 0.5, 6, -1) is True
    # inside == contact == floor: an "in" hit on the floor is rejected
    assert S.legal(tid, o_above, down, 0.5, 6, 6) is False
    assert S.legal(tid, o_below, up, 0.5, 6, 6) is True


def test_world_normals(cornell_oracle):
    S = cornell_oracle
    # the left wall uses a reflecting transform; its shading normal must point into the box (+x)
    tid = int(np.nonzero(S.tri_obj == 3)[0][0])
    assert np.allclose(S.world_normal(tid, (-1, 0, 0)), (1, 0, 0), atol=1e-6)
    # light faces down
    assert np.allclose(S.world_normal(0, (0, 0.99, 0)), (0, -1, 0), atol=1e-6)
    # mirror ball: smooth normal at a vertex equals the radial direction
    c = np.array([-0.5, -0.6, -0.3])
    tid = int(np.nonzero(S.tri_obj == 1)[0][100])
    p = S.verts[S.tris[tid][0]]
    n = S.world_normal(tid, p)
    assert np.allclose(n, (p - c) / np.linalg.norm(p - c), atol=1e-4)


def test_path_semantics(cornell_oracle, cornell):
    """samplePath bookkeeping: a path that scores ends on the emitter, camera vertex first."""
    cam = cornell.camera()
    scored = 0
    for pixel in range(0, 128 * 128, 97):
        verts, rgb, n = cornell_oracle.trace_path(cam, pixel, 0)
        assert 2 <= n <= 23  # camera vertex + <= maxDepth+1 scatters + terminator
        assert np.allclose(verts[0][:3], (0, 0, 2)) and verts[0][9] == 1.0
        if rgb.sum() > 0:
            scored += 1
            last = verts[n - 1]
            assert np.allclose(last[3:6], 0) and np.allclose(last[6:9], 50)  # emission, terminated
            assert abs(last[1] - 0.99) < 1e-5  # on the light plane
    assert scored > 0


def test_render_statistics(cornell_oracle, cornell):
    """independent seeds agree in the mean; the same seed reproduces bit for bit; shards add up"""
    cam = cornell.camera()
    a, ra_ = cornell_oracle.render(cam, 8, seed=1, threads=8)
    b, _ = cornell_oracle.render(cam, 8, seed=1, threads=3)
    assert np.array_equal(a, b)
    c, _ = cornell_oracle.render(cam, 8, seed=2, threads=8)
    assert abs(a.mean() - c.mean()) / a.mean() < 0.1
    lo, _ = cornell_oracle.render(cam, 3, spp_begin=0, seed=1, threads=8)
    hi, _ = cornell_oracle.render(cam, 5, spp_begin=3, seed=1, threads=8)
    assert np.allclose(lo + hi, a, rtol=1e-5, atol=1e-6)
    assert 2.0 < ra_ / (128 * 128 * 8) < 5.0  # a few rays per sample (~3.14)


def test_furnace_energy(orc, ra):
    """Known answer for the integrator: inside a closed diffuse box lit by one emissive wall the
    mean radiance seen is bounded by emission * albedo series; with a black box only directly
    visible emitter pixels are non-zero and equal the emission exactly (after vignetting)."""
    m = meshgen.cornell_with_meshes(1)
    objs = m["objects"].copy()
    objs["color"][1:] = 0.0  # everything but the light is black
    objs["type"][6:] = ra.OBJ_DIFFUSE
    raw = dict(verts=m["verts"], tris=m["tris"], tri_obj=m["tri_obj"], tri_local=np.zeros(len(m["tris"]), np.int32),
               tri_vn=m["tri_vn"], objects=objs)
    S = orc.OracleScene(raw)
    cam = ra.make_camera(64, 64, 40, position=(0, 0.5, 0.9), focus=(0, 0.99, 0), up=(0, 0, -1))
    img, _ = S.render(cam, 4, seed=1, threads=8)
    img /= 4
    centre = img[32, 32]
    # straight at the light: contribution = cos0 * emission / vignette(centre) with cos0 ~ 1
    assert np.allclose(centre, 50, rtol=2e-3)
    assert img.max() <= 50 * 1.01  # pixel jitter moves |front.dir| against the centre value
    S.close()


def test_find_inside_object(orc, ra):
    """Scene::findInsideObject (Scene.cpp:354-390): inside the medium's cube in any direction, not
    outside it, not when the medium is the object to skip."""
    s = meshgen.fog_box_scene((0.3, 0.3, 0.3), (0.2, 0.2, 0.2))
    S = orc.OracleScene(s.raw())
    rng = np.random.default_rng(3)
    for _ in range(50):
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        assert S.find_inside(rng.uniform(-1.4, 1.4, 3), d) == 0
        assert S.find_inside(rng.uniform(-1.4, 1.4, 3), d, current=0) == -1
        o = rng.uniform(1.6, 3, 3) * rng.choice([-1, 1], 3)
        assert S.find_inside(o, d) == -1
    S.close()
    s.close()


def test_medium_known_answers(orc, ra):
    """SceneVPMObject.cpp:4-233 against closed forms, camera inside the medium (IOR 1, so the boundary
    does nothing), looking straight at an emitter 2 units away:
      * purely absorbing grey medium: every path either reaches the emitter with weight
        T(d)/P_surface(d) = 1 or dies at a medium event, so the pixel estimates L * exp(-sigma_a d);
      * no medium coefficients at all reduces to the vacuum answer L (sampDist = inf)."""
    cam = ra.make_camera(32, 32, 400, position=(0, 0, 1), focus=(0, 0, -1))
    for sa, expect in ((0.5, 50 * np.exp(-0.5 * 2.0)), (1e-6, 50.0)):
        s = meshgen.fog_box_scene((0, 0, 0), (sa, sa, sa))
        S = orc.OracleScene(s.raw())
        spp = 256
        img, _ = S.render(cam, spp, seed=2, threads=8)
        got = (img / spp)[8:24, 8:24].mean()
        assert abs(got - expect) / expect < 0.02, (sa, got, expect)
        S.close()
        s.close()
    # a scattering medium: single scattering adds light, absorption removes it; the directly seen
    # emitter is attenuated by exp(-sigma_t d) and in-scattering can only add to that
    s = meshgen.fog_box_scene((0.4, 0.4, 0.4), (0.1, 0.1, 0.1))
    S = orc.OracleScene(s.raw())
    img, _ = S.render(cam, 256, seed=2, threads=8)
    got = (img / 256)[8:24, 8:24].mean()
    assert 50 * np.exp(-0.5 * 2.0) * 0.97 < got < 50.0, got
    S.close()
    s.close()


def test_pttest_in_medium_known_answer(orc, ra):
    """PathTracerTest.cpp:34-74 inside a purely absorbing grey medium (camera inside, IOR 1, emitter 2 units ahead): a
    sample either dies at a medium event (its connection has BSDF ds * phase = 0) or reaches the emitter as vertex 1,
    which is scored with throughput 1 - the segment's decay is applied after the vertex is scored (:70-74) and
    originProb is never divided out - so every surviving sample is exactly L and the pixel estimates
    L * P(no event before d) = L exp(-sigma d): the path tracer's answer, by another route."""
    cam = ra.make_camera(32, 32, 400, position=(0, 0, 1), focus=(0, 0, -1))
    s = meshgen.fog_box_scene((0, 0, 0), (0.5, 0.5, 0.5))
    S = orc.OracleScene(s.raw())
    spp = 256
    img, nr, ns = S.render_pttest(cam, spp, seed=2, threads=8)
    block = (img / spp)[8:24, 8:24]
    assert abs(block.mean() - 50 * np.exp(-1.0)) / (50 * np.exp(-1.0)) < 0.03, block.mean()
    one, _, _ = S.render_pttest(cam, 1, seed=2, threads=8)
    vals = np.unique(np.round(one[8:24, 8:24], 3))
    assert set(vals.tolist()) <= {0.0, 50.0}, vals  # a sample is all or nothing
    assert ns > 0  # the terminated medium vertices are RANDOM (SceneVPMObject.cpp:132): each one attempts a connection
    S.close()
    s.close()


def test_pttest_converges_to_path_tracer(cornell_oracle, cornell):
    """PathTracerTest is the reference's cross-check of PathTracer: same expectation, less noise.
    64x64 view of the Cornell scene; block means compared."""
    import renderer_amd as ra

    cam = ra.make_camera(64, 64, 44)
    a, _ = cornell_oracle.render(cam, 192, seed=11, threads=8)
    b, nr, ns = cornell_oracle.render_pttest(cam, 48, seed=12, threads=8)
    a, b = np.clip(a / 192, 0, 1), np.clip(b / 48, 0, 1)
    assert ns > 0 and nr > 0
    assert np.abs(a.mean((0, 1)) - b.mean((0, 1))).max() < 0.02
    blk = lambda x: x.reshape(8, 8, 8, 8, 3).mean((1, 3))
    assert np.abs(blk(a) - blk(b)).max() < 0.15


def test_pttest_direct_light_known_answer(orc, ra):
    """A diffuse floor under a square emitter, nothing else, maxDepth 1: the only light a floor pixel
    receives is the connection at its first vertex plus (MIS-weighted) emitter hits at depth 2, which
    the cap removes... so with maxDepth 1 the floor shows exactly the direct light E * rho/pi * G,
    integrated over the emitter: compare with a numerical quadrature."""
    import meshgen

    m = meshgen.floor_and_light(albedo=0.6, emission=5.0)
    s = ra.HostScene.from_arrays(m["verts"], m["tris"], m["tri_obj"], m["objects"], m["tri_vn"])
    S = orc.OracleScene(s.raw())
    cam = ra.make_camera(16, 16, 16, position=(0, 0.5, 2.5), focus=(0, -1, 0))
    img, _, _ = S.render_pttest(cam, 4000, max_depth=1, seed=3, threads=8)
    img /= 4000
    # centre pixel looks at the floor point where the view axis meets y = -1
    t = (-1 - 0.5) / (-1.5 / np.linalg.norm((0, -1.5, -2.5)))
    x = np.array([0, 0.5, 2.5]) + t * np.array([0, -1.5, -2.5]) / np.linalg.norm((0, -1.5, -2.5))
    # quadrature over the emitter square (y = 1, |x|,|z| <= 0.5, facing down)
    g = (np.arange(200) + 0.5) / 200 - 0.5
    yx, yz = np.meshgrid(g, g)
    dvec = np.stack([yx - x[0], np.full_like(yx, 1 - x[1]), yz - x[2]], -1)
    r2 = (dvec ** 2).sum(-1)
    cos_f = dvec[..., 1] / np.sqrt(r2)
    cos_l = dvec[..., 1] / np.sqrt(r2)
    expect = 5.0 * 0.6 / np.pi * (cos_f * cos_l / r2).mean() * 1.0
    got = img[7:9, 7:9].mean()
    assert abs(got - expect) / expect < 0.05, (got, expect)
    S.close()
    s.close()


def test_phong_known_answers(orc, ra):
    """ScenePhongObject.h:7-23 + SceneObject.cpp:227-248, 341-373. A Phong floor whose glossy component is
    black: a scatter picks the (dead) glossy material half the time and the choice is not divided out, so
    the path tracer sees HALF the radiance of the plain diffuse floor; the light connection of PTtest
    evaluates the SUM of the BSDFs, so its direct light (maxDepth 1) equals the diffuse floor's."""
    import meshgen

    cam = ra.make_camera(16, 16, 16, position=(0, 0.5, 2.5), focus=(0, -1, 0))
    imgs = {}
    for name, floor in (("diffuse", None),
                        ("phong", meshgen._obj(ra.OBJ_PHONG, (0.6, 0.6, 0.6), coeff=5.0, color2=(0, 0, 0)))):
        m = meshgen.floor_and_light(albedo=0.6, emission=5.0, floor=floor)
        s = ra.HostScene.from_arrays(m["verts"], m["tris"], m["tri_obj"], m["objects"], m["tri_vn"])
        S = orc.OracleScene(s.raw())
        pt, _ = S.render(cam, 6000, max_depth=3, seed=3, threads=8)
        nee, _, _ = S.render_pttest(cam, 2000, max_depth=1, seed=3, threads=8)
        imgs[name] = (pt[6:10, 6:10].mean() / 6000, nee[6:10, 6:10].mean() / 2000)
        S.close()
        s.close()
    assert abs(imgs["phong"][0] / imgs["diffuse"][0] - 0.5) < 0.06, imgs
    assert abs(imgs["phong"][1] / imgs["diffuse"][1] - 1.0) < 0.05, imgs


def test_fast_tree_and_its_restatement(cornell, cornell_oracle, mesh20k, mesh20k_oracle):
    """mirt_fast_tree (host-only): a binary hierarchy in which every triangle slot of the scene tree appears
    exactly once, as a leaf of its own with the (slightly widened) box of its three corners, and every node box is the union of its
    children's; the oracle's restatement of the fast path over it (conservative traversal + reference-chain
    check + exact re-run) returns exactly the reference closest hit, with fewer box tests than the reference
    tree needs."""
    for scene, S, nobj, inside in ((cornell, cornell_oracle, 8, 2), (mesh20k, mesh20k_oracle, 9, 8)):
        ref, ft = scene.tree(), scene.fast_tree()
        right, lb, le = ft["node_right"], ft["leaf_begin"], ft["leaf_end"]
        leaf = right < 0
        assert sorted(lb[leaf].tolist()) == list(range(scene.n_tris)) and np.array_equal(le[leaf], lb[leaf] + 1)
        assert len(right) == 2 * scene.n_tris - 1
        raw = scene.raw()
        pos = raw["verts"][raw["tris"].astype(np.int64)]  # [T, 3 corners, xyz], by global triangle id
        # a leaf's box: the triangle's corners, widened by the reach of the reference's edge-sign test (a hair for a
        # well-shaped triangle), never beyond the box of the reference leaf that holds the slot
        ref_leaf = np.nonzero(ref["node_right"] < 0)[0]
        leaf_of = np.empty(scene.n_tris, np.int64)
        for j in ref_leaf:
            leaf_of[ref["leaf_begin"][j]:ref["leaf_end"][j]] = j
        for i in np.nonzero(leaf)[0][:200]:
            p = pos[ref["slot_tid"][lb[i]]]
            j = leaf_of[lb[i]]
            assert (ft["node_min"][i] <= p.min(0)).all() and (ft["node_max"][i] >= p.max(0)).all()
            assert (ft["node_min"][i] >= ref["node_min"][j]).all() and (ft["node_max"][i] <= ref["node_max"][j]).all()
            assert (p.min(0) - ft["node_min"][i]).max() < 1e-3 and (ft["node_max"][i] - p.max(0)).max() < 1e-3
        inner = np.nonzero(~leaf)[0]
        l, r = inner + 1, right[inner]
        assert (r > l).all() and (r < len(right)).all()
        assert np.array_equal(ft["node_min"][inner], np.minimum(ft["node_min"][l], ft["node_min"][r]))
        assert np.array_equal(ft["node_max"][inner], np.maximum(ft["node_max"][l], ft["node_max"][r]))
        assert np.array_equal(ft["node_min"][0], ref["node_min"][0]) and np.array_equal(ft["node_max"][0], ref["node_max"][0])
        o, d, aux = meshgen.random_rays(20000, 15, nobj, inside_obj=inside)
        d[:, :2000] *= np.array([[1.0], [1e-9], [1.0]], np.float32)  # nearly axis-parallel
        d[1, 2000:3000] = 0
        for mode, a in ((0, None), (2, aux)):
            t, tid = S.intersect(o, d, a, mode, threads=8)
            t2, tid2, nb, nt = S.intersect_fast(ft, o, d, a, mode)
            assert np.array_equal(tid, tid2) and np.array_equal(t, t2)
            rb, rt = S.ref_counts(o[:, :2000], d[:, :2000], None if a is None else a[:2000], mode)
            assert nb / 20000 < 0.8 * rb / 2000
