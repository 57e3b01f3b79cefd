"""Procedural scenes for the large-mesh benchmark and the tests: displaced
icospheres with diffuse / glossy / refractive materials inside Cornell walls. Nothing is committed;
meshes are generated from a seed at run time."""
import numpy as np

import renderer_amd as ra


def icosphere(level):
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], np.int64)
    for _ in range(level):
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        e.sort(axis=1)
        key = e[:, 0] * (len(v) + 1) + e[:, 1]
        uniq, inv = np.unique(key, return_inverse=True)
        a, b = uniq // (len(v) + 1), uniq % (len(v) + 1)
        mid = v[a] + v[b]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        m = inv.reshape(3, -1) + len(v)  # midpoints of edges 01, 12, 20 per face
        v = np.concatenate([v, mid])
        f = np.concatenate([np.stack([f[:, 0], m[0], m[2]], 1), np.stack([f[:, 1], m[1], m[0]], 1),
                            np.stack([f[:, 2], m[2], m[1]], 1), np.stack([m[0], m[1], m[2]], 1)])
    return v, f


def _fbm(p, seed, octaves=4):
    rng = np.random.default_rng(seed)
    out = np.zeros(len(p))
    amp, freq = 0.5, 2.0
    for _ in range(octaves):
        w = rng.normal(size=(3, 3)) * freq
        ph = rng.uniform(0, 2 * np.pi, 3)
        out += amp * np.sin(p @ w[0] + ph[0]) * np.sin(p @ w[1] + ph[1]) * np.sin(p @ w[2] + ph[2])
        amp *= 0.5
        freq *= 2.0
    return out


def displaced_sphere(level, seed, radius, center, amp=0.15):
    v, f = icosphere(level)
    r = 1.0 + amp * _fbm(v, seed)
    v = v * r[:, None] * radius + np.asarray(center, np.float64)
    return v, f


def vertex_normals(v, f):
    fn = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    vn = np.zeros_like(v)
    for k in range(3):
        np.add.at(vn, f[:, k], fn)
    vn /= np.maximum(np.linalg.norm(vn, axis=1, keepdims=True), 1e-30)
    return vn


def _quad(p0, p1, p2, p3, normal):
    """two triangles whose face normal (p1-p0)x(p2-p0) equals `normal`'s direction"""
    v = np.array([p0, p1, p2, p3], np.float64)
    f = np.array([[0, 1, 2], [0, 2, 3]], np.int64)
    n = np.cross(v[1] - v[0], v[2] - v[0])
    if np.dot(n, normal) < 0:
        f = f[:, [0, 2, 1]]
    return v, f, np.tile(np.asarray(normal, np.float64), (4, 1))


def _obj(kind, color=(1, 1, 1), coeff=0.0, ior=1.0, decay=(1, 1, 1), has_vn=1, color2=(1, 1, 1)):
    o = np.zeros(1, ra.OBJECT_DTYPE)[0]
    o["color2"] = color2
    o["type"] = kind
    o["has_vn"] = has_vn
    o["color"] = color
    o["coeff"] = coeff
    o["ior"] = ior
    o["decay"] = decay
    return o


def cornell_with_meshes(level, seed=7, smooth=True):
    """Light + 5 walls + three displaced spheres (diffuse, glossy 5, glass 1.5).
    Returns dict(verts, tris, tri_obj, tri_vn, objects) ready for HostScene.from_arrays."""
    parts = []  # (verts, faces, vertex normals, object)
    a = 0.2
    parts.append(_quad((-a, 0.99, -a), (a, 0.99, -a), (a, 0.99, a), (-a, 0.99, a), (0, -1, 0)) +
                 (_obj(ra.OBJ_EMISSIVE, (50, 50, 50)),))
    parts.append(_quad((-1, -1, -1), (-1, -1, 1), (-1, 1, 1), (-1, 1, -1), (1, 0, 0)) +
                 (_obj(ra.OBJ_DIFFUSE, (0.75, 0.25, 0.25)),))
    parts.append(_quad((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1), (-1, 0, 0)) +
                 (_obj(ra.OBJ_DIFFUSE, (0.25, 0.25, 0.75)),))
    parts.append(_quad((-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1), (0, -1, 0)) +
                 (_obj(ra.OBJ_DIFFUSE, (0.5, 0.5, 0.5)),))
    parts.append(_quad((-1, -1, -1), (1, -1, -1), (1, -1, 1), (-1, -1, 1), (0, 1, 0)) +
                 (_obj(ra.OBJ_DIFFUSE, (0.5, 0.5, 0.5)),))
    parts.append(_quad((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (0, 0, 1)) +
                 (_obj(ra.OBJ_DIFFUSE, (0.25, 0.75, 0.25)),))
    mats = [_obj(ra.OBJ_DIFFUSE, (0.7, 0.7, 0.3)), _obj(ra.OBJ_GLOSSY, (0.75, 0.75, 0.75), coeff=5.0),
            _obj(ra.OBJ_REFRACTIVE, (1, 1, 1), ior=1.5)]
    centers = [(-0.5, -0.55, -0.3), (0.0, 0.1, -0.45), (0.5, -0.55, 0.2)]
    for k in range(3):
        v, f = displaced_sphere(level, seed + k, 0.36, centers[k])
        vn = vertex_normals(v, f)
        parts.append((v, f, vn, mats[k]))
    verts, tris, tri_obj, tri_vn, objects = [], [], [], [], []
    off = 0
    for oi, (v, f, vn, ob) in enumerate(parts):
        verts.append(v)
        tris.append(f + off)
        tri_obj.append(np.full(len(f), oi, np.int32))
        tri_vn.append(vn[f].reshape(len(f), 9))
        ob = ob.copy()
        ob["has_vn"] = 1 if smooth else 0
        objects.append(ob)
        off += len(v)
    return dict(verts=np.concatenate(verts).astype(np.float32), tris=np.concatenate(tris).astype(np.uint32),
                tri_obj=np.concatenate(tri_obj), tri_vn=np.concatenate(tri_vn).astype(np.float32),
                objects=np.array(objects, ra.OBJECT_DTYPE))


def _assemble(parts):
    verts, tris, tri_obj, tri_vn, objects = [], [], [], [], []
    off = 0
    for oi, (v, f, vn, ob) in enumerate(parts):
        verts.append(v)
        tris.append(f + off)
        tri_obj.append(np.full(len(f), oi, np.int32))
        tri_vn.append(vn[f].reshape(len(f), 9))
        objects.append(ob.copy())
        off += len(v)
    return dict(verts=np.concatenate(verts).astype(np.float32), tris=np.concatenate(tris).astype(np.uint32),
                tri_obj=np.concatenate(tri_obj), tri_vn=np.concatenate(tri_vn).astype(np.float32),
                objects=np.array(objects, ra.OBJECT_DTYPE))


def floor_and_light(albedo=0.6, emission=5.0, floor=None):
    """a large diffuse floor at y = -1 under a unit-square emitter at y = 1 facing down; nothing else"""
    parts = [_quad((-0.5, 1, -0.5), (0.5, 1, -0.5), (0.5, 1, 0.5), (-0.5, 1, 0.5), (0, -1, 0)) +
             (_obj(ra.OBJ_EMISSIVE, (emission,) * 3),),
             _quad((-20, -1, -20), (20, -1, -20), (20, -1, 20), (-20, -1, 20), (0, 1, 0)) +
             (floor if floor is not None else _obj(ra.OBJ_DIFFUSE, (albedo,) * 3),)]
    return _assemble(parts)


def mesh_scene(level, seed=7, smooth=True):
    m = cornell_with_meshes(level, seed, smooth)
    return ra.HostScene.from_arrays(m["verts"], m["tris"], m["tri_obj"], m["objects"],
                                    m["tri_vn"] if smooth else None)


def fan_scene(n_fan=300, with_wall=True):
    """A fan of n_fan triangles around one apex that is the minimum corner of the fan: every median
    split sends all of them left (a triangle goes left if ANY vertex is <= the split), so the
    reference's build leaves them in ONE leaf (KDTree.cpp:199-203)."""
    ang = np.linspace(0.0, 0.5 * np.pi, n_fan + 1)
    ring = np.stack([0.1 + 0.8 * np.cos(ang), 0.1 + 0.8 * np.sin(ang), 0.3 + 0.2 * np.sin(3 * ang)], axis=1)
    v = np.concatenate([np.array([[-0.9, -0.9, -0.9]]), ring])
    f = np.stack([np.zeros(n_fan, np.int64), np.arange(1, n_fan + 1), np.arange(2, n_fan + 2)], axis=1)
    parts = [(v, f, _obj(ra.OBJ_DIFFUSE, (0.7, 0.7, 0.3), has_vn=0))]
    if with_wall:
        qv, qf, _ = _quad((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1), (-1, 0, 0))
        parts.append((qv, qf, _obj(ra.OBJ_DIFFUSE, (0.5, 0.5, 0.5), has_vn=0)))
    verts, tris, tri_obj, objects, off = [], [], [], [], 0
    for oi, (pv, pf, ob) in enumerate(parts):
        verts.append(pv)
        tris.append(pf + off)
        tri_obj.append(np.full(len(pf), oi, np.int32))
        objects.append(ob)
        off += len(pv)
    return ra.HostScene.from_arrays(np.concatenate(verts).astype(np.float32), np.concatenate(tris).astype(np.uint32),
                                    np.concatenate(tri_obj), np.array(objects, ra.OBJECT_DTYPE), None)


def flat_fan_scene(n_fan=300):
    """A fan in the plane z = 0.25 around an apex that is its minimum corner (one leaf of n_fan triangles,
    like fan_scene), with ring vertices k/512 on a straight line: every coordinate and every product in
    the edge tests of a ray aimed at the middle of a spoke is exact, so such a ray hits BOTH triangles
    sharing the spoke at the same t and the tie rule decides (KDTree.cpp:305: the earlier triangle)."""
    k = np.arange(n_fan + 1, dtype=np.float64)
    ring = np.stack([k / 512.0, (n_fan - k) / 512.0, np.full(n_fan + 1, 0.25)], axis=1)
    v = np.concatenate([np.array([[0.0, 0.0, 0.25]]), ring])
    f = np.stack([np.zeros(n_fan, np.int64), np.arange(1, n_fan + 1), np.arange(2, n_fan + 2)], axis=1)
    objects = np.array([_obj(ra.OBJ_DIFFUSE, (0.7, 0.7, 0.3), has_vn=0)], ra.OBJECT_DTYPE)
    return ra.HostScene.from_arrays(v.astype(np.float32), f.astype(np.uint32), np.zeros(n_fan, np.int32), objects, None)


def _box(h, center=(0, 0, 0)):
    """axis-aligned cube of half-size h: 12 triangles, outward face normals"""
    cx, cy, cz = center
    c = lambda x, y, z: (cx + h * x, cy + h * y, cz + h * z)
    faces = [((-1, -1, -1), (-1, -1, 1), (-1, 1, 1), (-1, 1, -1), (-1, 0, 0)),
             ((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1), (1, 0, 0)),
             ((-1, -1, -1), (1, -1, -1), (1, -1, 1), (-1, -1, 1), (0, -1, 0)),
             ((-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1), (0, 1, 0)),
             ((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (0, 0, -1)),
             ((-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1), (0, 0, 1))]
    vs, fs, off = [], [], 0
    for p0, p1, p2, p3, n in faces:
        v, f, _ = _quad(c(*p0), c(*p1), c(*p2), c(*p3), n)
        vs.append(v)
        fs.append(f + off)
        off += 4
    return np.concatenate(vs), np.concatenate(fs)


def fog_box_scene(sigma_s, sigma_a, ior=1.0, g=0.0, light=50.0, extra=()):
    """A homogeneous-medium cube [-1.5,1.5]^3 (object 0) holding an emissive quad at z = -1 that
    faces +z (object 1) and optional extra parts [(verts, faces, object)], all flat-shaded."""
    fog = _obj(ra.OBJ_VPM, has_vn=0, ior=ior)
    fog["sigma_s"] = sigma_s
    fog["sigma_a"] = sigma_a
    fog["g"] = g
    fog["volumetric"] = 1
    bv, bf = _box(1.5)
    lv, lf, _ = _quad((-0.3, -0.3, -1), (0.3, -0.3, -1), (0.3, 0.3, -1), (-0.3, 0.3, -1), (0, 0, 1))
    parts = [(bv, bf, fog), (lv, lf, _obj(ra.OBJ_EMISSIVE, (light, light, light), has_vn=0))] + list(extra)
    verts, tris, tri_obj, objects, off = [], [], [], [], 0
    for oi, (pv, pf, ob) in enumerate(parts):
        verts.append(pv)
        tris.append(pf + off)
        tri_obj.append(np.full(len(pf), oi, np.int32))
        objects.append(ob)
        off += len(pv)
    return ra.HostScene.from_arrays(np.concatenate(verts).astype(np.float32), np.concatenate(tris).astype(np.uint32),
                                    np.concatenate(tri_obj), np.array(objects, ra.OBJECT_DTYPE), None)


def random_rays(n, seed, n_objects, inside_obj=-1):
    """half camera-like rays from (0,0,2), half interior rays; plus packed ids for mode 2"""
    rng = np.random.default_rng(seed)
    o = rng.uniform(-0.9, 0.9, size=(3, n)).astype(np.float32)
    o[:, : n // 2] = np.array([[0.0], [0.0], [2.0]], np.float32)
    d = rng.normal(size=(3, n)).astype(np.float32)
    d[2, : n // 2] = -np.abs(d[2, : n // 2]) - 2.0
    d /= np.linalg.norm(d, axis=0, keepdims=True)
    contact = rng.integers(-1, n_objects, n)
    inside = np.where(rng.random(n) < 0.8, -1, inside_obj)
    return o, d.astype(np.float32), ra.pack_ids(contact, inside)


# ---- random scenes with media (scripts/fuzz_media.py, tests/test_gpu_fog.py) -------------------------------------
def random_medium(rng, ior=None):
    ob = _obj(ra.OBJ_VPM, has_vn=0, ior=float(ior if ior is not None else (1.0 if rng.random() < 0.5 else rng.uniform(1.1, 1.5))))
    ob["sigma_s"] = 10.0 ** rng.uniform(-1.5, 0.8, 3) if rng.random() < 0.8 else (0, 0, 0)
    ob["sigma_a"] = 10.0 ** rng.uniform(-2.5, 0.0, 3)
    ob["g"] = float(rng.choice([0.0, rng.uniform(-0.8, 0.9)]))
    ob["volumetric"] = 1
    return ob


def random_media_scene(rng, rich=0, extreme=False, scale=1.0, offset=(0.0, 0.0, 0.0)):
    """scripts/fuzz_media.py's scene: an emissive quad, optionally an outer fog cube or a floor, and 1 .. 5 closed objects
    (cubes, displaced icospheres) that are media, glass, mirrors, glossy / Phong / diffuse objects or emitters, at random
    places, overlapping or nested as they fall. Returns (HostScene, [object types])."""
    RICH = rich
    parts = []
    lz = float(rng.uniform(-1.2, -0.6))
    lv, lf, _ = _quad((-0.3, -0.3, lz), (0.3, -0.3, lz), (0.3, 0.3, lz), (-0.3, 0.3, lz), (0, 0, 1))
    parts.append((lv, lf, _obj(ra.OBJ_EMISSIVE, (40, 40, 40), has_vn=0)))
    outer = rng.random()
    if outer < 0.5:
        bv, bf = _box(1.5)
        parts.append((bv, bf, random_medium(rng)))
    elif outer < 0.8:
        fv, ff, _ = _quad((-1.4, -0.9, -1.4), (1.4, -0.9, -1.4), (1.4, -0.9, 1.4), (-1.4, -0.9, 1.4), (0, 1, 0))
        parts.append((fv, ff, _obj(ra.OBJ_DIFFUSE, (0.7, 0.6, 0.5), has_vn=0)))
    for k in range(int(rng.integers(1, 6))):
        c = rng.uniform(-0.8, 0.8, 3)
        smooth = 0
        if rng.random() < 0.5:
            v, f = _box(float(rng.uniform(0.15, 0.7)), tuple(c))
        else:
            v, f = displaced_sphere(int(rng.integers(1, 3 + RICH)), int(rng.integers(0, 1000)), float(rng.uniform(0.15, 0.6)), c,
                                            amp=float(rng.choice([0.0, 0.12])))
            smooth = int(rng.integers(0, 2)) if RICH else 0  # interpolated vertex normals (also on a medium's boundary)
        u = rng.random()
        if u < 0.5:
            ob = random_medium(rng)
        elif u < 0.7:
            ob = _obj(ra.OBJ_REFRACTIVE, (1, 1, 1), ior=float(rng.uniform(1.2, 1.7)) if not RICH or rng.random() < 0.7 else -1.0)
        elif u < 0.85:
            ob = _obj(ra.OBJ_GLOSSY if not RICH or rng.random() < 0.6 else ra.OBJ_PHONG, tuple(rng.uniform(0.5, 0.9, 3)),
                              coeff=float(rng.uniform(3, 60)), color2=tuple(rng.uniform(0.1, 0.5, 3)))
        elif RICH and u < 0.9:
            ob = _obj(ra.OBJ_EMISSIVE, tuple(rng.uniform(2, 10, 3)))
        else:
            ob = _obj(ra.OBJ_DIFFUSE, tuple(rng.uniform(0.3, 0.9, 3)))
        ob["has_vn"] = smooth
        if extreme and rng.random() < 0.7:  # parameters at and beyond the edges of what makes sense
            pick = lambda vals: float(vals[int(rng.integers(0, len(vals)))])
            which = int(rng.integers(0, 6))
            if which == 0: ob["ior"] = pick([0.0, 1e-3, 1.0, 50.0, 1e6] if int(ob["type"]) != ra.OBJ_REFRACTIVE else [0.0, 1e-3, 1.0, 50.0, 1e6, -1.0])
            elif which == 1: ob["g"] = pick([-1.0, 1.0, 0.999, -0.999, 1e-6, 2.0])
            elif which == 2: ob["sigma_s"] = (pick([0.0, 1e-8, 1e4]),) * 3; ob["sigma_a"] = (pick([0.0, 1e-8, 1e4]),) * 3
            elif which == 3: ob["coeff"] = pick([0.0, -1.0, -0.5, 1e4, 1e-3])
            elif which == 4: ob["color"] = (pick([0.0, 1.0, 2.0, -0.5, 1e-4]), pick([0.0, 1.0, 2.0]), pick([0.0, 1.0, 1e3]))
            else: ob["decay"] = (pick([0.0, 1.0, 2.0, 0.5]),) * 3
        parts.append((v, f, ob))
    verts, tris, tri_obj, objects, vns, off = [], [], [], [], [], 0
    for oi, (pv, pf, ob) in enumerate(parts):
        verts.append(pv); tris.append(pf + off); tri_obj.append(np.full(len(pf), oi, np.int32)); objects.append(ob); off += len(pv)
        vns.append(vertex_normals(np.asarray(pv, np.float64), np.asarray(pf, np.int64))[pf].reshape(len(pf), 9))
    V = np.concatenate(verts) * scale + np.asarray(offset)
    return ra.HostScene.from_arrays(V.astype(np.float32), np.concatenate(tris).astype(np.uint32),
                                    np.concatenate(tri_obj), np.array(objects, ra.OBJECT_DTYPE),
                                    np.concatenate(vns).astype(np.float32)), [int(o["type"]) for o in objects]
