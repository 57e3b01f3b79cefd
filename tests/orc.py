"""ctypes binding of oracle/liboracle.so — the CPU restatement used only as the checker."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "liboracle.so")


class OrcCamera(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("sight_dist", C.c_uint32),
                ("position", C.c_float * 3), ("focus", C.c_float * 3), ("up", C.c_float * 3)]


_lib = None


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_LIB):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
        L = C.CDLL(ORACLE_LIB)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int32, C.c_void_p]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_scene_rebuild.argtypes = [C.c_void_p, C.c_int32]
        L.orc_scene_rebuild.restype = None
        L.orc_scene_set_textures.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.orc_scene_set_textures.restype = None
        L.orc_texture_color.argtypes = [C.c_void_p] * 3
        L.orc_texture_color.restype = None
        L.orc_tex_coord.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.orc_tex_coord.restype = None
        L.orc_tree_num_nodes.argtypes = [C.c_void_p]
        L.orc_tree_depth.argtypes = [C.c_void_p]
        L.orc_tree_flatten.argtypes = [C.c_void_p] * 7
        L.orc_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_int]
        L.orc_intersect_ref_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                               C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_intersect_ordered.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                            C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_box_intersect.restype = C.c_float
        L.orc_box_intersect.argtypes = [C.c_void_p] * 4
        L.orc_tri_intersect.restype = C.c_float
        L.orc_tri_intersect.argtypes = [C.c_void_p] * 5
        L.orc_world_normal.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.orc_legal.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_int32]
        L.orc_rng_draw.restype = C.c_uint32
        L.orc_rng_draw.argtypes = [C.c_uint32] * 4
        L.orc_camera_ray.argtypes = [C.POINTER(OrcCamera), C.c_uint32, C.c_float, C.c_float, C.c_void_p]
        L.orc_hemi_cos.argtypes = [C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.orc_render_pt.argtypes = [C.c_void_p, C.POINTER(OrcCamera), C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
        L.orc_render_pttest.argtypes = [C.c_void_p, C.POINTER(OrcCamera), C.c_uint32, C.c_uint32, C.c_uint32,
                                        C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]
        L.orc_visibility.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_int, C.c_void_p, C.c_int]
        for f in ("orc_local_frame", "orc_to_world", "orc_face_normal", "orc_interp_normal", "orc_normalize"):
            getattr(L, f).restype = None
        L.orc_intersect_fast.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 8 + [C.c_size_t, C.c_int, C.c_void_p,
                                                                                       C.c_void_p, C.POINTER(C.c_uint64),
                                                                                       C.POINTER(C.c_uint64)]
        L.orc_trace_path.restype = C.c_int32
        L.orc_trace_path.argtypes = [C.c_void_p, C.POINTER(OrcCamera), C.c_uint32, C.c_uint32, C.c_uint32,
                                     C.c_uint32, C.c_void_p, C.c_int32, C.c_void_p]
        _lib = L
    return _lib


def camera_from(cam):
    """from a renderer_amd.Camera (same layout)."""
    oc = OrcCamera()
    C.memmove(C.byref(oc), C.byref(cam), C.sizeof(oc))
    return oc


class OracleScene:
    def __init__(self, raw):
        """raw: dict as renderer_amd.HostScene.raw() returns."""
        self.verts = np.ascontiguousarray(raw["verts"], np.float32)
        self.tris = np.ascontiguousarray(raw["tris"], np.uint32)
        self.tri_obj = np.ascontiguousarray(raw["tri_obj"], np.int32)
        self.tri_local = np.ascontiguousarray(raw["tri_local"], np.int32)
        self.tri_vn = None if raw.get("tri_vn") is None else np.ascontiguousarray(raw["tri_vn"], np.float32)
        self.objects = np.ascontiguousarray(raw["objects"])
        assert self.objects.dtype.itemsize == 92  # sizeof(OrcObject)
        self.h = lib().orc_scene_create(len(self.verts), _p(self.verts), len(self.tris), _p(self.tris),
                                        _p(self.tri_obj), _p(self.tri_local), _p(self.tri_vn),
                                        len(self.objects), _p(self.objects))
        self.n_nodes = lib().orc_tree_num_nodes(self.h)
        self.depth = lib().orc_tree_depth(self.h)
        # textures (renderer_amd.make_texture dicts; OrcTexture has MirtTexture's layout)
        if raw.get("textures"):
            import renderer_amd as ra
            self.tri_vt = None if raw.get("tri_vt") is None else np.ascontiguousarray(raw["tri_vt"], np.float32)
            self.textures = raw["textures"]
            self.object_tex = np.ascontiguousarray(raw["object_tex"], ra.OBJECT_TEX_DTYPE)
            arr = ra.texture_structs(self.textures)
            lib().orc_scene_set_textures(self.h, _p(self.tri_vt), len(self.textures), C.cast(arr, C.c_void_p),
                                         _p(self.object_tex))

    def close(self):
        if self.h:
            lib().orc_scene_destroy(self.h)
            self.h = None

    def rebuild(self, strategy):
        """KDTree::build(strategy): 0 = LOOP, 1 = BEST"""
        lib().orc_scene_rebuild(self.h, strategy)
        self.n_nodes = lib().orc_tree_num_nodes(self.h)
        self.depth = lib().orc_tree_depth(self.h)

    def flatten(self):
        N, T = self.n_nodes, len(self.tris)
        out = dict(node_min=np.empty((N, 3), np.float32), node_max=np.empty((N, 3), np.float32),
                   node_right=np.empty(N, np.int32), leaf_begin=np.empty(N, np.int32),
                   leaf_end=np.empty(N, np.int32), slot_tid=np.empty(T, np.int32))
        lib().orc_tree_flatten(self.h, _p(out["node_min"]), _p(out["node_max"]), _p(out["node_right"]),
                               _p(out["leaf_begin"]), _p(out["leaf_end"]), _p(out["slot_tid"]))
        return out

    def intersect(self, o, d, aux=None, mode=0, threads=8):
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        n = o.shape[1]
        a = None if aux is None else np.ascontiguousarray(aux, np.int32)
        t = np.empty(n, np.float32)
        tid = np.empty(n, np.int32)
        lib().orc_intersect(self.h, _p(o), _p(d), _p(a), n, mode, _p(t), _p(tid), threads)
        return t, tid

    def visibility(self, o, d, aux, max_dist, mode=2, threads=8):
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        a = None if aux is None else np.ascontiguousarray(aux, np.int32)
        m = np.ascontiguousarray(max_dist, np.float32)
        vis = np.empty(o.shape[1], np.uint8)
        lib().orc_visibility(self.h, _p(o), _p(d), _p(a), _p(m), o.shape[1], mode, _p(vis), threads)
        return vis

    def intersect_fast(self, tree, o, d, aux=None, mode=0):
        """restatement of the product's fast path over `tree` (renderer_amd.HostScene.fast_tree())"""
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        a = None if aux is None else np.ascontiguousarray(aux, np.int32)
        n = o.shape[1]
        t = np.empty(n, np.float32)
        tid = np.empty(n, np.int32)
        nb, nt = C.c_uint64(), C.c_uint64()
        tr = {k: np.ascontiguousarray(v) for k, v in tree.items()}
        lib().orc_intersect_fast(self.h, len(tr["node_right"]), _p(tr["node_min"]), _p(tr["node_max"]),
                                 _p(tr["node_right"]), _p(tr["leaf_begin"]), _p(tr["leaf_end"]), _p(o), _p(d), _p(a), n,
                                 mode, _p(t), _p(tid), C.byref(nb), C.byref(nt))
        return t, tid, nb.value, nt.value

    def ref_counts(self, o, d, aux=None, mode=0):
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        a = None if aux is None else np.ascontiguousarray(aux, np.int32)
        nb, nt = C.c_uint64(), C.c_uint64()
        lib().orc_intersect_ref_counts(self.h, _p(o), _p(d), _p(a), o.shape[1], mode, C.byref(nb), C.byref(nt))
        return nb.value, nt.value

    def intersect_ordered(self, o, d, aux=None, mode=0):
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        n = o.shape[1]
        a = None if aux is None else np.ascontiguousarray(aux, np.int32)
        t = np.empty(n, np.float32)
        tid = np.empty(n, np.int32)
        nb, nt = C.c_uint64(), C.c_uint64()
        lib().orc_intersect_ordered(self.h, _p(o), _p(d), _p(a), n, mode, _p(t), _p(tid), C.byref(nb), C.byref(nt))
        return t, tid, nb.value, nt.value

    def world_normal(self, tid, pos):
        pos = np.ascontiguousarray(pos, np.float32)
        out = np.empty(3, np.float32)
        lib().orc_world_normal(self.h, int(tid), _p(pos), _p(out))
        return out

    def legal(self, tid, o, d, dist, contact, inside):
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        return bool(lib().orc_legal(self.h, int(tid), _p(o), _p(d), float(dist), int(contact), int(inside)))

    def render(self, cam, spp_count, spp_begin=0, max_depth=20, seed=1, threads=8):
        oc = camera_from(cam)
        img = np.zeros((oc.height, oc.width, 3), np.float32)
        nr = C.c_uint64()
        lib().orc_render_pt(self.h, C.byref(oc), spp_begin, spp_count, max_depth, seed, _p(img), C.byref(nr),
                            threads)
        return img, nr.value

    def render_pttest(self, cam, spp_count, spp_begin=0, max_depth=20, seed=1, threads=8):
        """PathTracerTest: returns (sum image, closest-hit rays, shadow rays)"""
        oc = camera_from(cam)
        img = np.zeros((oc.height, oc.width, 3), np.float32)
        nr, ns = C.c_uint64(), C.c_uint64()
        rc = lib().orc_render_pttest(self.h, C.byref(oc), spp_begin, spp_count, max_depth, seed, _p(img), C.byref(nr),
                                     C.byref(ns), threads)
        if rc != 0:
            raise RuntimeError("orc_render_pttest failed (%d)" % rc)
        return img, nr.value, ns.value

    def trace_path(self, cam, pixel, sample, max_depth=20, seed=1, cap=64):
        oc = camera_from(cam)
        verts = np.zeros((cap, 10), np.float32)
        rgb = np.zeros(3, np.float32)
        n = lib().orc_trace_path(self.h, C.byref(oc), pixel, sample, max_depth, seed, _p(verts), cap, _p(rgb))
        return verts[:min(n, cap)], rgb, n


def _find_inside(self, o, d, current=-1):
    a = [np.ascontiguousarray(x, np.float32) for x in (o, d)]
    lib().orc_find_inside.restype = C.c_int32
    lib().orc_find_inside.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    return int(lib().orc_find_inside(self.h, _p(a[0]), _p(a[1]), current))


OracleScene.find_inside = _find_inside


def box_intersect(bmin, bmax, o, d):
    a = [np.ascontiguousarray(x, np.float32) for x in (bmin, bmax, o, d)]
    return float(lib().orc_box_intersect(*[_p(x) for x in a]))


def tri_intersect(v0, v1, v2, o, d):
    a = [np.ascontiguousarray(x, np.float32) for x in (v0, v1, v2, o, d)]
    return float(lib().orc_tri_intersect(*[_p(x) for x in a]))


def rng_draw(seed, pixel, sample, dim):
    return int(lib().orc_rng_draw(seed, pixel, sample, dim))


def camera_ray(cam, pixel, u, v):
    oc = camera_from(cam)
    out = np.empty(3, np.float32)
    lib().orc_camera_ray(C.byref(oc), pixel, u, v, _p(out))
    return out


def hemi_cos(exp_term, u1, u2):
    out = np.empty(3, np.float32)
    lib().orc_hemi_cos(exp_term, u1, u2, _p(out))
    return out
