"""renderer_amd — ctypes bindings of libmirt.so (include/mirt.h, include/mirt_host.h).

The package holds only what the hot path needs: the HIP kernels + C-ABI (csrc/), the C++ host
surface that mirrors the reference's Scene / ConfigManager / Renderer (host/), and these bindings,
which tests and bench.py use. There is no CPU implementation of the path here: without the built
library, or without a GPU, the compute calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIRT_LIB") or os.path.join(_HERE, "libmirt.so")  # MIRT_LIB: A/B builds
REPO_ROOT = os.path.dirname(_HERE)
DATA_DIR = os.path.join(REPO_ROOT, "Data")

OBJ_DIFFUSE, OBJ_GLOSSY, OBJ_REFRACTIVE, OBJ_EMISSIVE, OBJ_VPM, OBJ_PHONG = range(6)
MODE_UNCONDITIONAL, MODE_LAST_TID, MODE_NO_SELF = range(3)
NOALLOC_MAX_RAYS = 4 << 20  # MIRT_NOALLOC_MAX_RAYS

# MirtObject as a numpy structured dtype (same layout as the C struct; also the oracle's OrcObject)
OBJECT_DTYPE = np.dtype(
    [("type", "<i4"), ("has_vn", "<i4"), ("volumetric", "<i4"), ("tri_begin", "<i4"), ("tri_count", "<i4"),
     ("color", "<f4", 3), ("coeff", "<f4"), ("ior", "<f4"), ("decay", "<f4", 3), ("sigma_s", "<f4", 3),
     ("sigma_a", "<f4", 3), ("g", "<f4"), ("color2", "<f4", 3)], align=False)
assert OBJECT_DTYPE.itemsize == 92


class Camera(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("sight_dist", C.c_uint32),
                ("position", C.c_float * 3), ("focus", C.c_float * 3), ("up", C.c_float * 3)]


class RenderParams(C.Structure):
    _fields_ = [("camera", Camera), ("spp_begin", C.c_uint32), ("spp_count", C.c_uint32),
                ("max_depth", C.c_uint32), ("seed", C.c_uint32), ("spp_per_launch", C.c_uint32)]


class Tile(C.Structure):
    """MirtTile (include/mirt.h): a pixel window of the image, row 0 = top"""
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("width", C.c_int32), ("height", C.c_int32)]


SHARD_SAMPLES, SHARD_TILES = 0, 1  # MIRT_SHARD_*


class RenderStats(C.Structure):
    _fields_ = [("n_samples", C.c_uint64), ("n_rays", C.c_uint64), ("gpu_ms", C.c_double),
                ("intersect_ms", C.c_double), ("n_shadow_rays", C.c_uint64)]


class SceneDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_nodes", C.c_int32), ("node_min", C.c_void_p),
                ("node_max", C.c_void_p), ("node_right", C.c_void_p), ("node_leaf_begin", C.c_void_p),
                ("node_leaf_end", C.c_void_p), ("n_tris", C.c_int32), ("slot_tid", C.c_void_p),
                ("tri_pos", C.c_void_p), ("tri_obj", C.c_void_p), ("tri_local", C.c_void_p),
                ("tri_vn", C.c_void_p), ("n_objects", C.c_int32), ("objects", C.c_void_p),
                ("n_obj_trees", C.c_int32), ("obj_trees", C.c_void_p),
                ("tri_vt", C.c_void_p), ("n_textures", C.c_int32), ("textures", C.c_void_p),
                ("object_tex", C.c_void_p)]


class Texture(C.Structure):
    """MirtTexture (include/mirt.h)"""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("channels", C.c_int32), ("pad", C.c_int32),
                ("texels", C.c_void_p), ("coord_transform", C.c_float * 16), ("color_transform", C.c_float * 16)]


# MirtObjectTex: which texture (index, -1 none) each of an object's materials reads
OBJECT_TEX_DTYPE = np.dtype([("diffuse", "<i4"), ("glossy_color", "<i4"), ("glossy_coeff", "<i4"), ("bump", "<i4")])


def make_texture(texels, coord_transform=None, color_transform=None):
    """A texture as the Python side passes it around: texels (H, W, 3) rgb or (H, W) gray, row 0 first in
    memory (the file order of Texture::loadFile), 4x4 row-major transforms (identity by default)."""
    t = np.ascontiguousarray(texels, np.float32)
    assert t.ndim == 2 or (t.ndim == 3 and t.shape[2] == 3)
    eye = np.eye(4, dtype=np.float32)
    return dict(texels=t, coord=eye if coord_transform is None else np.ascontiguousarray(coord_transform, np.float32),
                color=eye if color_transform is None else np.ascontiguousarray(color_transform, np.float32))


def texture_structs(textures):
    """ctypes array of Texture for a list of make_texture() dicts (the dicts keep the texels alive)."""
    arr = (Texture * max(1, len(textures)))()
    for k, t in enumerate(textures):
        arr[k].height, arr[k].width = t["texels"].shape[:2]
        arr[k].channels = 3 if t["texels"].ndim == 3 else 1
        arr[k].texels = t["texels"].ctypes.data
        arr[k].coord_transform[:] = [float(x) for x in np.asarray(t["coord"], np.float32).reshape(16)]
        arr[k].color_transform[:] = [float(x) for x in np.asarray(t["color"], np.float32).reshape(16)]
    return arr


# every symbol include/mirt.h and include/mirt_host.h declare (checked by the CPU tests)
MIRT_SYMBOLS = ["mirt_create", "mirt_destroy", "mirt_last_error", "mirt_upload_scene", "mirt_scene_serial", "mirt_intersect",
                "mirt_intersect_dev", "mirt_intersect_noalloc_dev", "mirt_visibility", "mirt_visibility_dev", "mirt_fast_tree", "mirt_intersect_count_dev", "mirt_intersect_audit_dev", "mirt_generate_rays_dev", "mirt_render_pt",
                "mirt_render_pt_dev", "mirt_render_pttest", "mirt_render_pttest_dev", "mirt_render_pt_multi", "mirt_render_pttest_multi", "mirt_device_count",
                "mirt_render_pt_tile_dev", "mirt_render_pttest_tile_dev", "mirt_shard_tile", "mirt_render_pt_multi_shard",
                "mirt_render_pttest_multi_shard", "mirt_context_device",
                "mirt_capture_rays_dev", "mirt_device_info"]
MIRTH_SYMBOLS = ["mirth_last_error", "mirth_load_config", "mirth_scene_from_arrays", "mirth_rebuild_tree", "mirth_set_textures", "mirth_free",
                 "mirth_counts", "mirth_get_raw", "mirth_get_tree", "mirth_get_camera", "mirth_get_render_cfg",
                 "mirth_describe", "mirth_upload", "mirth_render", "mirth_render_progressive", "mirth_fill_intersect", "mirth_test_visibility", "mirth_test_path_list_visibility", "mirth_world_vertex", "mirth_normal_transform",
                 "mirth_shutdown", "mirth_is_resident", "mirth_set_devices",
                 "mirth_save_pfm", "mirth_load_pfm"]

_lib = None


def lib():
    """Loads libmirt.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("renderer_amd: %s is missing; build it with `make -C renderer_amd` "
                               "(or __graft_entry__.build()). There is no CPU fallback." % LIB_PATH)
        # torch ships its own HIP runtime (same SONAME as /opt/rocm's). Two HIP runtimes in one
        # process cannot both see the GPU, so let torch's load first and libmirt.so bind to it.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.mirt_last_error.restype = C.c_char_p
        L.mirt_last_error.argtypes = [C.c_void_p]
        L.mirth_last_error.restype = C.c_char_p
        for name in MIRT_SYMBOLS + MIRTH_SYMBOLS:
            fn = getattr(L, name)
            if name not in ("mirt_last_error", "mirth_last_error", "mirth_free", "mirth_shutdown", "mirt_scene_serial"):
                fn.restype = C.c_int
        L.mirth_free.restype = None
        L.mirth_free.argtypes = [C.c_void_p]
        L.mirth_shutdown.restype = None
        L.mirt_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.mirt_destroy.argtypes = [C.c_void_p]
        L.mirt_upload_scene.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
        L.mirt_scene_serial.argtypes = [C.c_void_p]
        L.mirt_scene_serial.restype = C.c_uint64
        L.mirt_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                     C.c_void_p, C.c_void_p]
        L.mirt_intersect_dev.argtypes = L.mirt_intersect.argtypes + [C.c_void_p]
        L.mirt_intersect_noalloc_dev.argtypes = L.mirt_intersect_dev.argtypes
        L.mirt_visibility.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                      C.c_int, C.c_void_p]
        L.mirt_visibility_dev.argtypes = L.mirt_visibility.argtypes + [C.c_void_p]
        L.mirt_fast_tree.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)] + [C.c_void_p] * 5
        L.mirt_intersect_count_dev.argtypes = L.mirt_intersect.argtypes + [C.POINTER(C.c_uint64),
                                                                          C.POINTER(C.c_uint64)]
        L.mirt_intersect_audit_dev.argtypes = L.mirt_intersect.argtypes + [C.POINTER(C.c_uint64 * 8)]
        L.mirt_generate_rays_dev.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32,
                                             C.c_void_p, C.c_void_p]
        L.mirt_render_pt.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.POINTER(RenderStats)]
        L.mirt_render_pt_dev.argtypes = L.mirt_render_pt.argtypes
        L.mirt_render_pttest.argtypes = L.mirt_render_pt.argtypes
        L.mirt_render_pttest_dev.argtypes = L.mirt_render_pt.argtypes
        L.mirt_render_pt_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(RenderParams), C.c_void_p,
                                           C.POINTER(RenderStats)]
        L.mirt_render_pttest_multi.argtypes = L.mirt_render_pt_multi.argtypes
        L.mirt_render_pt_tile_dev.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.POINTER(Tile), C.c_void_p,
                                              C.POINTER(RenderStats)]
        L.mirt_render_pttest_tile_dev.argtypes = L.mirt_render_pt_tile_dev.argtypes
        L.mirt_shard_tile.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Tile)]
        L.mirt_render_pt_multi_shard.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(RenderParams), C.c_int, C.c_void_p,
                                                 C.POINTER(RenderStats)]
        L.mirt_render_pttest_multi_shard.argtypes = L.mirt_render_pt_multi_shard.argtypes
        L.mirt_context_device.argtypes = [C.c_void_p]
        L.mirt_device_count.argtypes = []
        L.mirth_set_devices.argtypes = [C.c_void_p, C.c_int32]
        L.mirt_capture_rays_dev.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_uint32, C.c_uint32,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                            C.POINTER(C.c_size_t)]
        L.mirt_device_info.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.mirth_load_config.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.mirth_scene_from_arrays.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]
        L.mirth_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int32 * 5)]
        L.mirth_set_textures.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.mirth_rebuild_tree.argtypes = [C.c_void_p, C.c_int32]
        L.mirth_get_raw.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.mirth_get_tree.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.mirth_get_camera.argtypes = [C.c_void_p, C.POINTER(Camera)]
        L.mirth_get_render_cfg.argtypes = [C.c_void_p, C.POINTER(C.c_uint32 * 3)]
        L.mirth_describe.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
        L.mirth_upload.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.mirth_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(RenderStats)]
        L.mirth_fill_intersect.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 8
        L.mirth_render_progressive.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.c_char_p, C.c_char_p,
                                               C.c_void_p, C.POINTER(RenderStats)]
        L.mirth_test_visibility.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 6
        L.mirth_test_path_list_visibility.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 10 + [C.c_size_t]
        L.mirth_world_vertex.argtypes = [C.c_void_p] * 3
        L.mirth_world_vertex.restype = None
        L.mirth_normal_transform.argtypes = [C.c_void_p] * 3
        L.mirth_normal_transform.restype = None
        L.mirth_is_resident.argtypes = [C.c_void_p]
        L.mirth_save_pfm.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32]
        L.mirth_load_pfm.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32 * 2)]
        _lib = L
    return _lib


class MirtError(RuntimeError):
    pass


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class HostScene:
    """A scene on the host: loaded from a config (ConfigManager::load) or made from arrays."""

    def __init__(self, handle):
        self.h = handle
        self._refresh_counts()

    def _refresh_counts(self):
        cnt = (C.c_int32 * 5)()
        lib().mirth_counts(self.h, C.byref(cnt))
        self.n_verts, self.n_tris, self.n_objects, self.n_nodes, self.depth = list(cnt)

    @staticmethod
    def load_config(path):
        h = C.c_void_p()
        if lib().mirth_load_config(os.fsencode(path), C.byref(h)) != 0:
            raise MirtError(lib().mirth_last_error().decode())
        return HostScene(h)

    @staticmethod
    def from_arrays(verts, tris, tri_obj, objects, tri_vn=None):
        verts = np.ascontiguousarray(verts, np.float32)
        tris = np.ascontiguousarray(tris, np.uint32)
        tri_obj = np.ascontiguousarray(tri_obj, np.int32)
        objects = np.ascontiguousarray(objects, OBJECT_DTYPE)
        if tri_vn is not None:
            tri_vn = np.ascontiguousarray(tri_vn, np.float32)
        h = C.c_void_p()
        rc = lib().mirth_scene_from_arrays(len(verts), _ptr(verts), len(tris), _ptr(tris), _ptr(tri_obj),
                                           _ptr(tri_vn), len(objects), _ptr(objects), C.byref(h))
        if rc != 0:
            raise MirtError(lib().mirth_last_error().decode())
        return HostScene(h)

    LOOP, BEST = 0, 1  # KDTree::Strategy (KDTree.h:24)

    def rebuild_tree(self, strategy):
        """KDTree::build(strategy): LOOP cycles the split axis with the depth, BEST (the loader's) picks it per node."""
        if lib().mirth_rebuild_tree(self.h, strategy) != 0:
            raise MirtError(lib().mirth_last_error().decode())
        self._refresh_counts()

    def set_textures(self, tri_vt, textures, object_tex):
        """tri_vt: (T, 9) corner texture coordinates or None; textures: list of make_texture() dicts;
        object_tex: (n_objects,) OBJECT_TEX_DTYPE."""
        vt = None if tri_vt is None else np.ascontiguousarray(tri_vt, np.float32).reshape(self.n_tris, 9)
        ot = np.ascontiguousarray(object_tex, OBJECT_TEX_DTYPE)
        assert len(ot) == self.n_objects
        arr = texture_structs(textures)
        if lib().mirth_set_textures(self.h, _ptr(vt), len(textures), C.cast(arr, C.c_void_p), _ptr(ot)) != 0:
            raise MirtError(lib().mirth_last_error().decode())

    def close(self):
        if self.h:
            lib().mirth_free(self.h)
            self.h = None

    def raw(self):
        """dict of the pre-tree scene arrays (world-space)."""
        V, T, O = self.n_verts, self.n_tris, self.n_objects
        out = dict(verts=np.empty((V, 3), np.float32), tris=np.empty((T, 3), np.uint32),
                   tri_obj=np.empty(T, np.int32), tri_local=np.empty(T, np.int32),
                   tri_vn=np.empty((T, 9), np.float32), objects=np.empty(O, OBJECT_DTYPE))
        lib().mirth_get_raw(self.h, _ptr(out["verts"]), _ptr(out["tris"]), _ptr(out["tri_obj"]),
                            _ptr(out["tri_local"]), _ptr(out["tri_vn"]), _ptr(out["objects"]))
        # textures, read through the descriptor (copies)
        d = self.describe()
        out["tri_vt"] = None
        if d.tri_vt:
            out["tri_vt"] = np.ctypeslib.as_array(C.cast(d.tri_vt, C.POINTER(C.c_float)), (T, 9)).copy()
        out["textures"], out["object_tex"] = [], None
        if d.n_textures > 0:
            arr = C.cast(d.textures, C.POINTER(Texture))
            for k in range(d.n_textures):
                X = arr[k]
                shape = (X.height, X.width, 3) if X.channels == 3 else (X.height, X.width)
                tex = np.ctypeslib.as_array(C.cast(X.texels, C.POINTER(C.c_float)), shape).copy()
                out["textures"].append(make_texture(tex, np.array(X.coord_transform[:], np.float32).reshape(4, 4),
                                                    np.array(X.color_transform[:], np.float32).reshape(4, 4)))
            if d.object_tex:
                n = O * OBJECT_TEX_DTYPE.itemsize
                buf = (C.c_char * n).from_address(d.object_tex)
                out["object_tex"] = np.frombuffer(buf, OBJECT_TEX_DTYPE, O).copy()
        return out

    def fast_tree(self):
        """mirt_fast_tree: the surface-area hierarchy over this scene's triangles (host only)."""
        N = max(1, 2 * self.n_tris)
        out = dict(node_min=np.empty((N, 3), np.float32), node_max=np.empty((N, 3), np.float32),
                   node_right=np.empty(N, np.int32), leaf_begin=np.empty(N, np.int32), leaf_end=np.empty(N, np.int32))
        n = C.c_int32()
        d = self.describe()
        rc = lib().mirt_fast_tree(C.byref(d), N, C.byref(n), _ptr(out["node_min"]), _ptr(out["node_max"]),
                                  _ptr(out["node_right"]), _ptr(out["leaf_begin"]), _ptr(out["leaf_end"]))
        if rc != 0:
            raise MirtError("mirt_fast_tree failed (%d): %s" % (rc, lib().mirt_last_error(None).decode()))
        return {k: v[:n.value] for k, v in out.items()}

    def tree(self):
        N, T = self.n_nodes, self.n_tris
        out = dict(node_min=np.empty((N, 3), np.float32), node_max=np.empty((N, 3), np.float32),
                   node_right=np.empty(N, np.int32), leaf_begin=np.empty(N, np.int32),
                   leaf_end=np.empty(N, np.int32), slot_tid=np.empty(T, np.int32))
        lib().mirth_get_tree(self.h, _ptr(out["node_min"]), _ptr(out["node_max"]), _ptr(out["node_right"]),
                             _ptr(out["leaf_begin"]), _ptr(out["leaf_end"]), _ptr(out["slot_tid"]))
        return out

    def camera(self):
        cam = Camera()
        lib().mirth_get_camera(self.h, C.byref(cam))
        return cam

    def render_cfg(self):
        out = (C.c_uint32 * 3)()
        lib().mirth_get_render_cfg(self.h, C.byref(out))
        return dict(max_depth=out[0], spp=out[1], seed=out[2])

    def describe(self):
        d = SceneDesc()
        lib().mirth_describe(self.h, C.byref(d))
        return d

    def upload(self):
        """Scene::buildKDTree's device step; returns a Device bound to the service's ONE shared context (not owned):
        a later upload of another scene re-points it. Queries made through a HostScene (fill_intersect,
        test_visibility, render) notice that and upload their own scene again first."""
        ctx = C.c_void_p()
        if lib().mirth_upload(self.h, C.byref(ctx)) != 0:
            raise MirtError(lib().mirth_last_error().decode())
        return Device(ctx, owned=False)

    def render(self, spp=0, seed=0):
        cam = self.camera()
        img = np.zeros((cam.height, cam.width, 3), np.float32)
        st = RenderStats()
        if lib().mirth_render(self.h, spp, seed, _ptr(img), C.byref(st)) != 0:
            raise MirtError(lib().mirth_last_error().decode())
        return img, st

    def render_progressive(self, spp, seed=0, output_iter=0, resume_from=None, save_state=None):
        """Renderer::render with periodic PFM dumps / resumable accumulators; spp = total passes wanted."""
        cam = self.camera()
        img = np.zeros((cam.height, cam.width, 3), np.float32)
        st = RenderStats()
        enc = lambda p: None if p is None else os.fsencode(str(p))
        if lib().mirth_render_progressive(self.h, spp, seed, output_iter, enc(resume_from), enc(save_state), _ptr(img),
                                          C.byref(st)) != 0:
            raise MirtError(lib().mirth_last_error().decode())
        return img, st

    def fill_intersect(self, o, d, contact, inside):
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        n = len(o)
        contact = np.ascontiguousarray(contact, np.int32)
        inside = np.ascontiguousarray(inside, np.int32)
        dist = np.empty(n, np.float32)
        tid, obj, local = (np.empty(n, np.int32) for _ in range(3))
        if lib().mirth_fill_intersect(self.h, n, _ptr(o), _ptr(d), _ptr(contact), _ptr(inside), _ptr(dist),
                                      _ptr(tid), _ptr(obj), _ptr(local)) != 0:
            raise MirtError(lib().mirth_last_error().decode())
        return dist, tid, obj, local

    def test_visibility(self, o, d, contact, inside, dist):
        """Scene::testVisibility on AoS rays [n,3]; dist[n] = distance to the connected vertex."""
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        n = len(o)
        contact = np.ascontiguousarray(contact, np.int32)
        inside = np.ascontiguousarray(inside, np.int32)
        dist = np.ascontiguousarray(dist, np.float32)
        vis = np.empty(n, np.uint8)
        if lib().mirth_test_visibility(self.h, n, _ptr(o), _ptr(d), _ptr(contact), _ptr(inside), _ptr(dist),
                                       _ptr(vis)) != 0:
            raise MirtError(lib().mirth_last_error().decode())
        return vis


def _path_list_visibility(self, start_paths, end_paths):
    """MCRenderer::testPathListVisibility. start_paths[p]: dict(o [n,3], contact [n], inside [n], random [n]);
    end_paths[p]: dict(o [m,3], random [m]). Returns a list of bool arrays [n, m]."""
    P = len(start_paths)
    assert P == len(end_paths)
    def cat(L, k, dt):
        w = 3 if k == "o" else 1
        parts = [np.asarray(x[k], dt).reshape(-1, w) for x in L]
        return np.ascontiguousarray(np.concatenate(parts) if parts else np.zeros((0, w), dt))
    so, eo = cat(start_paths, "o", np.float32), cat(end_paths, "o", np.float32)
    sc, si = cat(start_paths, "contact", np.int32), cat(start_paths, "inside", np.int32)
    sr, er = cat(start_paths, "random", np.uint8), cat(end_paths, "random", np.uint8)
    soff = np.concatenate([[0], np.cumsum([len(x["o"]) for x in start_paths])]).astype(np.int32)
    eoff = np.concatenate([[0], np.cumsum([len(x["o"]) for x in end_paths])]).astype(np.int32)
    words = int(sum((len(a["o"]) * len(b["o"]) + 31) // 32 for a, b in zip(start_paths, end_paths)))
    bits = np.zeros(max(words, 1), np.uint32)
    boff = np.zeros(P + 1, np.int32)
    if lib().mirth_test_path_list_visibility(self.h, P, _ptr(soff), _ptr(so), _ptr(sc), _ptr(si), _ptr(sr), _ptr(eoff),
                                             _ptr(eo), _ptr(er), _ptr(boff), _ptr(bits), len(bits)) != 0:
        raise MirtError(lib().mirth_last_error().decode())
    out = []
    for p, (a, b) in enumerate(zip(start_paths, end_paths)):
        n, m = len(a["o"]), len(b["o"])
        w = bits[boff[p]:boff[p + 1]]
        flat = ((w[:, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(-1)[:n * m].astype(bool)
        out.append(flat.reshape(n, m))
    return out


HostScene.test_path_list_visibility = _path_list_visibility


def save_pfm(path, img):
    """MCRenderer::saveImagePFM (MCRenderer.cpp:500-518) for an [H, W, 3] float32 image"""
    img = np.ascontiguousarray(img, np.float32)
    assert img.ndim == 3 and img.shape[2] == 3
    if lib().mirth_save_pfm(os.fsencode(str(path)), _ptr(img), img.shape[1], img.shape[0]) != 0:
        raise MirtError(lib().mirth_last_error().decode())


def pack_ids(contact, inside):
    contact = np.asarray(contact, np.int64)
    inside = np.asarray(inside, np.int64)
    return (((contact + 1) & 0xFFFF) | (((inside + 1) & 0xFFFF) << 16)).astype(np.uint32).view(np.int32)


class Device:
    """One MirtContext (one GPU)."""

    def __init__(self, ctx=None, device=0, owned=True):
        self.owned = owned
        if ctx is None:
            ctx = C.c_void_p()
            rc = lib().mirt_create(device, C.byref(ctx))
            if rc != 0:
                raise MirtError("mirt_create failed (%d): %s" % (rc, lib().mirt_last_error(None).decode()))
        self.ctx = ctx

    def close(self):
        if self.owned and self.ctx:
            lib().mirt_destroy(self.ctx)
        self.ctx = None

    def _check(self, rc):
        if rc != 0:
            raise MirtError("mirt error %d: %s" % (rc, lib().mirt_last_error(self.ctx).decode()))

    def info(self):
        buf = C.create_string_buffer(512)
        self._check(lib().mirt_device_info(self.ctx, buf, 512))
        return buf.value.decode()

    def upload(self, desc):
        self._check(lib().mirt_upload_scene(self.ctx, C.byref(desc)))

    def intersect(self, origins, directions, aux=None, mode=MODE_UNCONDITIONAL):
        """origins, directions: float32 [3, n] plane-major (host). Returns (t[n], tid[n])."""
        o = np.ascontiguousarray(origins, np.float32)
        d = np.ascontiguousarray(directions, np.float32)
        assert o.ndim == 2 and o.shape[0] == 3 and o.shape == d.shape
        n = o.shape[1]
        a = None if aux is None else np.ascontiguousarray(aux, np.int32)
        t = np.empty(n, np.float32)
        tid = np.empty(n, np.int32)
        self._check(lib().mirt_intersect(self.ctx, _ptr(o), _ptr(d), _ptr(a), n, mode, _ptr(t), _ptr(tid)))
        return t, tid

    def visibility(self, origins, directions, aux, max_dist, mode=MODE_NO_SELF):
        """MCRenderer::testVisibility for a batch (host arrays): uint8[n], 1 = nothing nearer than max_dist."""
        o = np.ascontiguousarray(origins, np.float32)
        d = np.ascontiguousarray(directions, np.float32)
        assert o.ndim == 2 and o.shape[0] == 3 and o.shape == d.shape
        n = o.shape[1]
        a = None if aux is None else np.ascontiguousarray(aux, np.int32)
        m = np.ascontiguousarray(max_dist, np.float32)
        assert m.shape == (n,)
        vis = np.empty(n, np.uint8)
        self._check(lib().mirt_visibility(self.ctx, _ptr(o), _ptr(d), _ptr(a), _ptr(m), n, mode, _ptr(vis)))
        return vis

    def visibility_dev(self, o_ptr, d_ptr, aux_ptr, max_dist_ptr, n, mode, vis_ptr, stream=None):
        self._check(lib().mirt_visibility_dev(self.ctx, o_ptr, d_ptr, aux_ptr, max_dist_ptr, n, mode, vis_ptr, stream))

    def intersect_dev(self, o_ptr, d_ptr, aux_ptr, n, mode, t_ptr, tid_ptr, stream=None):
        self._check(lib().mirt_intersect_dev(self.ctx, o_ptr, d_ptr, aux_ptr, n, mode, t_ptr, tid_ptr, stream))

    def intersect_noalloc_dev(self, o_ptr, d_ptr, aux_ptr, n, mode, t_ptr, tid_ptr, stream=None):
        """mirt_intersect_noalloc_dev: same answers, scratch from the static device array, n <= NOALLOC_MAX_RAYS"""
        self._check(lib().mirt_intersect_noalloc_dev(self.ctx, o_ptr, d_ptr, aux_ptr, n, mode, t_ptr, tid_ptr, stream))

    def intersect_count_dev(self, o_ptr, d_ptr, aux_ptr, n, mode, t_ptr, tid_ptr):
        nb, nt = C.c_uint64(), C.c_uint64()
        self._check(lib().mirt_intersect_count_dev(self.ctx, o_ptr, d_ptr, aux_ptr, n, mode, t_ptr, tid_ptr,
                                                   C.byref(nb), C.byref(nt)))
        return nb.value, nt.value

    def intersect_audit_dev(self, o_ptr, d_ptr, aux_ptr, n, mode, t_ptr, tid_ptr):
        """dict(box, tri, ambiguous, mismatch, walks) of the instrumented kernel"""
        a = (C.c_uint64 * 8)()
        self._check(lib().mirt_intersect_audit_dev(self.ctx, o_ptr, d_ptr, aux_ptr, n, mode, t_ptr, tid_ptr,
                                                   C.byref(a)))
        return dict(box=a[0], tri=a[1], ambiguous=a[2], mismatch=a[3], walks=a[4], wave_inner=a[5],
                    wave_leaf=a[6], wave_outer=a[7])

    def generate_rays_dev(self, cam, spp_begin, spp_count, seed, o_ptr, d_ptr):
        self._check(lib().mirt_generate_rays_dev(self.ctx, C.byref(cam), spp_begin, spp_count, seed, o_ptr, d_ptr))

    def render_pt(self, cam, spp_count, spp_begin=0, max_depth=20, seed=1, spp_per_launch=0):
        """Returns (rgb_sum [H, W, 3] float32 on the host, RenderStats)."""
        p = RenderParams(cam, spp_begin, spp_count, max_depth, seed, spp_per_launch)
        img = np.zeros((cam.height, cam.width, 3), np.float32)
        st = RenderStats()
        self._check(lib().mirt_render_pt(self.ctx, C.byref(p), _ptr(img), C.byref(st)))
        return img, st

    def render_pttest(self, cam, spp_count, spp_begin=0, max_depth=20, seed=1, spp_per_launch=0):
        """PathTracerTest (light connections + MIS): (rgb_sum [H, W, 3] float32 on the host, RenderStats)."""
        p = RenderParams(cam, spp_begin, spp_count, max_depth, seed, spp_per_launch)
        img = np.zeros((cam.height, cam.width, 3), np.float32)
        st = RenderStats()
        self._check(lib().mirt_render_pttest(self.ctx, C.byref(p), _ptr(img), C.byref(st)))
        return img, st

    def render_pttest_dev(self, cam, rgb_sum_ptr, spp_count, spp_begin=0, max_depth=20, seed=1, spp_per_launch=0):
        p = RenderParams(cam, spp_begin, spp_count, max_depth, seed, spp_per_launch)
        st = RenderStats()
        self._check(lib().mirt_render_pttest_dev(self.ctx, C.byref(p), rgb_sum_ptr, C.byref(st)))
        return st

    def render_pt_dev(self, cam, rgb_sum_ptr, spp_count, spp_begin=0, max_depth=20, seed=1, spp_per_launch=0):
        p = RenderParams(cam, spp_begin, spp_count, max_depth, seed, spp_per_launch)
        st = RenderStats()
        self._check(lib().mirt_render_pt_dev(self.ctx, C.byref(p), rgb_sum_ptr, C.byref(st)))
        return st

    def render_tile_dev(self, cam, tile, rgb_sum_ptr, spp_count, spp_begin=0, max_depth=20, seed=1, spp_per_launch=0,
                        pttest=False):
        """mirt_render_pt_tile_dev / mirt_render_pttest_tile_dev: ADDS the window's samples into the DEVICE buffer
        float[tile.height * tile.width * 3]; tile None = the whole image."""
        p = RenderParams(cam, spp_begin, spp_count, max_depth, seed, spp_per_launch)
        st = RenderStats()
        fn = lib().mirt_render_pttest_tile_dev if pttest else lib().mirt_render_pt_tile_dev
        self._check(fn(self.ctx, C.byref(p), None if tile is None else C.byref(tile), rgb_sum_ptr, C.byref(st)))
        return st

    def device_index(self):
        return lib().mirt_context_device(self.ctx)

    def capture_rays_dev(self, cam, spp_count, depth_lo, depth_hi, o_ptr, d_ptr, aux_ptr, capacity,
                         spp_begin=0, max_depth=20, seed=1):
        p = RenderParams(cam, spp_begin, spp_count, max_depth, seed, 0)
        n = C.c_size_t()
        self._check(lib().mirt_capture_rays_dev(self.ctx, C.byref(p), depth_lo, depth_hi, o_ptr, d_ptr, aux_ptr,
                                                capacity, C.byref(n)))
        return n.value


def render_pt_multi(devices, cam, spp_count, spp_begin=0, max_depth=20, seed=1, spp_per_launch=0, pttest=False,
                    shard=None):
    """mirt_render_pt_multi over a list of Device objects (each holding the scene): contiguous sample shards, one
    reduce on devices[0]'s GPU; shard=SHARD_TILES (mirt_render_pt_multi_shard): one image tile per device, all
    samples, gathered. Returns (rgb_sum [H, W, 3] float32 on the host, RenderStats)."""
    p = RenderParams(cam, spp_begin, spp_count, max_depth, seed, spp_per_launch)
    img = np.zeros((cam.height, cam.width, 3), np.float32)
    st = RenderStats()
    arr = (C.c_void_p * len(devices))(*[d.ctx for d in devices])
    if shard is None:
        fn = lib().mirt_render_pttest_multi if pttest else lib().mirt_render_pt_multi
        rc = fn(arr, len(devices), C.byref(p), _ptr(img), C.byref(st))
    else:
        fn = lib().mirt_render_pttest_multi_shard if pttest else lib().mirt_render_pt_multi_shard
        rc = fn(arr, len(devices), C.byref(p), int(shard), _ptr(img), C.byref(st))
    if rc != 0:
        raise MirtError("mirt error %d: %s" % (rc, lib().mirt_last_error(devices[0].ctx).decode()))
    return img, st


def set_devices(devices):
    """IntersectionHIP::setDevices: the host surface's renders (HostScene.render ...) use these GPUs from now on."""
    arr = np.ascontiguousarray(devices, np.int32)
    if lib().mirth_set_devices(_ptr(arr), len(arr)) != 0:
        raise MirtError(lib().mirth_last_error().decode())


def make_camera(width, height, sight_dist, position=(0, 0, 2), focus=(0, 0, 0), up=(0, 1, 0)):
    c = Camera()
    c.width, c.height, c.sight_dist = width, height, sight_dist
    c.position[:] = position
    c.focus[:] = focus
    c.up[:] = up
    return c


def shard_tile(width, height, world_size, rank):
    """mirt_shard_tile: the image tile of rank `rank` when an image is sharded by tiles (host arithmetic, no GPU)."""
    t = Tile()
    if lib().mirt_shard_tile(width, height, world_size, rank, C.byref(t)) != 0:
        raise MirtError("mirt_shard_tile: bad arguments")
    return t


def shard_samples(spp_total, world_size, rank):
    """Contiguous sample-index shard of rank `rank`: [begin, begin+count)."""
    base, rem = divmod(spp_total, world_size)
    begin = rank * base + min(rank, rem)
    return begin, base + (1 if rank < rem else 0)
