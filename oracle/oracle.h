/* oracle.h — C ABI of liboracle.so: a CPU restatement of the reference algorithm for the hot path
 * (closest-hit over the "KDTree" + the PathTracer sample loop).
 *
 * TEST INFRASTRUCTURE ONLY. Nothing in the product (renderer_amd/, libmirt.so) links, loads or
 * calls this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, as the
 * checker. It is deliberately written in the reference's own shape (pointer tree, recursive
 * both-children traversal, recursive samplePath with a stored vertex list) so that it can be read
 * side by side with the reference files each function cites.
 *
 * PARITY UNPINNED for the traversal, the predicate, the materials and the sample loop: the
 * reference ships no tests, golden vectors or images for this path, and its .cpp
 * files cannot be compiled in this image without writing stand-in headers (every one includes the
 * MSVC precompiled header "StdAfx.h" -> <tchar.h>, <SDKDDKVer.h>; the integrator base needs OpenCV
 * and <io.h>). Those parts are checked only against hand-derived known answers and invariants.
 * PINNED: the math layer under them. The reference's header-only code (nvVector.h, nvMatrix.h,
 * LocalFrame.h, HGPhaseFunc.h) does build here unmodified; oracle/ref_math.cpp drives it in place
 * (oracle/_ref/libref_math.so) and tests/golden/ref_math.npz records its outputs, which the
 * orc_* math entry points below and the product's host math reproduce bit for bit
 * (tests/test_ref_math.py).
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_DIFFUSE 0
#define ORC_GLOSSY 1
#define ORC_REFRACTIVE 2
#define ORC_EMISSIVE 3
#define ORC_VPM 4
#define ORC_PHONG 5

typedef struct OrcObject {
    int32_t type, has_vn, volumetric, tri_begin, tri_count;
    float color[3];
    float coeff;
    float ior;
    float decay[3];
    float sigma_s[3];
    float sigma_a[3];
    float g;
    float color2[3]; /* ORC_PHONG: glossy colour (color = diffuse) */
} OrcObject;

/* Texture.h:14-42 as data: same layout as MirtTexture (include/mirt.h) */
typedef struct OrcTexture {
    int32_t width, height, channels, pad; /* channels 3 (rgb) or 1 (gray) */
    const float* texels;
    float coord_transform[16]; /* row-major */
    float color_transform[16];
} OrcTexture;
/* which textures an object's materials read (-1: the constant of OrcObject); same layout as MirtObjectTex */
typedef struct OrcObjectTex {
    int32_t diffuse, glossy_color, glossy_coeff, bump; /* bump: SceneObject::bumpTex, gray, objects with texture coordinates only */
} OrcObjectTex;

typedef struct OrcCamera {
    int32_t width, height;
    uint32_t sight_dist;
    float position[3], focus[3], up[3];
} OrcCamera;

typedef struct OrcScene OrcScene;

/* Raw (pre-tree) scene: world-space vertices, global-index triangles (Scene.cpp:248-273). */
OrcScene* orc_scene_create(int32_t n_verts, const float* verts, int32_t n_tris,
                           const uint32_t* tris, const int32_t* tri_obj, const int32_t* tri_local,
                           const float* tri_vn /* [T*9] or NULL */, int32_t n_objects,
                           const OrcObject* objects);
void orc_scene_destroy(OrcScene* s);
/* Attaches texture coordinates (tri_vt: [n_tris*9], 3 corners x (x y z), by global tid; may be NULL) and
 * textures (texels are copied) to a scene; object_tex: [n_objects].                                 */
void orc_scene_set_textures(OrcScene* s, const float* tri_vt, int32_t n_textures, const OrcTexture* textures,
                            const OrcObjectTex* object_tex);
/* Texture::getColor (Texture.cpp:43-94) and SimpleShape::getTexCoord (SimpleShape.cpp:274-298) */
void orc_texture_color(const OrcTexture* tex, const float coord[3], float out[3]);
void orc_tex_coord(const OrcScene* s, int32_t tid, const float pos[3], float out[3]);
/* matrix4 inverse / matrix4 * vec4(v, w) as the texture path evaluates them (nvMatrix.h:233-369, 415-424) */
void orc_mat_inverse(const float m[16], float out[16]);
void orc_mat_apply(const float m[16], const float v[3], float w, float out[3]);

/* tree facts + pre-order flatten (KDTree.cpp:346-383) */
int32_t orc_tree_num_nodes(const OrcScene* s);
int32_t orc_tree_depth(const OrcScene* s);
void orc_tree_flatten(const OrcScene* s, float* node_min, float* node_max, int32_t* node_right,
                      int32_t* leaf_begin, int32_t* leaf_end, int32_t* slot_tid);

/* reference closest hit (recursive, both children, KDTree.cpp:295-339); rays plane-major
 * float[3][n]; mode 0 unconditional, 1 skip aux==tid (Intersection.frag:63-64), 2 NoSelfIntersection
 * with aux = (contact+1)|((inside+1)<<16). threads: OpenMP threads over rays (1 = serial).       */
void orc_intersect(const OrcScene* s, const float* origins, const float* directions,
                   const int32_t* aux, size_t n, int mode, float* t_out, int32_t* tid_out,
                   int threads);
/* MCRenderer::testVisibility (MCRenderer.cpp:106-118) per ray: 1 when the reference closest hit
 * under `mode` is a miss or dist - max_dist[i] >= -EPSILON (macros.h:5), else 0.                  */
void orc_visibility(const OrcScene* s, const float* origins, const float* directions,
                    const int32_t* aux, const float* max_dist, size_t n, int mode,
                    uint8_t* visible, int threads);
/* per-ray visit counters of the REFERENCE traversal order (both children, no pruning) */
void orc_intersect_ref_counts(const OrcScene* s, const float* origins, const float* directions,
                              const int32_t* aux, size_t n, int mode, uint64_t* n_box,
                              uint64_t* n_tri);
/* restatement of the PRODUCT's traversal order (near-first, distance-pruned) used only to count
 * node-box and triangle tests for the algorithmic-bytes figure; also returns hits.             */
void orc_intersect_ordered(const OrcScene* s, const float* origins, const float* directions,
                           const int32_t* aux, size_t n, int mode, float* t_out, int32_t* tid_out,
                           uint64_t* n_box, uint64_t* n_tri);

/* restatement of the product's FAST path over a tree GIVEN by the caller (pre-order, left child = i+1,
 * leaves carrying slot ranges of this scene's reference tree): conservative near-first traversal, then the
 * winner's chain of reference boxes by the reference test, exact re-run when that fails. Hits + counts. */
void orc_intersect_fast(const OrcScene* s, int32_t n_nodes, const float* node_min, const float* node_max,
                        const int32_t* node_right, const int32_t* leaf_begin, const int32_t* leaf_end,
                        const float* origins, const float* directions, const int32_t* aux, size_t n, int mode,
                        float* t_out, int32_t* tid_out, uint64_t* n_box, uint64_t* n_tri);

/* known-answer entry points for single functions */
float orc_box_intersect(const float bmin[3], const float bmax[3], const float o[3],
                        const float d[3]); /* KDTree.cpp:34-52 */
float orc_tri_intersect(const float v0[3], const float v1[3], const float v2[3], const float o[3],
                        const float d[3]); /* KDTree.cpp:267-293, no condition */
void orc_world_normal(const OrcScene* s, int32_t tid, const float pos[3],
                      float n_out[3]); /* SimpleShape.cpp:300-339 via SceneObject.cpp:250-254 */
int orc_legal(const OrcScene* s, int32_t tid, const float o[3], const float d[3], float dist,
              int32_t contact, int32_t inside); /* NoSelfIntersectionCondition.cpp:12-27 */
uint32_t orc_rng_draw(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t dim);
void orc_camera_ray(const OrcCamera* cam, uint32_t pixel, float u, float v, float dir_out[3]);
void orc_hemi_cos(float expTerm, float u1, float u2, float out[3]); /* RandGenerator.cpp:152-168 */
/* the math layer, one function each, pinned bit for bit against the reference's own headers
 * compiled in place (oracle/ref_math.cpp -> tests/golden/ref_math.npz) */
void orc_local_frame(const float n[3], float s_out[3], float t_out[3]); /* LocalFrame.h:14-23 */
void orc_to_world(const float n[3], const float a[3], float out[3]);    /* LocalFrame.h:25-28 */
float orc_hg_eval(float g, const float in[3], const float outd[3]);     /* HGPhaseFunc.h:11-17 */
void orc_face_normal(const float p0[3], const float p1[3], const float p2[3], float out[3]);
void orc_interp_normal(const float p0[3], const float p1[3], const float p2[3], const float n0[3],
                       const float n1[3], const float n2[3], const float position[3],
                       float out[3]); /* SimpleShape.cpp:311-314, 329-337 */
void orc_rotate(const float axis[3], float angle, const float v[3], float out[3]); /* smallFuncs.h:11-24 */
void orc_normalize(const float v[3], float out[3]); /* nvVector.h:522-526 */
void orc_div(const float v[3], float d, float out[3]); /* nvVector.h:376-380 */
float orc_dot(const float a[3], const float b[3]);  /* nvVector.h:501-507 */
float orc_length(const float a[3]);

/* PathTracer (PathTracer.cpp:12-44 + MCRenderer.cpp:149-224); rgb_sum[W*H*3] receives the SUM over
 * samples [spp_begin, spp_begin+spp_count); counter RNG keyed (seed, pixel, sample, dim).
 * n_rays (may be NULL): closest-hit queries issued. threads: OpenMP over pixels.               */
void orc_render_pt(const OrcScene* s, const OrcCamera* cam, uint32_t spp_begin, uint32_t spp_count,
                   uint32_t max_depth, uint32_t seed, float* rgb_sum, uint64_t* n_rays,
                   int threads);
/* PathTracerTest (PathTracerTest.cpp:7-135): the same paths plus one light connection per RANDOM
 * vertex, balance-heuristic MIS; no vignetting compensation (the reference applies none there).
 * Connection draws come from a second stream of the same (seed, pixel, sample): key' =
 * lowbias32(key ^ 0x5bd1e995), draw 6*vertexIndex + j. Returns -1 for scenes that hold a medium. */
int orc_render_pttest(const OrcScene* s, const OrcCamera* cam, uint32_t spp_begin, uint32_t spp_count,
                      uint32_t max_depth, uint32_t seed, float* rgb_sum, uint64_t* n_rays,
                      uint64_t* n_shadow_rays, int threads);
/* one path, verbose: writes up to cap vertices (origin xyz, dir xyz, color rgb, dirProb) = 10
 * floats each; returns the vertex count and the path's contribution in rgb[3].                 */
int32_t orc_trace_path(const OrcScene* s, const OrcCamera* cam, uint32_t pixel, uint32_t sample,
                       uint32_t max_depth, uint32_t seed, float* verts, int32_t cap, float rgb[3]);

/* Scene::findInsideObject (Scene.cpp:354-390) for one ray; current = object to skip or -1.
 * Returns the object index or -1. */
int32_t orc_find_inside(const OrcScene* s, const float o[3], const float d[3], int32_t current);

#ifdef __cplusplus
}
#endif
#endif
