/* mirt.h — C-ABI of libmirt.so, the MI355X (gfx950) closest-hit + wavefront path-tracing service.
 *
 * This is the drop-in boundary for ONE hot path of winmad/Renderer: batched closest-hit
 * ray/scene intersection over the flattened "KDTree" and the PathTracer per-pixel sample loop.
 * Every entry point names the reference interface it replaces (file:line under the reference tree).
 *
 * Conventions
 *   - every call returns int: MIRT_OK (0) or a negative MIRT_E_* code; mirt_last_error() gives text.
 *     The library never calls exit() (the reference does, Shader.cpp:9-12).
 *   - plain pointers and sizes only; the caller owns every host buffer; the library copies on upload
 *     and owns its device memory until mirt_destroy().
 *   - one context per GPU; calls on one context must be serialised by the caller; different
 *     contexts may be driven from different host threads.
 *   - ray / hit batches are SoA: `origins` and `directions` are plane-major float[3][n]
 *     (x-plane, y-plane, z-plane, each n floats) so that wave64 loads are fully coalesced.
 *   - a miss is t < 0 and tid = -1 (reference convention: Intersection.frag:83, Scene.cpp:443).
 */
#ifndef MIRT_H
#define MIRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIRT_OK            0
#define MIRT_E_INVALID    -1  /* bad argument / inconsistent scene description            */
#define MIRT_E_NO_DEVICE  -2  /* no HIP device, or HIP runtime error at context creation   */
#define MIRT_E_HIP        -3  /* HIP runtime error (text in mirt_last_error)               */
#define MIRT_E_NO_SCENE   -4  /* intersect / render called before mirt_upload_scene        */
#define MIRT_E_LIMIT      -5  /* a stated limit was exceeded (tree depth, object count...) */

/* object (material) kinds; they follow the reference's Scene*Object classes */
#define MIRT_OBJ_DIFFUSE     0  /* SceneDiffuseObject.h:5-21  + DiffuseMaterial.cpp:4-76     */
#define MIRT_OBJ_GLOSSY      1  /* SceneGlossyObject.h:5-19   + GlossyMaterial.cpp:4-56      */
#define MIRT_OBJ_REFRACTIVE  2  /* SceneRefractiveObject.h / SceneReflectiveObject.h (ior < 0
                                   = mirror) + RefractiveMaterial.cpp:4-104                  */
#define MIRT_OBJ_EMISSIVE    3  /* SceneEmissiveObject.cpp:4-19                              */
#define MIRT_OBJ_VPM         4  /* SceneVPMObject.cpp:4-233 (homogeneous medium)             */
#define MIRT_OBJ_PHONG       5  /* ScenePhongObject.h:7-23: a glossy and a diffuse material; every
                                   scatter picks one of the two with equal chance and does NOT divide
                                   by that chance (SceneObject.cpp:366-373); BSDFs add, pdfs average
                                   (SceneObject.cpp:227-248, 341-352). color = diffuse, color2 + coeff
                                   = glossy                                                        */

/* intersection modes */
#define MIRT_MODE_UNCONDITIONAL 0 /* KDTree::intersect(ray, tid, NULL), KDTree.cpp:335-339    */
#define MIRT_MODE_LAST_TID      1 /* skip triangle id == aux[i]; Intersection.frag:63-64      */
#define MIRT_MODE_NO_SELF       2 /* NoSelfIntersectionCondition::legal, .cpp:12-27; aux[i] =
                                     MIRT_PACK_IDS(contactObject, insideObject)                */

/* aux word for MIRT_MODE_NO_SELF: object ids in [-1, 65533]; -1 = NULL. */
#define MIRT_PACK_IDS(contact, inside) \
    ((int32_t)((uint32_t)(((contact) + 1) & 0xFFFF) | ((uint32_t)(((inside) + 1) & 0xFFFF) << 16)))
#define MIRT_MAX_OBJECTS 65533
#define MIRT_MAX_TREE_DEPTH 62 /* traversal stack levels available in LDS */

typedef struct MirtContext MirtContext;

/* One scene object = one material (the shipped configs use one Material per object). */
typedef struct MirtObject {
    int32_t type;          /* MIRT_OBJ_*                                                     */
    int32_t has_vn;        /* 1: smooth normals from per-corner vertex normals (tri_vn)      */
    int32_t volumetric;    /* SceneObject::isVolumetric()                                    */
    int32_t tri_begin;     /* first global triangle id of this object (Scene.cpp:270)        */
    int32_t tri_count;
    float   color[3];      /* diffuse / glossy albedo, emissive radiance, refractive surfColor */
    float   coeff;         /* glossy exponent (GlossyMaterial.h:20-23)                       */
    float   ior;           /* refrCoeff; < 0 = mirror; VPM: inrate; 1 otherwise              */
    float   decay[3];      /* refractive decayColor (RefractiveMaterial.h:15)                */
    float   sigma_s[3];    /* VPM                                                            */
    float   sigma_a[3];    /* VPM                                                            */
    float   g;             /* VPM Henyey-Greenstein g                                        */
    float   color2[3];     /* MIRT_OBJ_PHONG: colour of the glossy component                 */
} MirtObject;

/* One object's OWN tree (Scene::buildObjKDTrees, Scene.cpp:287-320), same layout as the scene
 * tree; needed for volumetric objects only, which Scene::findInsideObject (Scene.cpp:354-390)
 * queries one by one. slot_tid holds triangle ids LOCAL to the object.                          */
typedef struct MirtObjTree {
    int32_t  object;                /* index into MirtSceneDesc.objects                       */
    int32_t  n_nodes;
    const float*   node_min;
    const float*   node_max;
    const int32_t* node_right;
    const int32_t* node_leaf_begin;
    const int32_t* node_leaf_end;
    const int32_t* slot_tid;        /* [objects[object].tri_count]                            */
} MirtObjTree;

/* One texture (Texture.h:14-42): texels as Texture::loadFile (Texture.cpp:5-41) leaves them, row by row in
 * file order; the two matrices row-major, applied as M * (x, y, z, 1) (Texture.cpp:45, 79).
 * Lookup = Texture::getColor (Texture.cpp:43-94): bilinear, wrapping, texel centres at +0.5.       */
typedef struct MirtTexture {
    int32_t width, height;
    int32_t channels;           /* 3: rgb texels ("PF"); 1: one float per texel ("Pf", or after toGrayScale) */
    int32_t pad;
    const float* texels;        /* [width*height*channels]                                          */
    float coord_transform[16];  /* Texture::coordTransform                                          */
    float color_transform[16];  /* Texture::colorTransform (colorScale folded in, Texture.cpp:157-161) */
} MirtTexture;

/* Which textures an object's materials read, -1 = none (the constant colour / coeff of MirtObject).
 *   diffuse      DiffuseMaterial::tex       <Texture>       replaces color  (MIRT_OBJ_DIFFUSE, _PHONG)
 *   glossy_color GlossyMaterial::colorTex   <ColorTexture>  replaces color (MIRT_OBJ_GLOSSY) / color2 (_PHONG)
 *   glossy_coeff GlossyMaterial::coeffTex   <CoeffTexture>  replaces coeff (its .x)
 *   bump         SceneObject::bumpTex       <BumpTexture>   perturbs the shading normal, which the hit predicate
 *                of MIRT_MODE_NO_SELF reads too (SceneObject.cpp:250-283); a gray texture (ConfigManager.cpp:161);
 *                set it only for objects that have texture coordinates and a texture of more than one texel
 *                (SceneObject.cpp:253). Scenes with a bump texture run without the fast tree and cannot be
 *                instrumented (mirt_intersect_count_dev / _audit_dev refuse them).
 * Texture coordinates come from tri_vt through SimpleShape::getTexCoord (SimpleShape.cpp:274-298).
 * GlossyMaterial::scatter reads its two textures at getTexCoord(tid, (0,0,0)) (GlossyMaterial.cpp:11-12
 * use outRay.origin before it is set at :17); the device does the same.                            */
typedef struct MirtObjectTex {
    int32_t diffuse, glossy_color, glossy_coeff, bump;
} MirtObjectTex;

/* Flattened scene: what KDTree::serializeForGPU (KDTree.cpp:346-383) + Scene::buildKDTree
 * (Scene.cpp:248-275) hand to IntersectionGPU::loadKDTreeToShader (IntersectionGPU.cpp:243-255),
 * with integers kept as integers and the data the CPU hit predicate needs added.
 *   nodes    : pre-order; left child of inner node i is i+1; node_right[i] = right child or -1
 *              for a leaf; a leaf owns triangle SLOTS [node_leaf_begin[i], node_leaf_end[i]).
 *   slots    : triangles in leaf order (each triangle is in exactly one leaf, KDTree.cpp:208-224);
 *              slot_tid[s] = global triangle id (index into Scene's concatenated triangle list).
 *   per global triangle id: world-space corner positions, owning object, local triangle id and
 *              world-space (normal-matrix transformed, not normalised) corner normals.          */
typedef struct MirtSceneDesc {
    uint32_t struct_size;           /* sizeof(MirtSceneDesc), for versioning: a caller built
                                       against the shorter struct (no obj_trees) still works  */
    int32_t  n_nodes;
    const float*   node_min;        /* [n_nodes*3]                                            */
    const float*   node_max;        /* [n_nodes*3]                                            */
    const int32_t* node_right;      /* [n_nodes]                                              */
    const int32_t* node_leaf_begin; /* [n_nodes] (-1 for inner nodes)                         */
    const int32_t* node_leaf_end;   /* [n_nodes]                                              */
    int32_t  n_tris;
    const int32_t* slot_tid;        /* [n_tris]                                               */
    const float*   tri_pos;         /* [n_tris*9] v0 v1 v2 world positions, by global tid     */
    const int32_t* tri_obj;         /* [n_tris] object id, by global tid                      */
    const int32_t* tri_local;       /* [n_tris] triangle id inside its object                 */
    const float*   tri_vn;          /* [n_tris*9] corner normals by global tid; may be NULL if
                                       no object has has_vn                                    */
    int32_t  n_objects;
    const MirtObject* objects;      /* [n_objects]                                            */
    int32_t  n_obj_trees;           /* one per MIRT_OBJ_VPM object (required for rendering
                                       scenes that hold one; closest-hit queries do not use them) */
    const MirtObjTree* obj_trees;   /* [n_obj_trees]                                          */
    /* textures (optional tail: absent in a shorter struct_size, or all zero) */
    const float*   tri_vt;          /* [n_tris*9] corner texture coordinates (x y z each), by global
                                       tid; NULL when no object has any                          */
    int32_t  n_textures;
    const MirtTexture* textures;    /* [n_textures]                                           */
    const MirtObjectTex* object_tex; /* [n_objects] or NULL                                   */
} MirtSceneDesc;

/* Pinhole camera; fields as Camera.h:13-21 (sightDist is an unsigned int there). */
typedef struct MirtCamera {
    int32_t  width, height;
    uint32_t sight_dist;
    float    position[3];
    float    focus[3];
    float    up[3];
} MirtCamera;

typedef struct MirtRenderParams {
    MirtCamera camera;
    uint32_t spp_begin;      /* first global sample index rendered by this call              */
    uint32_t spp_count;      /* samples per pixel rendered by this call                      */
    uint32_t max_depth;      /* MCRenderer::maxDepth (default 20, MCRenderer.h:106); <= 250  */
    uint32_t seed;           /* RNG stream = f(seed, pixel, global sample index, dimension)  */
    uint32_t spp_per_launch; /* samples per pixel kept in flight per wavefront; 0 = auto     */
} MirtRenderParams;

/* A pixel window of the image (image tiles as a sharding axis: the north star's "samples-per-pixel and image tiles
 * shard across the GPUs"). Row 0 = top, like the image. No counterpart in the reference, which always renders the
 * whole image (PathTracer.cpp:17-18). */
typedef struct MirtTile {
    int32_t x0, y0;          /* top-left pixel of the window in the full image                 */
    int32_t width, height;   /* window size; x0 + width <= camera.width, y0 + height <= camera.height */
} MirtTile;

/* how mirt_render_pt_multi_shard / mirt_render_pttest_multi_shard cut the work over the contexts */
#define MIRT_SHARD_SAMPLES 0  /* contiguous sample-index shards of the whole image + one sum reduce */
#define MIRT_SHARD_TILES   1  /* every context renders ALL samples of its own tile (mirt_shard_tile) + one gather, no reduce */

typedef struct MirtRenderStats {
    uint64_t n_samples;      /* camera paths started                                         */
    uint64_t n_rays;         /* closest-hit queries issued (primary + secondary)             */
    double   gpu_ms;         /* device time of the whole call, HIP events                    */
    double   intersect_ms;   /* device time spent in the intersect kernel, HIP events        */
    uint64_t n_shadow_rays;  /* visibility queries issued (mirt_render_pttest only)          */
} MirtRenderStats;

/* --- context ------------------------------------------------------------------------------
 * replaces IntersectionGPU::init (IntersectionGPU.cpp:16-44: GL context + shader program).   */
int mirt_create(int device, MirtContext** out);
int mirt_destroy(MirtContext* ctx);
/* ctx may be NULL: then the last error of a failed mirt_create on this thread. */
const char* mirt_last_error(const MirtContext* ctx);

/* --- scene --------------------------------------------------------------------------------
 * replaces IntersectionGPU::loadKDTreeToShader (IntersectionGPU.cpp:243-255).                */
int mirt_upload_scene(MirtContext* ctx, const MirtSceneDesc* scene);
/* Number of successful mirt_upload_scene calls on this context so far (0: no scene). The reference keeps its
 * tree textures in static state that whoever called loadKDTreeToShader last has overwritten
 * (IntersectionGPU.cpp:5-14, 243-255); a caller that shares a context uses this to notice that the scene it
 * uploaded is no longer the resident one (IntersectionHIP::isResident in renderer_amd/host/scene.cpp).   */
uint64_t mirt_scene_serial(const MirtContext* ctx);

/* --- batched closest hit ------------------------------------------------------------------
 * replaces IntersectionGPU::query (IntersectionGPU.cpp:216-241) as called from
 * Scene::fillIntersectObject (Scene.cpp:431-453), and the CPU KDTree::intersect
 * (KDTree.cpp:295-339) as called from Scene::intersect (Scene.cpp:341-352).
 * origins, directions: float[3][n] plane-major; aux: int32[n] (ignored, may be NULL, in mode 0);
 * t_out: float[n]; tid_out: int32[n] global triangle id or -1. Host pointers.                */
int mirt_intersect(MirtContext* ctx, const float* origins, const float* directions,
                   const int32_t* aux, size_t n, int mode, float* t_out, int32_t* tid_out);
/* Same with DEVICE pointers; asynchronous on `stream` (a hipStream_t, NULL = context stream). It never synchronises; a
 * batch larger than any before allocates a larger list buffer (hipMalloc; the outgrown one is kept until the next
 * upload, so launches still in flight are unaffected). Every mode takes
 * the same route on a scene (the fast tree when the scene has one; MIRT_FAST_ALL_MODES=0 in the environment sends
 * modes 0 / 1 over the reference tree instead, for A/B runs; the answers are the same either way).            */
int mirt_intersect_dev(MirtContext* ctx, const float* origins, const float* directions,
                       const int32_t* aux, size_t n, int mode, float* t_out, int32_t* tid_out,
                       void* stream);
/* The same call for callers that CANNOT ALLOCATE (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc). It has no counterpart in the reference. It calls no hipMalloc / hipFree / event or
 * stream creation and never synchronises: the three scratch buffers mirt_intersect_dev keeps in hipMalloc'd
 * memory (the ray-queue words, the list of rays handed to the exact re-run, the traversal-stack levels beyond the
 * LDS share) live in ONE static device array of 48 MiB + 256 B that every GPU the library runs on holds until the
 * process ends. Limits: n <= MIRT_NOALLOC_MAX_RAYS per call (MIRT_E_LIMIT beyond it); the persistent grid is cut
 * down until its deep stack levels fit 32 MiB (same answers, fewer waves on very deep trees). Every call on a GPU shares that GPU's array (all contexts on it too): calls that overlap on one GPU RACE -
 * serialise them (one stream, or events). Results are bit-identical to mirt_intersect_dev's.              */
#define MIRT_NOALLOC_MAX_RAYS (4u << 20)
int mirt_intersect_noalloc_dev(MirtContext* ctx, const float* origins, const float* directions,
                               const int32_t* aux, size_t n, int mode, float* t_out, int32_t* tid_out,
                               void* stream);
/* --- batched visibility ---------------------------------------------------------------------
 * replaces MCRenderer::testVisibility (MCRenderer.cpp:106-118) for a batch: visible[i] = 1 when the
 * closest hit under `mode` (normally MIRT_MODE_NO_SELF, aux as above) does not exist or satisfies
 * dist - max_dist[i] >= -EPSILON (1e-5, macros.h:5), else 0; max_dist[i] is the distance between the
 * two vertices being connected. Evaluated as an any-hit query that stops at the first occluder and
 * never enters boxes beyond the limit; the answer is the reference's, not an approximation of it.
 * (The GL variant Scene::testVisibility, Scene.cpp:455-508, is two mirt_intersect calls in mode 1
 * plus its retest rule: see Scene::testVisibility in renderer_amd/host/scene.cpp.)
 * Host pointers / device pointers + stream, like mirt_intersect / mirt_intersect_dev.            */
int mirt_visibility(MirtContext* ctx, const float* origins, const float* directions, const int32_t* aux,
                    const float* max_dist, size_t n, int mode, uint8_t* visible);
int mirt_visibility_dev(MirtContext* ctx, const float* origins, const float* directions, const int32_t* aux,
                        const float* max_dist, size_t n, int mode, uint8_t* visible, void* stream);
/* The fast tree mirt_upload_scene builds for the predicate mode (no counterpart in the reference): a
 * binned surface-area hierarchy over the TRIANGLES of the scene tree, every leaf one triangle with the box
 * of its corners, in the same pre-order layout as MirtSceneDesc (left child = i + 1; node_right < 0 marks
 * a leaf, whose [leaf_begin, leaf_begin + 1) is the triangle's slot in the scene tree's slot order).
 * Host-only, needs no GPU: for tools and tests. Needs tri_pos and slot_tid of the descriptor. Returns
 * MIRT_E_LIMIT (with *n_nodes_out set) when capacity_nodes is too small; 2 * n_tris - 1 nodes at most.  */
int mirt_fast_tree(const MirtSceneDesc* scene, int32_t capacity_nodes, int32_t* n_nodes_out, float* node_min,
                   float* node_max, int32_t* node_right, int32_t* leaf_begin, int32_t* leaf_end);
/* Instrumented run (never timed): same results plus the number of node-box tests and
 * triangle tests summed over the batch (SURVEY.md §8(d) algorithmic-bytes formula). Device ptrs. */
int mirt_intersect_count_dev(MirtContext* ctx, const float* origins, const float* directions,
                             const int32_t* aux, size_t n, int mode, float* t_out,
                             int32_t* tid_out, uint64_t* n_box_tests, uint64_t* n_tri_tests);

/* Full audit of the same instrumented run: audit[0] node-box tests, [1] triangle tests, [2] boxes the
 * slab classifier could not prove (entered and flagged), [3] proven decisions that DISAGREE with the
 * exact reference box test evaluated beside them (must be 0), [4] boxes decided by the reference arithmetic
 * (the unproven ones plus every box of rays outside the classifier's model), [5] whole-wave inner steps, [6] whole-wave
 * triangle iterations, [7] whole-wave outer iterations (lane-utilisation diagnostics).              */
int mirt_intersect_audit_dev(MirtContext* ctx, const float* origins, const float* directions,
                             const int32_t* aux, size_t n, int mode, float* t_out,
                             int32_t* tid_out, uint64_t audit[8]);

/* --- camera rays --------------------------------------------------------------------------
 * replaces Camera::generateRay (Camera.cpp:4-24) for a batch: ray i = (sample s, pixel p) with
 * i = (s - spp_begin) * W*H + p. DEVICE pointers float[3][n], n = spp_count*W*H.             */
int mirt_generate_rays_dev(MirtContext* ctx, const MirtCamera* cam, uint32_t spp_begin,
                           uint32_t spp_count, uint32_t seed, float* origins, float* directions);

/* --- path tracing -------------------------------------------------------------------------
 * replaces PathTracer::renderPixels (PathTracer.cpp:4-58) + MCRenderer::samplePath
 * (MCRenderer.cpp:149-224). rgb_sum: float[W*H*3] (pixel-major r,g,b; row 0 = top), receives
 * the SUM over the call's samples of the vignetting-compensated path radiance; the caller
 * divides by the total spp (the reference keeps a running mean, PathTracer.cpp:23,42).
 * Host pointer version overwrites rgb_sum.                                                   */
int mirt_render_pt(MirtContext* ctx, const MirtRenderParams* params, float* rgb_sum,
                   MirtRenderStats* stats /* may be NULL */);
/* DEVICE pointer version ADDS into rgb_sum (so a multi-call / multi-GPU caller can reduce). */
int mirt_render_pt_dev(MirtContext* ctx, const MirtRenderParams* params, float* rgb_sum,
                       MirtRenderStats* stats /* may be NULL */);
/* replaces PathTracerTest::renderPixels + colorByConnectingLights (PathTracerTest.cpp:7-135), the
 * reference's cross-check integrator ("PTtest", ConfigManager.cpp:285-288): the same eye paths,
 * plus one connection to the emission sampler (Scene.cpp:91-98, SceneObject.cpp:137-160) per RANDOM
 * vertex, tested with MCRenderer::testVisibility (one any-hit launch per bounce) and combined with
 * emitter hits by the balance heuristic. No vignetting compensation (the reference applies none
 * there). Inside a homogeneous medium the estimator is the reference's as well: scattering events connect to the
 * light through the phase function (SceneVPMObject.cpp:239-267), segments decay by the transmittance, and the
 * distance-sampling density originProb is NOT divided out (PathTracerTest.cpp:70-74 never does), so there the image is
 * the reference PathTracerTest's, not the path tracer's. Same buffers / semantics as
 * mirt_render_pt / mirt_render_pt_dev; stats->n_shadow_rays counts the visibility queries.      */
int mirt_render_pttest(MirtContext* ctx, const MirtRenderParams* params, float* rgb_sum,
                       MirtRenderStats* stats /* may be NULL */);
int mirt_render_pttest_dev(MirtContext* ctx, const MirtRenderParams* params, float* rgb_sum,
                           MirtRenderStats* stats /* may be NULL */);

/* The same two integrators restricted to a pixel window (image-tile sharding). rgb_sum: DEVICE float[tile->height *
 * tile->width * 3], pixel-major inside the WINDOW (local row 0 = image row tile->y0); the call ADDS into it like
 * mirt_render_pt_dev. Camera rays, RNG streams (keyed by the pixel of the full image and the global sample index)
 * and the vignetting term are the full frame's, and samples are added to the accumulator one by one in sample order,
 * so tiles rendered separately and put side by side are BIT-EQUAL to the full-frame render from the same (zeroed)
 * accumulator, whatever the tile sizes and spp_per_launch (tests/test_gpu_tiles.py). tile == NULL: the whole image.
 * stats->n_samples counts the window's paths. No counterpart in the reference (PathTracer.cpp:17-18 loops over
 * every pixel); for callers that shard an image over GPUs by tiles instead of by samples.                          */
int mirt_render_pt_tile_dev(MirtContext* ctx, const MirtRenderParams* params, const MirtTile* tile, float* rgb_sum,
                            MirtRenderStats* stats /* may be NULL */);
int mirt_render_pttest_tile_dev(MirtContext* ctx, const MirtRenderParams* params, const MirtTile* tile, float* rgb_sum,
                                MirtRenderStats* stats /* may be NULL */);
/* The tiling MIRT_SHARD_TILES uses, for callers that run one process per GPU and gather the tiles themselves
 * (renderer_amd.shard_tiles): tile k of n_shards of a width x height image. The image is cut into a gx x gy grid
 * (gy = the largest divisor of n_shards with gy * gy <= n_shards, gx = n_shards / gy; k = row * gx + column), in steps
 * of 16 pixels where the image allows it (the camera pass walks a window in 16 x 16 blocks), the first cuts one step
 * wider when it does not divide. A tile may be empty (width or height 0) when there are more shards than pixels.
 * Returns MIRT_E_INVALID for non-positive sizes or k outside [0, n_shards).                                         */
int mirt_shard_tile(int32_t width, int32_t height, int32_t n_shards, int32_t k, MirtTile* out);

/* --- several GPUs -----------------------------------------------------------------------------
 * The same seam (MCRenderer::renderPixels as Renderer::render calls it, MCRenderer.h:114-117, Renderer.cpp:46-50) over
 * n_ctx contexts, normally one per GPU (several on one device work too). The reference has no counterpart: its only
 * parallelism is OpenMP over pixels (PathTracer.cpp:17). Every context must hold the same scene (mirt_upload_scene on
 * each). The call's samples [spp_begin, spp_begin + spp_count) are cut into n_ctx contiguous shards (the first
 * spp_count % n_ctx shards get one sample more); every context renders its shard into its own device accumulator, from
 * its own host thread; the accumulators are then summed on ctxs[0]'s device in context order (peer copy + add kernel:
 * W*H*3 floats per context, 12 MiB at 1024^2) and the total lands in the HOST buffer rgb_sum, which is overwritten.
 * Sample streams are keyed by the global sample index, so the image equals the one-context image up to fp32 summation
 * order. stats: n_samples / n_rays / n_shadow_rays / intersect_ms summed, gpu_ms = the slowest context's.
 * Calls on the contexts involved must not overlap with other calls on them.                                       */
int mirt_render_pt_multi(MirtContext* const* ctxs, int n_ctx, const MirtRenderParams* params, float* rgb_sum,
                         MirtRenderStats* stats /* may be NULL */);
int mirt_render_pttest_multi(MirtContext* const* ctxs, int n_ctx, const MirtRenderParams* params, float* rgb_sum,
                             MirtRenderStats* stats /* may be NULL */);
/* The same with the sharding axis chosen by the caller: shard = MIRT_SHARD_SAMPLES is mirt_render_pt_multi /
 * mirt_render_pttest_multi; shard = MIRT_SHARD_TILES gives context k the tile mirt_shard_tile(width, height, n_ctx, k)
 * and ALL the call's samples, renders every tile into its own device accumulator (mirt_render_pt_tile_dev) from its
 * own host thread, and GATHERS the tiles into the host image rgb_sum (overwritten): no reduce, and the image is
 * bit-equal to the one-context image. The scene is replicated either way. stats as above.                          */
int mirt_render_pt_multi_shard(MirtContext* const* ctxs, int n_ctx, const MirtRenderParams* params, int shard,
                               float* rgb_sum, MirtRenderStats* stats /* may be NULL */);
int mirt_render_pttest_multi_shard(MirtContext* const* ctxs, int n_ctx, const MirtRenderParams* params, int shard,
                                   float* rgb_sum, MirtRenderStats* stats /* may be NULL */);
/* the device index a context was created on (mirt_create's argument); -1 for NULL */
int mirt_context_device(const MirtContext* ctx);
/* number of HIP devices (0 when there is none; never an error) */
int mirt_device_count(void);

/* Secondary-ray capture for the closest-hit benchmark: run the path tracer
 * for spp_count samples and copy out every secondary ray (origin, direction, packed ids) issued at
 * bounce depths [depth_lo, depth_hi]. DEVICE pointers; capacity in rays; *n_captured on host.  */
int mirt_capture_rays_dev(MirtContext* ctx, const MirtRenderParams* params, uint32_t depth_lo,
                          uint32_t depth_hi, float* origins, float* directions, int32_t* aux,
                          size_t capacity, size_t* n_captured);

/* device / library facts for logs */
int mirt_device_info(MirtContext* ctx, char* buf, size_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* MIRT_H */
