/* mirt_host.h — C shims over the C++ host surface (renderer_amd/host/host.h) so that tests and
 * bench.py can drive the same code a C++ user of Scene / ConfigManager / Renderer drives.
 * None of these compute on the CPU: scene loading, tree build and flattening are host work in the
 * reference too (ConfigManager.cpp:166-384, Scene.cpp:248-285, KDTree.cpp:129-246, 346-383);
 * mirth_upload / mirth_render go to the GPU through include/mirt.h.
 * Every call returns 0 on success, -1 on failure with text in mirth_last_error().
 */
#ifndef MIRT_HOST_H
#define MIRT_HOST_H
#include "mirt.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct MirthScene MirthScene;

const char* mirth_last_error(void);

/* Renderer::loadConfig (Renderer.cpp:81-91) without touching the GPU. */
int mirth_load_config(const char* config_path, MirthScene** out);
/* A scene from arrays (procedural meshes): world-space vertices, global-index triangles, owning
 * object per triangle (object triangle ranges must be contiguous and ordered), optional corner
 * normals [T*9]; builds the tree like Scene::buildKDTree. */
int mirth_scene_from_arrays(int32_t n_verts, const float* verts, int32_t n_tris, const uint32_t* tris,
                            const int32_t* tri_obj, const float* tri_vn, int32_t n_objects,
                            const MirtObject* objects, MirthScene** out);
/* KDTree::build(Strategy) (KDTree.cpp:230-246, KDTree.h:24) on the scene's tree: strategy 0 = LOOP (splitLoop, the split
 * axis cycles x, y, z with the depth), 1 = BEST (splitBest, what loading builds and every reference caller uses).
 * Any resident copy becomes stale (the next upload sends the new tree). The answers of the closest-hit query do not
 * depend on the strategy except where two triangles tie (the traversal order decides those).                        */
int mirth_rebuild_tree(MirthScene* s, int32_t strategy);
/* Gives a scene made from arrays its textures (copied): corner texture coordinates [T*9] by global triangle
 * id (may be NULL), the textures, and which of them each object's materials read [n_objects]
 * (MirtTexture / MirtObjectTex, include/mirt.h). A scene loaded from a config has them from its XML.   */
int mirth_set_textures(MirthScene* s, const float* tri_vt, int32_t n_textures, const MirtTexture* textures,
                       const MirtObjectTex* object_tex);
void mirth_free(MirthScene* s);

/* sizes: n_verts, n_tris, n_objects, n_nodes, tree depth */
int mirth_counts(const MirthScene* s, int32_t out[5]);
/* raw (pre-tree) scene, caller-allocated arrays sized from mirth_counts; any pointer may be NULL */
int mirth_get_raw(const MirthScene* s, float* verts, uint32_t* tris, int32_t* tri_obj,
                  int32_t* tri_local, float* tri_vn, MirtObject* objects);
/* flattened tree (pre-order) */
int mirth_get_tree(const MirthScene* s, float* node_min, float* node_max, int32_t* node_right,
                   int32_t* leaf_begin, int32_t* leaf_end, int32_t* slot_tid);
int mirth_get_camera(const MirthScene* s, MirtCamera* cam);
/* integrator settings read from the config: max_depth, spp, seed */
int mirth_get_render_cfg(const MirthScene* s, uint32_t out[3]);
/* the descriptor the device upload takes; pointers live as long as the scene */
int mirth_describe(const MirthScene* s, MirtSceneDesc* desc);

/* Scene::buildKDTree's upload step: IntersectionHIP::init + loadKDTreeToDevice; *ctx_out receives
 * the service's context so callers can use the mirt_* entry points on the same scene. */
int mirth_upload(MirthScene* s, MirtContext** ctx_out);
/* Renderer::render (Renderer.cpp:46-50) through PathTracerHIP::renderPixels: mean image
 * rgb[W*H*3]; spp/seed 0 = take the config's. */
int mirth_render(MirthScene* s, uint32_t spp, uint32_t seed, float* rgb_mean, MirtRenderStats* stats);
/* The same with the reference's progressive output (PathTracer.cpp:47-55, MCRenderer::showCurrentResult,
 * MCRenderer.cpp:396-437): output_iter > 0 dumps the running mean to savePath and to
 * "<savePath stem>_iter_<s>_time_<seconds>.pfm" after every pass s with s % output_iter == 0 (the series
 * compare.cpp:167-251 reads). resume_from / save_state (may be NULL): accumulator files (radiance sums +
 * pass count) so that a render can be continued later; `spp` is the TOTAL number of passes wanted. */
int mirth_render_progressive(MirthScene* s, uint32_t spp, uint32_t seed, int32_t output_iter, const char* resume_from,
                             const char* save_state, float* rgb_mean, MirtRenderStats* stats);
/* Scene::fillIntersectObject (Scene.cpp:431-453) on AoS rays: o[n*3], d[n*3], contact/inside
 * object ids (-1 = NULL) -> dist[n], global tid[n] (-1 = miss), objID[n], local triangle id[n] */
int mirth_fill_intersect(MirthScene* s, size_t n, const float* o, const float* d, const int32_t* contact,
                         const int32_t* inside, float* dist, int32_t* tid, int32_t* obj, int32_t* local);
/* Scene::testVisibility (Scene.cpp:455-508) on AoS rays: dist[n] = distance to the vertex being
 * connected (Ray::intersectDist there) -> visible[n] (1 = unoccluded, MCRenderer.cpp:115-117 rule) */
int mirth_test_visibility(MirthScene* s, size_t n, const float* o, const float* d, const int32_t* contact,
                          const int32_t* inside, const float* dist, uint8_t* visible);
/* MCRenderer::testPathListVisibility (MCRenderer.cpp:520-570) on flattened path lists: path pair p owns start
 * vertices [start_off[p], start_off[p+1]) and end vertices [end_off[p], end_off[p+1]); per vertex its origin
 * [*,3] and whether its direction is RANDOM (1) or DEFINITE (0); start vertices also their contact / inside object
 * ids (-1 = NULL). bits_off[p] (n_paths + 1 entries, filled here) gives the first 32-bit word of pair p in
 * bits[]; bit j * |end| + k of a pair = connection (start j, end k) unoccluded. bits_capacity in words.       */
int mirth_test_path_list_visibility(MirthScene* s, int32_t n_paths, const int32_t* start_off, const float* start_o,
                                    const int32_t* start_contact, const int32_t* start_inside,
                                    const uint8_t* start_random, const int32_t* end_off, const float* end_o,
                                    const uint8_t* end_random, int32_t* bits_off, uint32_t* bits,
                                    size_t bits_capacity);
/* Host math used while flattening a scene: transform * (v,1) (SimpleShape.h:46) and
 * transpose(inverse(transform)) * (vn,0) (SimpleShape.cpp:310,322); m[r*4+c] = element(r,c). */
void mirth_world_vertex(const float m[16], const float v[3], float out[3]);
void mirth_normal_transform(const float m[16], const float vn[3], float out[3]);
void mirth_shutdown(void);
/* Renders go to these devices from now on (IntersectionHIP::setDevices: devices[0] is the service's own device, a
 * device may be listed twice; n = 0 or 1: single GPU): sample shards + one reduce through mirt_render_pt_multi.    */
int mirth_set_devices(const int32_t* devices, int32_t n);
/* 1 when this scene is the one the shared device service holds (IntersectionHIP::isResident: same Scene, same
 * build of its tree and textures, nobody uploaded through the context since), else 0. Queries made through a
 * scene check this themselves and upload again when needed; exported for tests and tools.            */
int mirth_is_resident(const MirthScene* s);

int mirth_save_pfm(const char* path, const float* rgb, int32_t w, int32_t h);
int mirth_load_pfm(const char* path, float* rgb, int32_t capacity_pixels, int32_t wh[2]);

#ifdef __cplusplus
}
#endif
#endif
