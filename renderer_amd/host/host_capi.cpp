// host_capi.cpp — the extern "C" shims of include/mirt_host.h over the C++ host surface.
#include <cstring>
#include <stdexcept>
#include <string>

#include "../../include/mirt_host.h"
#include "host.h"

using namespace mirt_host;

struct MirthScene {
    Renderer renderer;
};

static thread_local std::string g_err;

#define GUARD(...)                       \
    try {                                \
        __VA_ARGS__;                     \
        return 0;                        \
    } catch (const std::exception& e) {  \
        g_err = e.what();                \
        return -1;                       \
    }

extern "C" {

const char* mirth_last_error(void) { return g_err.c_str(); }

int mirth_load_config(const char* config_path, MirthScene** out) {
    if (!config_path || !out) {
        g_err = "mirth_load_config: NULL argument";
        return -1;
    }
    *out = nullptr;
    MirthScene* s = new MirthScene;
    s->renderer.uploadOnLoad = false;
    try {
        s->renderer.loadConfig(config_path);
    } catch (const std::exception& e) {
        g_err = e.what();
        delete s;
        return -1;
    }
    *out = s;
    return 0;
}

int mirth_scene_from_arrays(int32_t n_verts, const float* verts, int32_t n_tris, const uint32_t* tris,
                            const int32_t* tri_obj, const float* tri_vn, int32_t n_objects,
                            const MirtObject* objects, MirthScene** out) {
    if (!verts || !tris || !tri_obj || !objects || !out || n_verts <= 0 || n_tris <= 0 || n_objects <= 0) {
        g_err = "mirth_scene_from_arrays: NULL or empty argument";
        return -1;
    }
    *out = nullptr;
    MirthScene* s = new MirthScene;
    try {
        Scene& sc = s->renderer.scene;
        // one SceneObject per object id; each keeps a copy of the shared vertex list (identity
        // transform), so world positions equal the input exactly
        int prev = -1;
        for (int32_t t = 0; t < n_tris; t++) {
            int o = tri_obj[t];
            if (o < 0 || o >= n_objects || o < prev || o > prev + 1)
                throw std::runtime_error("mirth_scene_from_arrays: tri_obj must be ordered 0..n_objects-1 without gaps");
            prev = o;
            for (int k = 0; k < 3; k++)
                if (tris[3 * t + k] >= (uint32_t)n_verts)
                    throw std::runtime_error("mirth_scene_from_arrays: vertex index out of range");
        }
        if (prev != n_objects - 1) throw std::runtime_error("mirth_scene_from_arrays: an object has no triangles");
        // Build the flattened scene directly (the per-object meshes would only be re-concatenated).
        sc.tree.destroy();
        sc.tree.vertexPositionList.resize(n_verts);
        for (int32_t i = 0; i < n_verts; i++)
            sc.tree.vertexPositionList[i] = vec3f(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);
        sc.objectDescs.assign(objects, objects + n_objects);
        for (auto& m : sc.objectDescs) m.tri_count = 0;
        sc.tree.triangleList.resize(n_tris);
        sc.triPos.resize(9 * (size_t)n_tris);
        sc.triObj.assign(tri_obj, tri_obj + n_tris);
        sc.triLocal.resize(n_tris);
        sc.triVn.assign(9 * (size_t)n_tris, 0.f);
        if (tri_vn) memcpy(sc.triVn.data(), tri_vn, sizeof(float) * 9 * (size_t)n_tris);
        for (int32_t t = 0; t < n_tris; t++) {
            MirtObject& m = sc.objectDescs[tri_obj[t]];
            if (m.tri_count == 0) m.tri_begin = t;
            KDTree::Triangle& tr = sc.tree.triangleList[t];
            tr.vertexIndices = vec3ui{tris[3 * t], tris[3 * t + 1], tris[3 * t + 2]};
            tr.sourceInformation.objID = (unsigned)tri_obj[t];
            tr.sourceInformation.triangleID = (unsigned)m.tri_count;
            sc.triLocal[t] = m.tri_count;
            m.tri_count++;
            for (int k = 0; k < 3; k++)
                for (int c = 0; c < 3; c++) sc.triPos[9 * (size_t)t + 3 * k + c] = verts[3 * tris[3 * t + k] + c];
        }
        for (int32_t o = 0; o < n_objects; o++) {
            sc.objTriangleOffsetMap.push_back((unsigned)sc.objectDescs[o].tri_begin);
            if (!tri_vn) sc.objectDescs[o].has_vn = 0;
        }
        sc.tree.build();
        sc.buildObjKDTrees();
        sc.touch();
    } catch (const std::exception& e) {
        g_err = e.what();
        delete s;
        return -1;
    }
    *out = s;
    return 0;
}

int mirth_rebuild_tree(MirthScene* s, int32_t strategy) {
    if (!s || (strategy != KDTree::LOOP && strategy != KDTree::BEST)) {
        g_err = "mirth_rebuild_tree: NULL scene or unknown strategy (0 = LOOP, 1 = BEST)";
        return -1;
    }
    try {
        s->renderer.scene.tree.build((KDTree::Strategy)strategy);
        s->renderer.scene.touch();  // a resident copy is stale now
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
    return 0;
}

int mirth_set_textures(MirthScene* s, const float* tri_vt, int32_t n_textures, const MirtTexture* textures,
                       const MirtObjectTex* object_tex) {
    if (!s || n_textures < 0 || (n_textures > 0 && (!textures || !object_tex))) {
        g_err = "mirth_set_textures: NULL argument";
        return -1;
    }
    Scene& sc = s->renderer.scene;
    const size_t T = sc.tree.triangleList.size();
    for (int k = 0; k < n_textures; k++)
        if (textures[k].width <= 0 || textures[k].height <= 0 || !textures[k].texels ||
            (textures[k].channels != 1 && textures[k].channels != 3)) {
            g_err = "mirth_set_textures: bad texture";
            return -1;
        }
    for (size_t o = 0; n_textures > 0 && o < sc.objectDescs.size(); o++)
        for (int k : {object_tex[o].diffuse, object_tex[o].glossy_color, object_tex[o].glossy_coeff, object_tex[o].bump})
            if (k < -1 || k >= n_textures) {
                g_err = "mirth_set_textures: object_tex index out of range";
                return -1;
            }
    sc.triVt.clear();
    if (tri_vt) sc.triVt.assign(tri_vt, tri_vt + 9 * T);
    sc.textures.clear();
    sc.textureDescs.clear();
    sc.objectTex.clear();
    for (int k = 0; k < n_textures; k++) {
        const MirtTexture& X = textures[k];
        std::shared_ptr<Texture> t(new Texture);
        t->width = X.width;
        t->height = X.height;
        t->hasColor = X.channels == 3;
        const size_t n = (size_t)X.width * X.height;
        t->colors.clear();
        if (t->hasColor) {
            t->colors.resize(n);
            for (size_t i = 0; i < n; i++) t->colors[i] = vec3f(X.texels[3 * i], X.texels[3 * i + 1], X.texels[3 * i + 2]);
        } else {
            t->values.assign(X.texels, X.texels + n);
        }
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) {
                t->coordTransform(r, c) = X.coord_transform[4 * r + c];
                t->colorTransform(r, c) = X.color_transform[4 * r + c];
            }
        sc.textures.push_back(t);
        sc.textureDescs.push_back(t->describe());
    }
    if (n_textures > 0) sc.objectTex.assign(object_tex, object_tex + sc.objectDescs.size());
    sc.touch();  // a copy on the device was uploaded without these textures
    return 0;
}

void mirth_free(MirthScene* s) { delete s; }

int mirth_counts(const MirthScene* s, int32_t out[5]) {
    if (!s || !out) return -1;
    const Scene& sc = s->renderer.scene;
    out[0] = (int32_t)sc.tree.vertexPositionList.size();
    out[1] = (int32_t)sc.tree.triangleList.size();
    out[2] = (int32_t)sc.objectDescs.size();
    out[3] = sc.tree.numNodes();
    out[4] = sc.tree.depth;
    return 0;
}

int mirth_get_raw(const MirthScene* s, float* verts, uint32_t* tris, int32_t* tri_obj, int32_t* tri_local,
                  float* tri_vn, MirtObject* objects) {
    if (!s) return -1;
    const Scene& sc = s->renderer.scene;
    const size_t V = sc.tree.vertexPositionList.size(), T = sc.tree.triangleList.size();
    if (verts)
        for (size_t i = 0; i < V; i++)
            for (int c = 0; c < 3; c++) verts[3 * i + c] = sc.tree.vertexPositionList[i][c];
    if (tris)
        for (size_t i = 0; i < T; i++)
            for (int c = 0; c < 3; c++) tris[3 * i + c] = sc.tree.triangleList[i].vertexIndices[c];
    if (tri_obj) memcpy(tri_obj, sc.triObj.data(), sizeof(int32_t) * T);
    if (tri_local) memcpy(tri_local, sc.triLocal.data(), sizeof(int32_t) * T);
    if (tri_vn) memcpy(tri_vn, sc.triVn.data(), sizeof(float) * 9 * T);
    if (objects) memcpy(objects, sc.objectDescs.data(), sizeof(MirtObject) * sc.objectDescs.size());
    return 0;
}

int mirth_get_tree(const MirthScene* s, float* node_min, float* node_max, int32_t* node_right,
                   int32_t* leaf_begin, int32_t* leaf_end, int32_t* slot_tid) {
    if (!s) return -1;
    const KDTree& t = s->renderer.scene.tree;
    const size_t N = t.nodeRight.size();
    if (node_min) memcpy(node_min, t.nodeMin.data(), sizeof(float) * 3 * N);
    if (node_max) memcpy(node_max, t.nodeMax.data(), sizeof(float) * 3 * N);
    if (node_right) memcpy(node_right, t.nodeRight.data(), sizeof(int32_t) * N);
    if (leaf_begin) memcpy(leaf_begin, t.nodeLeafBegin.data(), sizeof(int32_t) * N);
    if (leaf_end) memcpy(leaf_end, t.nodeLeafEnd.data(), sizeof(int32_t) * N);
    if (slot_tid) memcpy(slot_tid, t.slotTid.data(), sizeof(int32_t) * t.slotTid.size());
    return 0;
}

int mirth_get_camera(const MirthScene* s, MirtCamera* cam) {
    if (!s || !cam) return -1;
    *cam = s->renderer.camera.toMirt();
    return 0;
}

int mirth_get_render_cfg(const MirthScene* s, uint32_t out[3]) {
    if (!s || !out) return -1;
    const PathTracerHIP* pt = dynamic_cast<const PathTracerHIP*>(s->renderer.mcRenderer);
    out[0] = pt ? (uint32_t)pt->getMaxDepth() : 20u;
    out[1] = pt ? pt->spp : 0u;
    out[2] = pt ? pt->seed : 1u;
    return 0;
}

int mirth_describe(const MirthScene* s, MirtSceneDesc* desc) {
    if (!s || !desc) return -1;
    *desc = s->renderer.scene.describe();
    return 0;
}

int mirth_upload(MirthScene* s, MirtContext** ctx_out) {
    if (!s) return -1;
    GUARD({
        IntersectionHIP::init();
        IntersectionHIP::loadKDTreeToDevice(s->renderer.scene);
        if (ctx_out) *ctx_out = IntersectionHIP::context();
    })
}

int mirth_render(MirthScene* s, uint32_t spp, uint32_t seed, float* rgb_mean, MirtRenderStats* stats) {
    if (!s || !rgb_mean) return -1;
    GUARD({
        IntersectionHIP::init();
        IntersectionHIP::loadKDTreeToDevice(s->renderer.scene);  // the service holds one scene at a time
        if (!s->renderer.mcRenderer) s->renderer.mcRenderer = new PathTracerHIP(&s->renderer);
        PathTracerHIP* pt = dynamic_cast<PathTracerHIP*>(s->renderer.mcRenderer);
        if (!pt) throw std::runtime_error("mirth_render: the configured integrator is not the path tracer");
        if (spp) pt->spp = spp;
        if (seed) pt->seed = seed;
        std::vector<vec3f> px = s->renderer.render();
        for (size_t i = 0; i < px.size(); i++) {
            rgb_mean[3 * i] = px[i].x;
            rgb_mean[3 * i + 1] = px[i].y;
            rgb_mean[3 * i + 2] = px[i].z;
        }
        if (stats) *stats = pt->lastStats;
    })
}

int mirth_render_progressive(MirthScene* s, uint32_t spp, uint32_t seed, int32_t output_iter, const char* resume_from,
                             const char* save_state, float* rgb_mean, MirtRenderStats* stats) {
    if (!s || !rgb_mean) return -1;
    GUARD({
        IntersectionHIP::init();
        IntersectionHIP::loadKDTreeToDevice(s->renderer.scene);
        if (!s->renderer.mcRenderer) s->renderer.mcRenderer = new PathTracerHIP(&s->renderer);
        PathTracerHIP* pt = dynamic_cast<PathTracerHIP*>(s->renderer.mcRenderer);
        if (!pt) throw std::runtime_error("mirth_render_progressive: the configured integrator is not a path tracer");
        if (spp) pt->spp = spp;
        if (seed) pt->seed = seed;
        const bool was = pt->progressive;
        const int old_iter = pt->getOutputIter();
        if (output_iter > 0) {
            pt->progressive = true;
            pt->setOutputIter(output_iter);
        }
        if (resume_from && *resume_from) pt->loadState(resume_from, s->renderer.camera);
        std::vector<vec3f> px = s->renderer.render();
        pt->progressive = was;
        pt->setOutputIter(old_iter);
        if (save_state && *save_state) pt->saveState(save_state, s->renderer.camera);
        for (size_t i = 0; i < px.size(); i++) {
            rgb_mean[3 * i] = px[i].x;
            rgb_mean[3 * i + 1] = px[i].y;
            rgb_mean[3 * i + 2] = px[i].z;
        }
        if (stats) *stats = pt->lastStats;
    })
}

int mirth_fill_intersect(MirthScene* s, size_t n, const float* o, const float* d, const int32_t* contact,
                         const int32_t* inside, float* dist, int32_t* tid, int32_t* obj, int32_t* local) {
    if (!s || !o || !d || !dist || !tid) return -1;
    GUARD({
        Scene& sc = s->renderer.scene;
        if (sc.objects.empty()) throw std::runtime_error("mirth_fill_intersect: needs a scene loaded from a config");
        std::vector<Ray> rays(n);
        for (size_t i = 0; i < n; i++) {
            rays[i].origin = vec3f(o[3 * i], o[3 * i + 1], o[3 * i + 2]);
            rays[i].direction = vec3f(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
            int c = contact ? contact[i] : -1, in = inside ? inside[i] : -1;
            if (c >= (int)sc.objects.size() || in >= (int)sc.objects.size())
                throw std::runtime_error("mirth_fill_intersect: object id out of range");
            rays[i].contactObject = c >= 0 ? sc.objects[c] : nullptr;
            rays[i].insideObject = in >= 0 ? sc.objects[in] : nullptr;
        }
        sc.fillIntersectObject(rays);
        for (size_t i = 0; i < n; i++) {
            bool hit = rays[i].intersectObject != nullptr;
            dist[i] = hit ? rays[i].intersectDist : -1.f;
            tid[i] = hit ? rays[i].intersect_tid : -1;
            if (obj) obj[i] = hit ? rays[i].intersectObject->objectIndex : -1;
            if (local) local[i] = hit ? (int32_t)rays[i].intersectObjectTriangleID : -1;
        }
    })
}

int mirth_test_visibility(MirthScene* s, size_t n, const float* o, const float* d, const int32_t* contact,
                          const int32_t* inside, const float* dist, uint8_t* visible) {
    if (!s || !o || !d || !dist || !visible) return -1;
    GUARD({
        Scene& sc = s->renderer.scene;
        if (sc.objects.empty()) throw std::runtime_error("mirth_test_visibility: needs a scene loaded from a config");
        std::vector<Ray> rays(n);
        for (size_t i = 0; i < n; i++) {
            rays[i].origin = vec3f(o[3 * i], o[3 * i + 1], o[3 * i + 2]);
            rays[i].direction = vec3f(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
            int c = contact ? contact[i] : -1, in = inside ? inside[i] : -1;
            if (c >= (int)sc.objects.size() || in >= (int)sc.objects.size())
                throw std::runtime_error("mirth_test_visibility: object id out of range");
            rays[i].contactObject = c >= 0 ? sc.objects[c] : nullptr;
            rays[i].insideObject = in >= 0 ? sc.objects[in] : nullptr;
            rays[i].intersectDist = dist[i];
        }
        std::vector<bool> v = sc.testVisibility(rays);
        for (size_t i = 0; i < n; i++) visible[i] = v[i] ? 1 : 0;
    })
}

int mirth_test_path_list_visibility(MirthScene* s, int32_t n_paths, const int32_t* start_off, const float* start_o,
                                    const int32_t* start_contact, const int32_t* start_inside,
                                    const uint8_t* start_random, const int32_t* end_off, const float* end_o,
                                    const uint8_t* end_random, int32_t* bits_off, uint32_t* bits,
                                    size_t bits_capacity) {
    if (!s || n_paths < 0 || !start_off || !start_o || !start_random || !end_off || !end_o || !end_random ||
        !bits_off || !bits)
        return -1;
    GUARD({
        Scene& sc = s->renderer.scene;
        if (sc.objects.empty()) throw std::runtime_error("mirth_test_path_list_visibility: needs a scene loaded from a config");
        if (!s->renderer.mcRenderer) throw std::runtime_error("mirth_test_path_list_visibility: the config names no renderer");
        std::vector<MCRenderer::Path> sp(n_paths), ep(n_paths);
        for (int32_t p = 0; p < n_paths; p++) {
            for (int32_t i = start_off[p]; i < start_off[p + 1]; i++) {
                Ray r;
                r.origin = vec3f(start_o[3 * i], start_o[3 * i + 1], start_o[3 * i + 2]);
                const int c = start_contact ? start_contact[i] : -1, in = start_inside ? start_inside[i] : -1;
                if (c >= (int)sc.objects.size() || in >= (int)sc.objects.size())
                    throw std::runtime_error("mirth_test_path_list_visibility: object id out of range");
                r.contactObject = c >= 0 ? sc.objects[c] : nullptr;
                r.insideObject = in >= 0 ? sc.objects[in] : nullptr;
                r.directionSampleType = start_random[i] ? Ray::RANDOM : Ray::DEFINITE;
                sp[p].push_back(r);
            }
            for (int32_t i = end_off[p]; i < end_off[p + 1]; i++) {
                Ray r;
                r.origin = vec3f(end_o[3 * i], end_o[3 * i + 1], end_o[3 * i + 2]);
                r.directionSampleType = end_random[i] ? Ray::RANDOM : Ray::DEFINITE;
                ep[p].push_back(r);
            }
        }
        const std::vector<std::vector<unsigned>> v = s->renderer.mcRenderer->testPathListVisibility(sp, ep);
        size_t at = 0;
        for (int32_t p = 0; p < n_paths; p++) {
            bits_off[p] = (int32_t)at;
            if (at + v[p].size() > bits_capacity) throw std::runtime_error("mirth_test_path_list_visibility: bits_capacity too small");
            for (unsigned w : v[p]) bits[at++] = w;
        }
        bits_off[n_paths] = (int32_t)at;
    })
}

// the host math layer, exposed so tests can pin it against the reference's own headers
void mirth_world_vertex(const float m[16], const float v[3], float out[3]) {
    matrix4 M;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) M(r, c) = m[r * 4 + c];
    vec3f p = M.apply(vec3f(v[0], v[1], v[2]), 1.f);
    out[0] = p.x, out[1] = p.y, out[2] = p.z;
}

void mirth_normal_transform(const float m[16], const float vn[3], float out[3]) {
    matrix4 M;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) M(r, c) = m[r * 4 + c];
    vec3f p = M.inverse().transposed().apply(vec3f(vn[0], vn[1], vn[2]), 0.f);
    out[0] = p.x, out[1] = p.y, out[2] = p.z;
}

void mirth_shutdown(void) { IntersectionHIP::shutdown(); }

int mirth_set_devices(const int32_t* devices, int32_t n) {
    if (n < 0 || (n > 0 && !devices)) return -1;
    GUARD({ IntersectionHIP::setDevices(std::vector<int>(devices, devices + n)); })
}

int mirth_is_resident(const MirthScene* s) { return s && IntersectionHIP::isResident(s->renderer.scene) ? 1 : 0; }

int mirth_save_pfm(const char* path, const float* rgb, int32_t w, int32_t h) {
    if (!path || !rgb || w <= 0 || h <= 0) return -1;
    GUARD({
        std::vector<vec3f> px((size_t)w * h);
        for (size_t i = 0; i < px.size(); i++) px[i] = vec3f(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
        MCRenderer::saveImagePFM(path, px, w, h);
    })
}

int mirth_load_pfm(const char* path, float* rgb, int32_t capacity_pixels, int32_t wh[2]) {
    if (!path || !wh) return -1;
    std::vector<vec3f> px;
    int w = 0, h = 0;
    if (!MCRenderer::loadImagePFM(path, px, w, h)) {
        g_err = std::string("mirth_load_pfm: cannot read ") + path;
        return -1;
    }
    wh[0] = w;
    wh[1] = h;
    if (rgb) {
        if ((long long)w * h > capacity_pixels) {
            g_err = "mirth_load_pfm: buffer too small";
            return -1;
        }
        for (size_t i = 0; i < px.size(); i++) {
            rgb[3 * i] = px[i].x;
            rgb[3 * i + 1] = px[i].y;
            rgb[3 * i + 2] = px[i].z;
        }
    }
    return 0;
}

}  // extern "C"
