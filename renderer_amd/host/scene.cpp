// scene.cpp — Scene (tree build + flattening for the device), Camera, IntersectionHIP.
#include <cstring>
#include <stdexcept>

#include "host.h"

namespace mirt_host {

MirtCamera Camera::toMirt() const {
    MirtCamera c;
    c.width = width;
    c.height = height;
    c.sight_dist = sightDist;
    for (int k = 0; k < 3; k++) {
        c.position[k] = position[k];
        c.focus[k] = focus[k];
        c.up[k] = up[k];
    }
    return c;
}

// ---- IntersectionHIP: the static service IntersectionGPU was (IntersectionGPU.cpp:5-14) -----------
static MirtContext* g_ctx = nullptr;
static const Scene* g_resident = nullptr;        // the scene whose tree is on the device ...
static unsigned long long g_resident_gen = 0;    // ... and which build of it
static uint64_t g_resident_serial = 0;           // mirt_scene_serial right after that upload (someone else may upload through the context)
static unsigned long long g_next_generation = 0; // stamps handed to Scene::buildKDTree

static std::vector<MirtContext*> g_all;  // g_ctx first, then the extra contexts of setDevices

void IntersectionHIP::init(int device) {
    if (g_ctx) return;
    int rc = mirt_create(device, &g_ctx);
    if (rc != MIRT_OK)
        throw std::runtime_error(std::string("IntersectionHIP::init: ") + mirt_last_error(nullptr));
    g_all.assign(1, g_ctx);
}

void IntersectionHIP::setDevices(const std::vector<int>& devices) {
    // the service's first context must sit on devices[0]: one created earlier on another device (init(), a previous
    // setDevices) is replaced, so that batch queries and the reduce run where the caller was told they run
    const int first = devices.empty() ? 0 : devices[0];
    if (g_ctx && mirt_context_device(g_ctx) != first) shutdown();
    init(first);
    for (size_t k = 1; k < g_all.size(); k++) mirt_destroy(g_all[k]);
    g_all.assign(1, g_ctx);
    g_resident = nullptr;  // the new contexts hold no scene yet: the next query / render uploads to all
    g_resident_gen = 0;
    for (size_t k = 1; k < devices.size(); k++) {
        MirtContext* c = nullptr;
        if (mirt_create(devices[k], &c) != MIRT_OK)
            throw std::runtime_error(std::string("IntersectionHIP::setDevices: ") + mirt_last_error(nullptr));
        g_all.push_back(c);
    }
}

const std::vector<MirtContext*>& IntersectionHIP::contexts() { return g_all; }

void IntersectionHIP::shutdown() {
    for (size_t k = 1; k < g_all.size(); k++) mirt_destroy(g_all[k]);
    g_all.clear();
    if (g_ctx) mirt_destroy(g_ctx);
    g_ctx = nullptr;
    g_resident = nullptr;
    g_resident_gen = 0;
}

MirtContext* IntersectionHIP::context() { return g_ctx; }

void IntersectionHIP::loadKDTreeToDevice(const Scene& scene) {
    if (!g_ctx) throw std::runtime_error("IntersectionHIP::loadKDTreeToDevice: init() was not called");
    MirtSceneDesc d = scene.describe();
    for (MirtContext* c : g_all)
        if (mirt_upload_scene(c, &d) != MIRT_OK) {
            g_resident = nullptr;  // (a failed upload leaves no scene on the device)
            g_resident_gen = 0;
            throw std::runtime_error(std::string("IntersectionHIP::loadKDTreeToDevice: ") + mirt_last_error(c));
        }
    g_resident = &scene;
    g_resident_gen = scene.generation;
    g_resident_serial = mirt_scene_serial(g_ctx);
}

bool IntersectionHIP::isResident(const Scene& scene) {
    return g_ctx && g_resident == &scene && scene.generation != 0 && g_resident_gen == scene.generation &&
           mirt_scene_serial(g_ctx) == g_resident_serial;
}

void IntersectionHIP::ensureResident(const Scene& scene) {
    if (isResident(scene)) return;
    if (!scene.tree.built()) throw std::runtime_error("IntersectionHIP: the scene has no tree (Scene::buildKDTree was not called)");
    init();
    loadKDTreeToDevice(scene);
}

std::vector<IntersectionHIP::Hit> IntersectionHIP::query(const std::vector<Ray>& rays, int mode) {
    if (!g_ctx) throw std::runtime_error("IntersectionHIP::query: init() was not called");
    const size_t n = rays.size();
    std::vector<Hit> hits(n);
    if (n == 0) return hits;
    // AoS records -> the plane-major SoA the device reads
    std::vector<float> o(3 * n), d(3 * n), t(n);
    std::vector<int32_t> aux(n), tid(n);
    for (size_t i = 0; i < n; i++) {
        for (int c = 0; c < 3; c++) {
            o[c * n + i] = rays[i].origin[c];
            d[c * n + i] = rays[i].direction[c];
        }
        aux[i] = rays[i].last_tid;
    }
    if (mirt_intersect(g_ctx, o.data(), d.data(), aux.data(), n, mode, t.data(), tid.data()) != MIRT_OK)
        throw std::runtime_error(std::string("IntersectionHIP::query: ") + mirt_last_error(g_ctx));
    for (size_t i = 0; i < n; i++) hits[i] = Hit{t[i], tid[i]};
    return hits;
}

std::vector<IntersectionHIP::Hit> IntersectionHIP::query(const std::vector<Ray>& rays) {
    return query(rays, MIRT_MODE_LAST_TID);
}

// ---- Scene ------------------------------------------------------------------------------------------
Scene::~Scene() { clear(); }

void Scene::clear() {
    generation = 0;
    if (g_resident == this) g_resident = nullptr;  // (the device copy stays until the next upload; nothing may query it)
    for (SceneObject* o : objects) delete o;
    objects.clear();
    tree.destroy();
    for (KDTree& t : objKDTrees) t.destroy();
    objKDTrees.clear();
    objTreeDescs.clear();
    objTriangleOffsetMap.clear();
    triPos.clear();
    triObj.clear();
    triLocal.clear();
    triVn.clear();
    objectDescs.clear();
    triVt.clear();
    textures.clear();
    textureDescs.clear();
    objectTex.clear();
}

// Scene::buildKDTree (Scene.cpp:248-285): concatenate every object's world-space vertices and
// index-shifted triangles, remember (objID, triangleID) per global triangle, build the tree; then
// hand the flattened tree to the device (the reference does the same for its GL path, :276-284).
void Scene::touch() { generation = ++g_next_generation; }

void Scene::buildKDTree(bool upload) {
    touch();
    tree.destroy();
    objTriangleOffsetMap.clear();
    triPos.clear();
    triObj.clear();
    triLocal.clear();
    triVn.clear();
    objectDescs.clear();
    triVt.clear();
    textures.clear();
    textureDescs.clear();
    objectTex.clear();
    // corner texture coordinates are flattened only when some object has them for all its faces
    // (SimpleShape::getTexCoord, SimpleShape.cpp:274-298, indexes faceVertexTexCoordIndexList by face)
    bool any_vt = false;
    for (const SceneObject* obj : objects)
        any_vt |= !obj->vertexTexCoordList.empty() && obj->faceVertexTexCoordIndexList.size() == obj->faceVertexIndexList.size();
    unsigned vi_offset = 0, tri_offset = 0;
    for (unsigned oi = 0; oi < objects.size(); oi++) {
        SceneObject* obj = objects[oi];
        obj->objectIndex = (int)oi;
        for (unsigned i = 0; i < obj->getVertexNum(); i++)
            tree.vertexPositionList.push_back(obj->getWorldVertexPosition((int)i));
        const bool smooth = obj->hasVertexNormals();
        const bool has_vt = !obj->vertexTexCoordList.empty() && obj->faceVertexTexCoordIndexList.size() == obj->faceVertexIndexList.size();
        for (unsigned i = 0; i < obj->getTriangleNum(); i++) {
            if (any_vt)
                for (int k = 0; k < 3; k++) {
                    vec3f vt;  // (an object without texture coordinates reads (0,0,0), SimpleShape.cpp:276-277)
                    if (has_vt) {
                        const unsigned ti = obj->faceVertexTexCoordIndexList[i][k];
                        if (ti >= obj->vertexTexCoordList.size()) throw std::runtime_error("Scene: texture coordinate index out of range in " + obj->name);
                        vt = obj->vertexTexCoordList[ti];
                    }
                    triVt.push_back(vt.x);
                    triVt.push_back(vt.y);
                    triVt.push_back(vt.z);
                }
            KDTree::Triangle tri;
            vec3ui v = obj->getVertexIndices((int)i);
            tri.vertexIndices = vec3ui{v.x + vi_offset, v.y + vi_offset, v.z + vi_offset};
            tri.sourceInformation.objID = oi;
            tri.sourceInformation.triangleID = i;
            tree.triangleList.push_back(tri);
            for (int k = 0; k < 3; k++) {
                const vec3f& p = tree.vertexPositionList[tri.vertexIndices[k]];
                triPos.push_back(p.x);
                triPos.push_back(p.y);
                triPos.push_back(p.z);
            }
            triObj.push_back((int32_t)oi);
            triLocal.push_back((int32_t)i);
            vec3f vn[3];
            if (smooth) obj->getWorldCornerNormals(i, vn);
            for (int k = 0; k < 3; k++) {
                triVn.push_back(vn[k].x);
                triVn.push_back(vn[k].y);
                triVn.push_back(vn[k].z);
            }
        }
        MirtObject m = obj->material;
        m.has_vn = smooth ? 1 : 0;
        m.tri_begin = (int32_t)tri_offset;
        m.tri_count = (int32_t)obj->getTriangleNum();
        obj->material = m;
        objectDescs.push_back(m);
        MirtObjectTex ot;
        ot.diffuse = addTexture(obj->diffuseTex);
        ot.glossy_color = addTexture(obj->glossyColorTex);
        ot.glossy_coeff = addTexture(obj->glossyCoeffTex);
        // SceneObject.cpp:253: the bump branch needs a texture of more than one texel and texture coordinates
        ot.bump = obj->bumpTex && obj->bumpTex->size() > 1 && has_vt ? addTexture(obj->bumpTex) : -1;
        objectTex.push_back(ot);
        objTriangleOffsetMap.push_back(tri_offset);
        tri_offset += obj->getTriangleNum();
        vi_offset += obj->getVertexNum();
    }
    textureDescs.clear();
    for (const auto& t : textures) textureDescs.push_back(t->describe());
    tree.build();
    buildObjKDTrees();
    if (useGPU && upload) {
        IntersectionHIP::init();
        IntersectionHIP::loadKDTreeToDevice(*this);
    }
}

// Scene::buildObjKDTrees (Scene.cpp:287-320), restricted to the objects findInsideObject looks at.
// Works from the concatenated lists (an object's triangles are contiguous there, with the same
// vertex sharing as in the object's own lists), so scenes made from arrays get them too.
void Scene::buildObjKDTrees() {
    for (KDTree& t : objKDTrees) t.destroy();
    objKDTrees.assign(objectDescs.size(), KDTree());
    objTreeDescs.clear();
    for (unsigned oi = 0; oi < objectDescs.size(); oi++) {
        const MirtObject& m = objectDescs[oi];
        if (m.type != MIRT_OBJ_VPM) continue;
        KDTree& t = objKDTrees[oi];
        std::vector<int> local(tree.vertexPositionList.size(), -1);
        for (int i = 0; i < m.tri_count; i++) {
            const KDTree::Triangle& g = tree.triangleList[(size_t)m.tri_begin + i];
            KDTree::Triangle tri;
            for (int k = 0; k < 3; k++) {
                unsigned vi = g.vertexIndices[k];
                if (local[vi] < 0) {
                    local[vi] = (int)t.vertexPositionList.size();
                    t.vertexPositionList.push_back(tree.vertexPositionList[vi]);
                }
                tri.vertexIndices[k] = (unsigned)local[vi];
            }
            tri.sourceInformation.objID = oi;
            tri.sourceInformation.triangleID = (unsigned)i;
            t.triangleList.push_back(tri);
        }
        t.build();
    }
    for (unsigned oi = 0; oi < objKDTrees.size(); oi++) {
        const KDTree& t = objKDTrees[oi];
        if (!t.built()) continue;
        MirtObjTree d;
        d.object = (int32_t)oi;
        d.n_nodes = t.numNodes();
        d.node_min = t.nodeMin.data();
        d.node_max = t.nodeMax.data();
        d.node_right = t.nodeRight.data();
        d.node_leaf_begin = t.nodeLeafBegin.data();
        d.node_leaf_end = t.nodeLeafEnd.data();
        d.slot_tid = t.slotTid.data();
        objTreeDescs.push_back(d);
    }
}

int Scene::addTexture(const std::shared_ptr<Texture>& t) {
    if (!t) return -1;
    for (size_t k = 0; k < textures.size(); k++)
        if (textures[k] == t) return (int)k;
    textures.push_back(t);
    return (int)textures.size() - 1;
}

MirtSceneDesc Scene::describe() const {
    MirtSceneDesc d;
    memset(&d, 0, sizeof d);
    d.struct_size = sizeof d;
    d.n_nodes = tree.numNodes();
    d.node_min = tree.nodeMin.data();
    d.node_max = tree.nodeMax.data();
    d.node_right = tree.nodeRight.data();
    d.node_leaf_begin = tree.nodeLeafBegin.data();
    d.node_leaf_end = tree.nodeLeafEnd.data();
    d.n_tris = (int32_t)tree.triangleList.size();
    d.slot_tid = tree.slotTid.data();
    d.tri_pos = triPos.data();
    d.tri_obj = triObj.data();
    d.tri_local = triLocal.data();
    d.tri_vn = triVn.data();
    d.n_objects = (int32_t)objectDescs.size();
    d.objects = objectDescs.data();
    d.n_obj_trees = (int32_t)objTreeDescs.size();
    d.obj_trees = objTreeDescs.data();
    d.tri_vt = triVt.empty() ? nullptr : triVt.data();
    d.n_textures = (int32_t)textureDescs.size();
    d.textures = textureDescs.empty() ? nullptr : textureDescs.data();
    d.object_tex = textureDescs.empty() || objectTex.size() != objectDescs.size() ? nullptr : objectTex.data();
    return d;
}

// Scene::intersect (Scene.cpp:341-352) without a condition, as a one-ray batch on the device.
float Scene::intersect(const Ray& ray, ObjSourceInformation& objSource) {
    std::vector<IntersectionHIP::Ray> one(1);
    one[0].origin = ray.origin;
    one[0].direction = ray.direction;
    one[0].last_tid = -1;
    IntersectionHIP::ensureResident(*this);
    IntersectionHIP::Hit h = IntersectionHIP::query(one, MIRT_MODE_UNCONDITIONAL)[0];
    if (h.dist >= 0) {
        if (h.tid < 0 || (size_t)h.tid >= tree.triangleList.size()) throw std::runtime_error("Scene::intersect: triangle id out of range");
        objSource = tree.triangleList[h.tid].sourceInformation;
    }
    return h.dist;
}

int Scene::getContactTreeTid(const Ray& ray) const {
    if (ray.contactObject)
        for (size_t i = 0; i < objects.size(); i++)
            if (ray.contactObject == objects[i]) return (int)(objTriangleOffsetMap[i] + ray.contactObjectTriangleID);
    return -1;
}

// Scene::fillIntersectObject (Scene.cpp:431-453). The reference's GL path can only skip the
// triangle the ray starts on (last_tid); here rays that carry contact/inside objects use the CPU
// branch's NoSelfIntersectionCondition, which is what image parity is defined against.
void Scene::fillIntersectObject(std::vector<Ray>& rays) {
    std::vector<IntersectionHIP::Ray> q(rays.size());
    for (size_t i = 0; i < rays.size(); i++) {
        q[i].origin = rays[i].origin;
        q[i].direction = rays[i].direction;
        int contact = rays[i].contactObject ? rays[i].contactObject->objectIndex : -1;
        int inside = rays[i].insideObject ? rays[i].insideObject->objectIndex : -1;
        q[i].last_tid = MIRT_PACK_IDS(contact, inside);
    }
    IntersectionHIP::ensureResident(*this);
    std::vector<IntersectionHIP::Hit> hits = IntersectionHIP::query(q, MIRT_MODE_NO_SELF);
    for (size_t i = 0; i < rays.size(); i++) {
        if (hits[i].dist >= 0) {
            if (hits[i].tid < 0 || (size_t)hits[i].tid >= tree.triangleList.size())
                throw std::runtime_error("Scene::fillIntersectObject: triangle id out of range");
            const ObjSourceInformation& osi = tree.triangleList[hits[i].tid].sourceInformation;
            rays[i].intersectObject = objects[osi.objID];
            rays[i].intersectDist = hits[i].dist;
            rays[i].intersectObjectTriangleID = osi.triangleID;
            rays[i].intersect_tid = hits[i].tid;
        }
    }
}

// Scene::testVisibility (Scene.cpp:455-508). The reference's GL route skips only the start triangle
// (last_tid) and then re-shoots rays whose first hit was an illegal self / inside hit; that retest
// is its approximation of NoSelfIntersectionCondition. Here the query runs the predicate itself
// (as fillIntersectObject does), i.e. the rule of the CPU route MCRenderer::testVisibility
// (MCRenderer.cpp:115-117), as one any-hit batch: no second pass is needed.
std::vector<bool> Scene::testVisibility(const std::vector<Ray>& rays) {
    const size_t n = rays.size();
    std::vector<bool> visibilityList(n);
    if (n == 0) return visibilityList;
    IntersectionHIP::ensureResident(*this);
    MirtContext* ctx = IntersectionHIP::context();
    std::vector<float> o(3 * n), d(3 * n), lim(n);
    std::vector<int32_t> aux(n);
    std::vector<uint8_t> vis(n);
    for (size_t i = 0; i < n; i++) {
        for (int c = 0; c < 3; c++) {
            o[c * n + i] = rays[i].origin[c];
            d[c * n + i] = rays[i].direction[c];
        }
        int contact = rays[i].contactObject ? rays[i].contactObject->objectIndex : -1;
        int inside = rays[i].insideObject ? rays[i].insideObject->objectIndex : -1;
        aux[i] = MIRT_PACK_IDS(contact, inside);
        lim[i] = rays[i].intersectDist;
    }
    if (mirt_visibility(ctx, o.data(), d.data(), aux.data(), lim.data(), n, MIRT_MODE_NO_SELF, vis.data()) != MIRT_OK)
        throw std::runtime_error(std::string("Scene::testVisibility: ") + mirt_last_error(ctx));
    for (size_t i = 0; i < n; i++) visibilityList[i] = vis[i] != 0;
    return visibilityList;
}

}  // namespace mirt_host
