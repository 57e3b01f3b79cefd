// host.h — the host surface of the renderer, kept in the reference's own shape (Scene / Ray /
// Camera / SceneObject / KDTree / MCRenderer / ConfigManager / Renderer, same method names and
// argument meaning) so that a user of the reference finds the calls they know. Everything that
// was a CPU or OpenGL hot loop behind these calls now goes through the C-ABI of include/mirt.h:
//   IntersectionGPU::init / loadKDTreeToShader / query  (IntersectionGPU.h:40-45)
//        -> IntersectionHIP::init / loadKDTreeToDevice / query
//   PathTracer::renderPixels                              (PathTracer.cpp:4-58)
//        -> PathTracerHIP::renderPixels
// There is no CPU intersection or CPU path tracer in this library: without a GPU these calls fail.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mirt.h"
#include "vecmath.h"
#include "xml_lite.h"

namespace mirt_host {

class Scene;
class SceneObject;

// Ray.h:8-61 — the fields the batched GPU path reads or fills
class Ray {
public:
    enum SampleType { RANDOM, DEFINITE };  // Ray.h:11
    SampleType directionSampleType = RANDOM, originSampleType = DEFINITE;  // Ray.h:44-46 defaults
    vec3f origin;
    vec3f direction;  // terminated if length < 0.5 (Ray.h:16)
    vec3f color;
    float directionProb, originProb;
    SceneObject* insideObject;
    SceneObject* contactObject;
    unsigned contactObjectTriangleID;
    int current_tid, intersect_tid;
    int pixelID;
    SceneObject* intersectObject;  // filled outside scatter()
    unsigned intersectObjectTriangleID;
    float intersectDist;
    Ray()
        : color(1, 1, 1), directionProb(1), originProb(1), insideObject(nullptr), contactObject(nullptr),
          contactObjectTriangleID(0), current_tid(-1), intersect_tid(-1), pixelID(-1),
          intersectObject(nullptr), intersectObjectTriangleID(0), intersectDist(-1) {}
};

// SimpleShape.h:18-61 — mesh + transform; .obj loading with the reference loader's rules
class SimpleShape {
public:
    std::string name;
    vec3f minCoord, maxCoord;
    std::vector<vec3f> vertexList, vertexNormalList, vertexTexCoordList;
    std::vector<vec3ui> faceVertexIndexList, faceVertexNormalIndexList, faceVertexTexCoordIndexList;
    matrix4 transform;
    virtual ~SimpleShape() {}
    void setTransform(const matrix4& mat) { transform = mat; }
    matrix4 getTransform() const { return transform; }
    vec3f getWorldVertexPosition(int vi) const { return transform.apply(vertexList[vi], 1.f); }
    vec3ui getVertexIndices(int ti) const { return faceVertexIndexList[ti]; }
    unsigned getVertexNum() const { return (unsigned)vertexList.size(); }
    unsigned getTriangleNum() const { return (unsigned)faceVertexIndexList.size(); }
    void getBoundingBox(vec3f& lo, vec3f& hi) const;
    // throws std::runtime_error when the file cannot be read or is not a triangle .obj
    // splitShapes != nullptr: <GroupObject> mode (SimpleShape.cpp:85-96): every "g name" line starts a
    // new sub-shape that receives the vertices / normals / faces that follow it (indices rebased to the
    // counts at the "g" line); the caller owns the returned shapes
    void loadShape(const std::string& fileName, bool normalize = true, std::vector<SimpleShape*>* splitShapes = nullptr);
    matrix4 unitize();           // returns SimpleShape.cpp:270's matrix (see geometry.cpp)
    void applyTransform();       // SimpleShape.cpp:222-236
    // world-space corner normals of face fi, transformed by transpose(inverse(transform)) and NOT
    // normalised (what SimpleShape::getWorldNormal interpolates, SimpleShape.cpp:310,317-323);
    // false when the shape has no usable vertex normals (flat shading)
    bool getWorldCornerNormals(unsigned fi, vec3f out[3]) const;
    bool hasVertexNormals() const;
};

class ConfigManager;

// Texture.h:14-42 — an image (.pfm only, as in the reference: Texture.cpp:5-41) with a coordinate and a
// colour transform. The lookups run on the device (mirt_dev.h tex_color / tex_grad); this class loads, holds and
// describes the data.
class Texture {
public:
    bool hasColor = true;
    int width = 1, height = 1;
    std::vector<float> values;   // gray texels
    std::vector<vec3f> colors;   // rgb texels
    matrix4 coordTransform, colorTransform;
    void setCoordTransform(const matrix4& t) { coordTransform = t; }
    void setColorTransform(const matrix4& t) { colorTransform = t; }
    void loadFile(const std::string& fileName);  // throws std::runtime_error (the reference crashes on a missing file)
    void setColor(const vec3f& color) {
        hasColor = true;
        colors.assign(1, color);
        values.clear();
        width = height = 1;
    }
    void toGrayScale();                                  // Texture.cpp:164-176
    int size() const { return width * height; }
    void loadTextureFromXML(const ConfigManager* cm, XmlNode* nodeTex);  // Texture.cpp:149-162
    MirtTexture describe() const;                        // pointers stay valid while the texture lives
};

// SceneObject.h:19-89 + the Scene*Object wrappers: one object = mesh + one material description
class SceneObject : public SimpleShape {
public:
    Scene* scene;
    int objectIndex;
    MirtObject material;  // type + parameters, what the device shades with
    // image textures of the object's materials (null: the constant in `material`): DiffuseMaterial::tex,
    // GlossyMaterial::colorTex / coeffTex (DiffuseMaterial.h:14, GlossyMaterial.h:14-15)
    std::shared_ptr<Texture> diffuseTex, glossyColorTex, glossyCoeffTex;
    std::shared_ptr<Texture> bumpTex;  // SceneObject::bumpTex (SceneObject.h:34), gray (ConfigManager.cpp:161)
    explicit SceneObject(Scene* scene);
    bool emissive() const { return material.type == MIRT_OBJ_EMISSIVE; }
    bool isVolumetric() const { return material.volumetric != 0; }
    float getRefrCoeff() const;
};

// KDTree.h:10-72 — same build rules (median split, BEST strategy, KDTree.cpp:129-246), built
// directly into the pre-order arrays that KDTree::serializeForGPU (KDTree.cpp:346-383) produces.
class KDTree {
public:
    struct Ray {
        vec3f origin, direction;
    };
    struct ObjSourceInformation {
        unsigned objID, triangleID;
    };
    struct Triangle {
        vec3ui vertexIndices;
        ObjSourceInformation sourceInformation;
    };
    std::vector<vec3f> vertexPositionList;
    std::vector<Triangle> triangleList;
    // flattened, pre-order; left child of inner node i is i+1
    std::vector<float> nodeMin, nodeMax;  // 3 per node
    std::vector<int32_t> nodeRight, nodeLeafBegin, nodeLeafEnd, slotTid;
    int depth = 0;
    // KDTree.h:24: LOOP cycles the split axis x, y, z, x, ... with the depth (splitLoop, KDTree.cpp:54-127); BEST (the
    // default, KDTree.h:54) takes per node the axis whose median split is most balanced (splitBest, KDTree.cpp:129-228)
    enum Strategy { LOOP, BEST };
    void build(Strategy strategy = BEST);
    void destroy();
    bool built() const { return !nodeRight.empty(); }
    vec3f getDiagonal() const;
    int numNodes() const { return (int)nodeRight.size(); }

private:
    void buildNode(std::vector<unsigned>& order, size_t begin, size_t end, int level,
                   std::vector<unsigned>& stamp, unsigned& stampGen, std::vector<unsigned>& uniq,
                   std::vector<float>& coords, int loopDim = -1);
};

// Camera.h:12-30
class Camera {
public:
    vec3f focus, position, up;
    unsigned sightDist = 0;
    int width = 0, height = 0;
    MirtCamera toMirt() const;
};

// IntersectionGPU.h:12-46 — the batch closest-hit service, now HIP
class IntersectionHIP {
public:
    struct Ray {
        vec3f origin, direction;
        int last_tid;
    };
    struct Hit {  // vec4f(dist, tid, 0, 0) in the reference (Intersection.frag:83)
        float dist;
        int tid;
    };
    // throws std::runtime_error on failure (the reference exit(1)s, Shader.cpp:9-12)
    static void init(int device = 0);
    static void shutdown();
    static MirtContext* context();
    // Several GPUs (the reference has one GL context, IntersectionGPU.cpp:16-44): one more context per listed device
    // BESIDE the first, which stays the one batch queries go to. devices[0] must be the device init() was (or will be)
    // called with; a device may be listed twice. loadKDTreeToDevice uploads to all of them, and PathTracerHIP then
    // renders through mirt_render_pt_multi (sample shards + one reduce). An empty list or one entry = single GPU.
    static void setDevices(const std::vector<int>& devices);
    static const std::vector<MirtContext*>& contexts();  // [0] == context()
    static void loadKDTreeToDevice(const Scene& scene);
    // The service holds ONE scene at a time (like the reference's static GL state, IntersectionGPU.cpp:5-14).
    // loadKDTreeToDevice records which (the Scene and the generation of its tree); every query made through a
    // Scene first calls ensureResident, which uploads that scene again when another one (or an older tree of
    // the same one) is on the device, so a query can never be answered from somebody else's geometry.
    static bool isResident(const Scene& scene);
    static void ensureResident(const Scene& scene);
    // mode MIRT_MODE_LAST_TID with ray.last_tid, like IntersectionGPU::query
    static std::vector<Hit> query(const std::vector<Ray>& rays);
    // general form: aux per ray, any MIRT_MODE_*
    static std::vector<Hit> query(const std::vector<Ray>& rays, int mode);
};

// Scene.h:10-107 — the parts on the path
class Scene {
public:
    typedef KDTree::ObjSourceInformation ObjSourceInformation;
    KDTree tree;
    std::vector<unsigned> objTriangleOffsetMap;
    std::vector<SceneObject*> objects;
    bool useGPU = true;  // the HIP path is the only one this library has
    unsigned long long generation = 0;  // unique stamp of the current tree + textures (0: none built), see IntersectionHIP::isResident
    void touch();                       // new stamp: whatever is on the device no longer describes this scene
    // flattened per-triangle data (by global triangle id), filled by buildKDTree
    std::vector<float> triPos;       // 9 per triangle
    std::vector<int32_t> triObj, triLocal;
    std::vector<float> triVn;        // 9 per triangle
    std::vector<MirtObject> objectDescs;
    // textures: corner texture coordinates (9 per triangle; empty when no object has any), the textures the
    // objects' materials read and which object reads which
    std::vector<float> triVt;
    std::vector<std::shared_ptr<Texture>> textures;
    std::vector<MirtTexture> textureDescs;
    std::vector<MirtObjectTex> objectTex;
    int addTexture(const std::shared_ptr<Texture>& t);  // index in `textures` (shared textures are stored once)
    // Scene::objKDTrees (Scene.h / Scene.cpp:287-320): built for the volumetric objects, the only
    // ones findInsideObject ever queries (Scene.cpp:369); empty trees elsewhere
    std::vector<KDTree> objKDTrees;
    std::vector<MirtObjTree> objTreeDescs;
    void buildObjKDTrees();
    ~Scene();
    void setGPU(bool on) { useGPU = on; }
    bool usingGPU() const { return useGPU; }
    void clear();
    void buildKDTree(bool upload = true);
    MirtSceneDesc describe() const;  // pointers stay valid until the scene changes
    float intersect(const Ray& ray, ObjSourceInformation& objSource);  // unconditional, 1-ray batch
    void fillIntersectObject(std::vector<Ray>& rays);                  // Scene.cpp:431-453
    // Scene::testVisibility (Scene.cpp:455-508): rays[i].intersectDist holds the distance to the
    // vertex being connected; true = nothing legal nearer than it (MCRenderer.cpp:115-117 rule)
    std::vector<bool> testVisibility(const std::vector<Ray>& rays);
    int getContactTreeTid(const Ray& ray) const;                       // Scene.cpp:420-429
    vec3f getDiagonal() const { return tree.getDiagonal(); }
};

class Renderer;

// MCRenderer.h:12-119
class MCRenderer {
protected:
    Renderer* renderer;
    std::string savePath;
    unsigned maxDepth;
    int outputIter;

public:
    explicit MCRenderer(Renderer* r) : renderer(r), maxDepth(20), outputIter(5) {}
    virtual ~MCRenderer() {}
    void setMaxDepth(int d) { maxDepth = (unsigned)d; }
    int getMaxDepth() const { return (int)maxDepth; }
    void setSavePath(const std::string& p) { savePath = p; }
    const std::string& getSavePath() const { return savePath; }
    virtual std::vector<vec3f> renderPixels(const Camera& camera);
    // MCRenderer::testVisibility (MCRenderer.cpp:106-118), a one-ray batch on the device
    bool testVisibility(const Ray& startRay, const Ray& endRay);
    // MCRenderer::testPathListVisibility (MCRenderer.cpp:520-570): for every pair of paths, one bit per
    // (start vertex j, end vertex k) at index j * |end| + k, set when both vertices have RANDOM directions and
    // the segment between them is unoccluded; the connections go to the device in batches of > 2^20 rays
    typedef std::vector<Ray> Path;
    std::vector<std::vector<unsigned>> testPathListVisibility(const std::vector<Path>& startPathList,
                                                              const std::vector<Path>& endPathList);
    // PFM writer with the reference's layout (MCRenderer.cpp:500-518)
    void setOutputIter(int n) { outputIter = n; }
    int getOutputIter() const { return outputIter; }
    // MCRenderer::showCurrentResult (MCRenderer.cpp:396-437) without the OpenCV window: writes the
    // running mean to savePath; with time and iter also "<savePath up to its first '.'>_iter_<iter>_time_<time>.pfm",
    // the series compare.cpp:167-251 turns into an iteration / time / RMSE table
    void showCurrentResult(const std::vector<vec3f>& pixelColors, int w, int h, const unsigned* time = nullptr,
                           const unsigned* iter = nullptr) const;
    static void saveImagePFM(const std::string& fileName, const std::vector<vec3f>& pixels, int w, int h);
    static bool loadImagePFM(const std::string& fileName, std::vector<vec3f>& pixels, int& w, int& h);
};

// PathTracer.h / PathTracer.cpp:4-58 with the sample loop on the GPU. spp is configurable here
// (the reference's is hard-wired to "forever", PathTracer.h:7,14).
class PathTracerHIP : public MCRenderer {
public:
    unsigned spp = 64;
    unsigned seed = 1;
    unsigned sppPerLaunch = 0;
    // Progressive output (PathTracer.cpp:47-55): when set, the image is dumped after every pass s with
    // s % outputIter == 0, as the reference does, i.e. the passes run in slices 1, outputIter, outputIter, ...
    // Slicing does not change the image: sample streams are keyed by the global sample index.
    bool progressive = false;
    // Resumable accumulators (the reference has none): radiance SUMS + the number of passes they hold.
    // renderPixels continues after sppDone; saveState / loadState move them through a file.
    // A render starts from zero unless `resume` is set (loadState sets it; it is cleared by the render).
    std::vector<float> sums;
    unsigned sppDone = 0;
    bool resume = false;
    void saveState(const std::string& fileName, const Camera& camera) const;
    void loadState(const std::string& fileName, const Camera& camera);  // throws on a mismatching image / seed / depth
    void resetState() { sums.clear(); sppDone = 0; }
    MirtRenderStats lastStats{};
    explicit PathTracerHIP(Renderer* r) : MCRenderer(r) {}
    std::vector<vec3f> renderPixels(const Camera& camera) override;

protected:
    // the device entry point this integrator runs on (mirt_render_pt)
    virtual int launch(MirtContext* ctx, const MirtRenderParams* p, float* rgb_sum, MirtRenderStats* stats);
};

// PathTracerTest.h / PathTracerTest.cpp:7-135 ("PTtest", ConfigManager.cpp:285-288): the reference's
// cross-check integrator, path tracing + one light connection per RANDOM vertex with MIS
// (useConnection is always on there, PathTracerTest.h:16). Runs on mirt_render_pttest.
class PathTracerTestHIP : public PathTracerHIP {
public:
    explicit PathTracerTestHIP(Renderer* r) : PathTracerHIP(r) {}

protected:
    int launch(MirtContext* ctx, const MirtRenderParams* p, float* rgb_sum, MirtRenderStats* stats) override;
};

// ConfigManager.h:20-70
class ConfigManager {
public:
    explicit ConfigManager(Renderer* r) : renderer(r) {}
    void load(const std::string& configFilePath);
    std::string getRootPath() const { return rootFolder; }
    std::string getFullPath(const std::string& path) const { return resolve(path); }  // ConfigManager.h getFullPath

private:
    Renderer* renderer;
    std::string rootFolder, currentPath;
    std::map<std::string, std::unique_ptr<XmlDocument>> path_doc;
    XmlNode* findNode(const std::string& filePath, const std::string& nodeTag, const std::string& nodeName);
    XmlNode* findNode(XmlNode* root, const std::string& nodeTag, const std::string& nodeName);
    std::pair<std::string, std::string> getPathAndName(XmlNode* node);
    SceneObject* generateSceneObject(XmlNode* nodeObj, XmlNode* nodeMat, const SimpleShape* shape = nullptr);
    std::string resolve(const std::string& path) const;
};

// Renderer.h:10-32
class Renderer {
public:
    Scene scene;
    Camera camera;
    MCRenderer* mcRenderer = nullptr;
    ConfigManager* configManager;
    bool uploadOnLoad = true;  // tests that only inspect the host data switch this off
    Renderer();
    ~Renderer();
    void loadConfig(const std::string& configFilePath);
    std::vector<vec3f> render();
    Camera& getCamera() { return camera; }
};

// xmlHelper.cpp:4-23
matrix4 readMatrix(const char* value);
vec3f readVec(const char* value);

}  // namespace mirt_host
