// config.cpp — xmlHelper, ConfigManager (XML -> Scene/Camera/integrator), MCRenderer, PathTracerHIP,
// Renderer. Follows ConfigManager.cpp:25-433, xmlHelper.cpp:4-23, MCRenderer.cpp:500-518,
// Renderer.cpp:46-91 for the parts on the path-tracing path; every other integrator type is refused
// with a clear error instead of being silently ignored.
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "host.h"

namespace mirt_host {

// xmlHelper.cpp:4-16: "[r0][r1][r2][r3]", row-major text
matrix4 readMatrix(const char* value) {
    float v[16];
    int got = sscanf(value, "[%f, %f, %f, %f][%f, %f, %f, %f][%f, %f, %f, %f][%f, %f, %f, %f]", v, v + 1,
                     v + 2, v + 3, v + 4, v + 5, v + 6, v + 7, v + 8, v + 9, v + 10, v + 11, v + 12, v + 13,
                     v + 14, v + 15);
    if (got != 16) throw std::runtime_error(std::string("readMatrix: cannot parse '") + value + "'");
    matrix4 m;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) m(r, c) = v[4 * r + c];
    return m;
}

// xmlHelper.cpp:18-23: "(x, y, z)"
vec3f readVec(const char* value) {
    vec3f v;
    if (sscanf(value, " (%f, %f, %f)", &v.x, &v.y, &v.z) != 3)
        throw std::runtime_error(std::string("readVec: cannot parse '") + value + "'");
    return v;
}

static std::string folderOf(const std::string& path) {
    size_t p = path.find_last_of("/\\");
    return p == std::string::npos ? std::string(".") : path.substr(0, p);
}

static bool fileExists(const std::string& p) { return access(p.c_str(), F_OK) == 0; }

std::string ConfigManager::resolve(const std::string& path) const {
    if (fileExists(path)) return path;  // ConfigManager.cpp:196-199, 238-241, 391-392
    return rootFolder + "/" + path;
}

XmlNode* ConfigManager::findNode(const std::string& filePath, const std::string& nodeTag,
                                 const std::string& nodeName) {
    if (filePath.empty()) return nullptr;
    std::string full = resolve(filePath);
    auto it = path_doc.find(full);
    if (it == path_doc.end()) {
        std::ifstream in(full.c_str(), std::ios::binary);
        if (!in) throw std::runtime_error("ConfigManager: cannot open " + full);
        std::stringstream ss;
        ss << in.rdbuf();
        std::unique_ptr<XmlDocument> doc(new XmlDocument);
        doc->parse(ss.str());
        it = path_doc.emplace(full, std::move(doc)).first;
    }
    return findNode(it->second.get(), nodeTag, nodeName);
}

// ConfigManager.cpp:415-425: first node of that tag with no Name, or any node when no name is asked
XmlNode* ConfigManager::findNode(XmlNode* root, const std::string& nodeTag, const std::string& nodeName) {
    for (XmlNode* n = root->first_node(nodeTag.c_str()); n; n = n->next_sibling(nodeTag.c_str())) {
        const std::string* nm = n->first_attribute("Name");
        if (!nm || nodeName.empty()) return n;
        if (*nm == nodeName) return n;
    }
    return nullptr;
}

std::pair<std::string, std::string> ConfigManager::getPathAndName(XmlNode* node) {
    std::pair<std::string, std::string> pn;
    pn.first = node->first_node("path") ? node->first_node("path")->value() : currentPath;
    pn.second = node->first_node("name") ? node->first_node("name")->value() : "";
    return pn;
}

static void setColor(float dst[3], const vec3f& v) {
    dst[0] = v.x;
    dst[1] = v.y;
    dst[2] = v.z;
}

// ConfigManager.cpp:25-164 with the defaults of the Scene*Object constructors
SceneObject* ConfigManager::generateSceneObject(XmlNode* nodeObj, XmlNode* nodeMat, const SimpleShape* shape) {
    XmlNode* typeNode = nodeMat->first_node("type");
    if (!typeNode) throw std::runtime_error("ConfigManager: <Material> without <type>");
    const std::string type = typeNode->value();
    std::unique_ptr<SceneObject> obj(new SceneObject(&renderer->scene));
    MirtObject& m = obj->material;
    if (nodeObj->first_node("transform")) obj->setTransform(readMatrix(nodeObj->first_node("transform")->value()));
    // a <GroupObject> member takes its mesh AND transform from the split shape (ConfigManager.cpp:34-35:
    // the assignment comes after setTransform, so the member's own <transform> is overwritten)
    if (shape) *static_cast<SimpleShape*>(obj.get()) = *shape;
    // image textures (DiffuseMaterial.cpp:104-110, GlossyMaterial.cpp:88-98): the <color> / <coeff> constant
    // stays in the material record, a texture node replaces it at shading time
    auto loadTex = [&](XmlNode* parent, const char* tag) -> std::shared_ptr<Texture> {
        XmlNode* n = parent->first_node(tag);
        if (!n) return nullptr;
        std::shared_ptr<Texture> t(new Texture);
        t->loadTextureFromXML(this, n);
        return t;
    };
    // ConfigManager.cpp:158-162: any material may carry a bump texture; it perturbs the object's shading normal
    // (SceneObject.cpp:250-283) and is reduced to one channel
    if (nodeMat->first_node("BumpTexture")) {
        obj->bumpTex = loadTex(nodeMat, "BumpTexture");
        obj->bumpTex->toGrayScale();
    }
    if (type == "Diffuse") {
        m.type = MIRT_OBJ_DIFFUSE;
        if (nodeMat->first_node("color")) setColor(m.color, readVec(nodeMat->first_node("color")->value()));
        obj->diffuseTex = loadTex(nodeMat, "Texture");
    } else if (type == "Glossy") {
        m.type = MIRT_OBJ_GLOSSY;
        if (nodeMat->first_node("color")) setColor(m.color, readVec(nodeMat->first_node("color")->value()));
        if (nodeMat->first_node("coeff")) m.coeff = (float)atof(nodeMat->first_node("coeff")->value());
        obj->glossyColorTex = loadTex(nodeMat, "ColorTexture");
        obj->glossyCoeffTex = loadTex(nodeMat, "CoeffTexture");
    } else if (type == "Phong") {  // ConfigManager.cpp:48-58, ScenePhongObject.h:7-23
        m.type = MIRT_OBJ_PHONG;
        XmlNode* g = nodeMat->first_node("GlossyComponent");
        XmlNode* df = nodeMat->first_node("DiffuseComponent");
        if (!g || !df) throw std::runtime_error("ConfigManager: Phong material needs <GlossyComponent> and <DiffuseComponent>");
        for (int c = 0; c < 3; c++) m.color2[c] = 1.f;  // GlossyMaterial / DiffuseMaterial default to white
        if (g->first_node("color")) setColor(m.color2, readVec(g->first_node("color")->value()));
        if (g->first_node("coeff")) m.coeff = (float)atof(g->first_node("coeff")->value());
        if (df->first_node("color")) setColor(m.color, readVec(df->first_node("color")->value()));
        obj->glossyColorTex = loadTex(g, "ColorTexture");
        obj->glossyCoeffTex = loadTex(g, "CoeffTexture");
        obj->diffuseTex = loadTex(df, "Texture");
    } else if (type == "Emissive") {
        m.type = MIRT_OBJ_EMISSIVE;
        if (!nodeMat->first_node("color")) throw std::runtime_error("ConfigManager: Emissive material needs <color>");
        setColor(m.color, readVec(nodeMat->first_node("color")->value()));
    } else if (type == "Refractive") {  // SceneRefractiveObject.h:12-20
        m.type = MIRT_OBJ_REFRACTIVE;
        m.ior = 1.5f;
        if (nodeMat->first_node("coeff")) m.ior = (float)atof(nodeMat->first_node("coeff")->value());
        if (nodeMat->first_node("surfColor")) setColor(m.color, readVec(nodeMat->first_node("surfColor")->value()));
        if (nodeMat->first_node("decayColor")) setColor(m.decay, readVec(nodeMat->first_node("decayColor")->value()));
    } else if (type == "Reflective") {  // SceneReflectiveObject.h:9-17: refrCoeff -1 = mirror
        m.type = MIRT_OBJ_REFRACTIVE;
        m.ior = -1.f;
        if (nodeMat->first_node("color")) setColor(m.color, readVec(nodeMat->first_node("color")->value()));
    } else if (type == "VPM") {
        m.type = MIRT_OBJ_VPM;
        m.volumetric = 1;
        if (!nodeMat->first_node("sigma_s") || !nodeMat->first_node("sigma_a") ||
            !nodeMat->first_node("inrate") || !nodeMat->first_node("g"))
            throw std::runtime_error("ConfigManager: VPM material needs sigma_s, sigma_a, inrate, g");
        setColor(m.sigma_s, readVec(nodeMat->first_node("sigma_s")->value()));
        setColor(m.sigma_a, readVec(nodeMat->first_node("sigma_a")->value()));
        m.ior = (float)atof(nodeMat->first_node("inrate")->value());
        m.g = (float)atof(nodeMat->first_node("g")->value());
    } else {
        throw std::runtime_error("ConfigManager: material type '" + type + "' is outside this build");
    }
    return obj.release();
}

// ConfigManager::load (ConfigManager.cpp:166-384)
void ConfigManager::load(const std::string& configFilePath) {
    path_doc.clear();
    renderer->scene.clear();
    rootFolder = folderOf(configFilePath);
    XmlNode* nodeConfig = findNode(configFilePath, "Config", "");
    if (!nodeConfig) throw std::runtime_error("ConfigManager: no <Config> in " + configFilePath);
    if (nodeConfig->first_node("useGPU"))  // ConfigManager.cpp:176-181; the HIP path is always on
        renderer->scene.setGPU(true);

    XmlNode* nodeSceneConfig = nodeConfig->first_node("Scene");
    if (!nodeSceneConfig) throw std::runtime_error("ConfigManager: <Config> without <Scene>");
    std::pair<std::string, std::string> pn = getPathAndName(nodeSceneConfig);
    XmlNode* nodeScene = findNode(pn.first, "Scene", pn.second);
    if (!nodeScene) nodeScene = findNode(configFilePath, "Scene", pn.second);
    if (!nodeScene) throw std::runtime_error("ConfigManager: scene '" + pn.second + "' not found");
    currentPath = pn.first;

    // <GroupObject> (ConfigManager.cpp:192-226): one .obj split at its "g" lines into named sub-shapes,
    // each member <Object Name="..."> taking the sub-shape of that name; group members come first
    for (XmlNode* nodeGroup = nodeScene->first_node("GroupObject"); nodeGroup; nodeGroup = nodeGroup->next_sibling("GroupObject")) {
        if (!nodeGroup->first_node("filePath")) throw std::runtime_error("ConfigManager: <GroupObject> without <filePath>");
        const std::string path = resolve(nodeGroup->first_node("filePath")->value());
        SimpleShape ss;
        if (nodeGroup->first_node("transform")) ss.setTransform(readMatrix(nodeGroup->first_node("transform")->value()));
        std::vector<SimpleShape*> shapes;
        ss.loadShape(path, true, &shapes);
        struct Owner {
            std::vector<SimpleShape*>& v;
            ~Owner() { for (SimpleShape* p : v) delete p; }
        } owner{shapes};
        std::map<std::string, SimpleShape*> name_shape;
        for (SimpleShape* p : shapes) name_shape[p->name] = p;
        for (XmlNode* nodeObj = nodeGroup->first_node("Object"); nodeObj; nodeObj = nodeObj->next_sibling("Object")) {
            XmlNode* matRef = nodeObj->first_node("Material");
            if (!matRef) throw std::runtime_error("ConfigManager: <Object> without <Material>");
            pn = getPathAndName(matRef);
            XmlNode* nodeMat = findNode(pn.first, "Material", pn.second);
            if (!nodeMat && matRef->first_node("name")) nodeMat = findNode(nodeScene, "Material", pn.second);
            if (!nodeMat) nodeMat = matRef;
            const std::string* nm = nodeObj->first_attribute("Name");
            if (!nm || !name_shape.count(*nm))  // the reference dereferences a NULL shape here
                throw std::runtime_error("ConfigManager: <GroupObject> member '" + (nm ? *nm : std::string("?")) +
                                         "' has no group of that name in " + path);
            renderer->scene.objects.push_back(generateSceneObject(nodeObj, nodeMat, name_shape[*nm]));
        }
    }

    for (XmlNode* nodeObj = nodeScene->first_node("Object"); nodeObj; nodeObj = nodeObj->next_sibling("Object")) {
        XmlNode* matRef = nodeObj->first_node("Material");
        if (!matRef) throw std::runtime_error("ConfigManager: <Object> without <Material>");
        pn = getPathAndName(matRef);
        // lookup order: named file -> scene-local -> inline (ConfigManager.cpp:230-235)
        XmlNode* nodeMat = findNode(pn.first, "Material", pn.second);
        if (!nodeMat && matRef->first_node("name")) nodeMat = findNode(nodeScene, "Material", pn.second);
        if (!nodeMat) nodeMat = matRef;
        if (!nodeObj->first_node("filePath")) throw std::runtime_error("ConfigManager: <Object> without <filePath>");
        std::string path = resolve(nodeObj->first_node("filePath")->value());
        SceneObject* obj = generateSceneObject(nodeObj, nodeMat);
        try {
            obj->loadShape(path, true);  // ConfigManager.cpp:245
        } catch (...) {
            delete obj;
            throw;
        }
        renderer->scene.objects.push_back(obj);
    }

    XmlNode* camName = nodeSceneConfig->first_node("camera");
    XmlNode* nodeCam = findNode(nodeScene, "Camera", camName ? camName->value() : "");
    if (!nodeCam) throw std::runtime_error("ConfigManager: camera not found");
    Camera& cam = renderer->camera;
    auto need = [&](const char* tag) {
        XmlNode* n = nodeCam->first_node(tag);
        if (!n) throw std::runtime_error(std::string("ConfigManager: <Camera> without <") + tag + ">");
        return n->value();
    };
    cam.width = atoi(need("width"));
    cam.height = atoi(need("height"));
    cam.sightDist = (unsigned)atoi(need("sightDist"));
    cam.focus = readVec(need("focus"));
    cam.position = readVec(need("position"));
    cam.up = readVec(need("up"));

    renderer->scene.buildKDTree(renderer->uploadOnLoad);

    delete renderer->mcRenderer;
    renderer->mcRenderer = nullptr;
    // <renderer>name</renderer> or <Renderer><type>name</type>...</Renderer> (ConfigManager.cpp:278-378)
    std::string typeName;
    XmlNode* opts = nodeConfig;
    if (nodeConfig->first_node("renderer")) typeName = nodeConfig->first_node("renderer")->value();
    if (XmlNode* r = nodeConfig->first_node("Renderer")) {
        if (!r->first_node("type")) throw std::runtime_error("ConfigManager: <Renderer> without <type>");
        typeName = r->first_node("type")->value();
        opts = r;
    }
    const bool isTest = typeName == "PTtest" || typeName == "PathTracerTest";  // ConfigManager.cpp:285-288
    if (typeName == "PathTracer" || typeName == "PT" || typeName == "PTHIP" || isTest) {
        PathTracerHIP* pt = isTest ? new PathTracerTestHIP(renderer) : new PathTracerHIP(renderer);
        if (opts->first_node("spp")) pt->spp = (unsigned)atoi(opts->first_node("spp")->value());
        if (opts->first_node("seed")) pt->seed = (unsigned)atoi(opts->first_node("seed")->value());
        renderer->mcRenderer = pt;
    } else {
        throw std::runtime_error("ConfigManager: renderer type '" + typeName +
                                 "' is outside this build (PathTracer / PT / PTHIP / PTtest only)");
    }
    if (opts->first_node("outputIter")) {  // not in the reference's XML (MCRenderer.h:109 hard-wires 5)
        renderer->mcRenderer->setOutputIter(atoi(opts->first_node("outputIter")->value()));
        if (PathTracerHIP* pt = dynamic_cast<PathTracerHIP*>(renderer->mcRenderer)) pt->progressive = true;
    }
    if (opts->first_node("maxDepth")) renderer->mcRenderer->setMaxDepth(atoi(opts->first_node("maxDepth")->value()));
    if (nodeConfig->first_node("savePath")) renderer->mcRenderer->setSavePath(nodeConfig->first_node("savePath")->value());
}

// ---- MCRenderer -------------------------------------------------------------------------------------
std::vector<vec3f> MCRenderer::renderPixels(const Camera& camera) {
    return std::vector<vec3f>((size_t)camera.width * camera.height, vec3f(0, 0, 0));  // MCRenderer.h:114-117
}

// MCRenderer::testVisibility (MCRenderer.cpp:106-118): shoot startRay with its own
// NoSelfIntersectionCondition; visible when nothing legal lies nearer than the end vertex.
bool MCRenderer::testVisibility(const Ray& startRay, const Ray& endRay) {
    std::vector<Ray> one(1, startRay);
    one[0].intersectDist = (startRay.origin - endRay.origin).length();
    return renderer->scene.testVisibility(one)[0];
}

// MCRenderer::testPathListVisibility (MCRenderer.cpp:520-570)
std::vector<std::vector<unsigned>> MCRenderer::testPathListVisibility(const std::vector<Path>& startPathList,
                                                                      const std::vector<Path>& endPathList) {
    if (startPathList.size() != endPathList.size())
        throw std::runtime_error("testPathListVisibility: the two lists must pair up");
    std::vector<std::vector<unsigned>> bits(startPathList.size());
    auto connectable = [](const Ray& a, const Ray& b) {
        return a.directionSampleType == Ray::RANDOM && b.directionSampleType == Ray::RANDOM;
    };
    std::vector<Ray> batch;
    size_t first = 0;  // first path pair whose connections are in `batch`
    for (size_t i = 0; i < startPathList.size(); i++) {
        const Path& sp = startPathList[i];
        const Path& ep = endPathList[i];
        bits[i].assign((sp.size() * ep.size() + 31) / 32, 0u);
        for (const Ray& a : sp)
            for (const Ray& b : ep) {
                if (!connectable(a, b)) continue;
                Ray v = a;
                v.direction = b.origin - a.origin;
                v.intersectDist = v.direction.length();
                v.direction.normalize();
                batch.push_back(v);
            }
        if (batch.size() <= 1024 * 1024 && i + 1 != startPathList.size()) continue;
        const std::vector<bool> vis = renderer->scene.testVisibility(batch);
        size_t next = 0;
        for (size_t p = first; p <= i; p++) {
            const size_t ne = endPathList[p].size(), total = startPathList[p].size() * ne;
            for (size_t q = 0; q < total; q++)
                if (connectable(startPathList[p][q / ne], endPathList[p][q % ne]) && vis[next++])
                    bits[p][q / 32] |= 1u << (q % 32);
        }
        first = i + 1;
        batch.clear();
    }
    return bits;
}

// MCRenderer.cpp:500-518: "PF\nW H\n-1.000000\n", rows bottom-to-top, channels R,G,B, little-endian
void MCRenderer::saveImagePFM(const std::string& fileName, const std::vector<vec3f>& pixels, int w, int h) {
    FILE* f = fopen(fileName.c_str(), "wb");
    if (!f) throw std::runtime_error("saveImagePFM: cannot open " + fileName);
    fprintf(f, "PF\n%d %d\n-1.000000\n", w, h);
    for (int j = h - 1; j >= 0; j--)
        for (int i = 0; i < w; i++) {
            const vec3f& p = pixels[(size_t)j * w + i];
            float rgb[3] = {p.x, p.y, p.z};
            fwrite(rgb, sizeof(float), 3, f);
        }
    fclose(f);
}

bool MCRenderer::loadImagePFM(const std::string& fileName, std::vector<vec3f>& pixels, int& w, int& h) {
    FILE* f = fopen(fileName.c_str(), "rb");
    if (!f) return false;
    char magic[3] = {0, 0, 0};
    float scale = 0;
    if (fscanf(f, "%2s %d %d %f", magic, &w, &h, &scale) != 4 || strcmp(magic, "PF") != 0 || w <= 0 || h <= 0) {
        fclose(f);
        return false;
    }
    fgetc(f);  // the single whitespace after the scale
    pixels.assign((size_t)w * h, vec3f());
    bool ok = true;
    for (int j = h - 1; j >= 0 && ok; j--)
        for (int i = 0; i < w; i++) {
            float rgb[3];
            if (fread(rgb, sizeof(float), 3, f) != 3) {
                ok = false;
                break;
            }
            pixels[(size_t)j * w + i] = vec3f(rgb[0], rgb[1], rgb[2]);
        }
    fclose(f);
    return ok;
}

// MCRenderer::showCurrentResult (MCRenderer.cpp:396-437)
void MCRenderer::showCurrentResult(const std::vector<vec3f>& pixelColors, int w, int h, const unsigned* time,
                                   const unsigned* iter) const {
    if (!savePath.empty()) saveImagePFM(savePath, pixelColors, w, h);
    if (iter && time) {
        std::string fileName;
        for (char c : savePath) {
            if (c == '.') break;
            fileName.push_back(c);
        }
        fileName += "_iter_" + std::to_string(*iter) + "_time_" + std::to_string(*time) + ".pfm";
        saveImagePFM(fileName, pixelColors, w, h);
    }
}

// ---- PathTracerHIP ----------------------------------------------------------------------------------
// PathTracer::renderPixels (PathTracer.cpp:4-58): spp passes of one sample per pixel, running mean.
// Here the passes run on the device in slices; the sums come back per slice and are divided by the
// number of passes they hold.
std::vector<vec3f> PathTracerHIP::renderPixels(const Camera& camera) {
    if (!renderer) throw std::runtime_error("PathTracerHIP::renderPixels: no Renderer");
    IntersectionHIP::ensureResident(renderer->scene);  // (another scene may have been uploaded since this one was)
    MirtContext* ctx = IntersectionHIP::context();
    const size_t npix = (size_t)camera.width * camera.height;
    if (!resume || sums.size() != 3 * npix) {
        sums.assign(3 * npix, 0.f);
        sppDone = 0;
    }
    resume = false;
    lastStats = MirtRenderStats{};
    std::vector<vec3f> pixelColors(npix);
    auto mean = [&]() {
        const float inv = sppDone ? 1.f / float(sppDone) : 0.f;
        for (size_t i = 0; i < npix; i++)
            pixelColors[i] = vec3f(sums[3 * i] * inv, sums[3 * i + 1] * inv, sums[3 * i + 2] * inv);
    };
    const auto t_start = std::chrono::steady_clock::now();
    std::vector<float> slice(3 * npix);
    while (sppDone < spp) {
        // progressive: stop after the next pass s with s % outputIter == 0 (PathTracer.cpp:47), s = sppDone + n - 1
        unsigned n = spp - sppDone;
        if (progressive && outputIter > 0) {
            const unsigned k = (unsigned)outputIter;
            const unsigned next = ((sppDone + k - 1) / k) * k;  // first s >= sppDone with s % k == 0
            n = std::min(n, next - sppDone + 1);
        }
        MirtRenderParams p;
        memset(&p, 0, sizeof p);
        p.camera = camera.toMirt();
        p.spp_begin = sppDone;
        p.spp_count = n;
        p.max_depth = maxDepth;
        p.seed = seed;
        p.spp_per_launch = sppPerLaunch;
        MirtRenderStats st{};
        if (launch(ctx, &p, slice.data(), &st) != MIRT_OK)
            throw std::runtime_error(std::string("PathTracerHIP::renderPixels: ") + mirt_last_error(ctx));
        for (size_t i = 0; i < 3 * npix; i++) sums[i] += slice[i];
        sppDone += n;
        lastStats.n_samples += st.n_samples;
        lastStats.n_rays += st.n_rays;
        lastStats.n_shadow_rays += st.n_shadow_rays;
        lastStats.gpu_ms += st.gpu_ms;
        lastStats.intersect_ms += st.intersect_ms;
        if (progressive && outputIter > 0 && (sppDone - 1) % (unsigned)outputIter == 0) {
            mean();
            unsigned s = sppDone - 1;
            unsigned nowTime = (unsigned)std::chrono::duration_cast<std::chrono::seconds>(
                                   std::chrono::steady_clock::now() - t_start).count();
            showCurrentResult(pixelColors, camera.width, camera.height, &nowTime, &s);
        }
    }
    mean();
    showCurrentResult(pixelColors, camera.width, camera.height);
    return pixelColors;
}

// accumulator file: "MIRTACC1", then uint32 {width, height, sppDone, seed, maxDepth}, then W*H*3 fp32 sums
void PathTracerHIP::saveState(const std::string& fileName, const Camera& camera) const {
    const size_t npix = (size_t)camera.width * camera.height;
    if (sums.size() != 3 * npix) throw std::runtime_error("saveState: no accumulators for this camera");
    FILE* f = fopen(fileName.c_str(), "wb");
    if (!f) throw std::runtime_error("saveState: cannot open " + fileName);
    const uint32_t hdr[5] = {(uint32_t)camera.width, (uint32_t)camera.height, sppDone, seed, maxDepth};
    bool ok = fwrite("MIRTACC1", 1, 8, f) == 8 && fwrite(hdr, 4, 5, f) == 5 &&
              fwrite(sums.data(), sizeof(float), sums.size(), f) == sums.size();
    ok = (fclose(f) == 0) && ok;
    if (!ok) throw std::runtime_error("saveState: short write to " + fileName);
}

void PathTracerHIP::loadState(const std::string& fileName, const Camera& camera) {
    FILE* f = fopen(fileName.c_str(), "rb");
    if (!f) throw std::runtime_error("loadState: cannot open " + fileName);
    char magic[8];
    uint32_t hdr[5];
    std::vector<float> in;
    bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, "MIRTACC1", 8) == 0 && fread(hdr, 4, 5, f) == 5;
    if (ok) {
        if (hdr[0] != (uint32_t)camera.width || hdr[1] != (uint32_t)camera.height || hdr[3] != seed || hdr[4] != maxDepth) {
            fclose(f);
            throw std::runtime_error("loadState: " + fileName + " was written for another image size, seed or maxDepth");
        }
        in.resize((size_t)hdr[0] * hdr[1] * 3);
        ok = fread(in.data(), sizeof(float), in.size(), f) == in.size();
    }
    fclose(f);
    if (!ok) throw std::runtime_error("loadState: " + fileName + " is not an accumulator file");
    sums.swap(in);
    sppDone = hdr[2];
    resume = true;
}

// one context: mirt_render_pt; several (IntersectionHIP::setDevices): sample shards over all of them + one reduce
int PathTracerHIP::launch(MirtContext* ctx, const MirtRenderParams* p, float* rgb_sum, MirtRenderStats* stats) {
    const std::vector<MirtContext*>& all = IntersectionHIP::contexts();
    if (all.size() > 1 && all[0] == ctx) return mirt_render_pt_multi(all.data(), (int)all.size(), p, rgb_sum, stats);
    return mirt_render_pt(ctx, p, rgb_sum, stats);
}

int PathTracerTestHIP::launch(MirtContext* ctx, const MirtRenderParams* p, float* rgb_sum, MirtRenderStats* stats) {
    const std::vector<MirtContext*>& all = IntersectionHIP::contexts();
    if (all.size() > 1 && all[0] == ctx) return mirt_render_pttest_multi(all.data(), (int)all.size(), p, rgb_sum, stats);
    return mirt_render_pttest(ctx, p, rgb_sum, stats);
}

// ---- Renderer ---------------------------------------------------------------------------------------
Renderer::Renderer() { configManager = new ConfigManager(this); }

Renderer::~Renderer() {
    delete mcRenderer;
    delete configManager;
}

void Renderer::loadConfig(const std::string& configFilePath) { configManager->load(configFilePath); }

std::vector<vec3f> Renderer::render() {
    if (!mcRenderer) throw std::runtime_error("Renderer::render: no integrator configured");
    return mcRenderer->renderPixels(camera);  // Renderer.cpp:46-50
}

}  // namespace mirt_host
