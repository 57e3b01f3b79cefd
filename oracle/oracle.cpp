// oracle.cpp — CPU restatement of the reference hot path. TEST INFRASTRUCTURE ONLY (see oracle.h).
// PARITY UNPINNED (see oracle.h): no reference fixtures exist and the reference cannot be built here.
//
// Written in the reference's own shape so it can be read beside the files it cites:
//   tree build      KDTree.cpp:129-246      box test     KDTree.cpp:34-52
//   triangle test   KDTree.cpp:267-293      traversal    KDTree.cpp:295-339
//   flatten         KDTree.cpp:346-383      predicate    NoSelfIntersectionCondition.cpp:12-27
//   normals/frames  SimpleShape.cpp:300-339, LocalFrame.h:14-28
//   materials       DiffuseMaterial.cpp:4-76, GlossyMaterial.cpp:4-56, RefractiveMaterial.cpp:4-104,
//                   SceneEmissiveObject.cpp:4-19, SceneVPMObject.cpp:4-237 (homogeneous medium),
//                   Scene.cpp:287-320, 354-390 (per-object trees, findInsideObject)
//   path sampling   MCRenderer.cpp:149-224  accumulation PathTracer.cpp:23-42
//   camera          Camera.cpp:4-48,60-78,92-97
// The one deliberate difference: the reference's global MT19937 (RandGenerator.cpp:9, racy under
// OpenMP) is replaced by the counter-based stream the product uses (DESIGN.md "RNG"), consumed in
// the same order the reference consumes its generator.
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <unordered_set>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ---- nvVector.h semantics --------------------------------------------------------------------
struct V3 {
    float x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(float a, float b, float c) : x(a), y(b), z(c) {}
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
};
inline V3 operator+(const V3& a, const V3& b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(const V3& a, const V3& b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(const V3& a) { return V3(-a.x, -a.y, -a.z); }
inline V3 operator*(const V3& a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, const V3& a) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(const V3& a, const V3& b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
// nvVector.h:376-380: dividing by zero leaves the vector unchanged
inline V3 operator/(const V3& a, float d) {
    if (d == 0) return a;
    return V3(a.x / d, a.y / d, a.z / d);
}
// nvVector.h:501-507: r = 0; r += x*x; r += y*y; r += z*z
inline float dot(const V3& a, const V3& b) {
    float r = 0;
    r += a.x * b.x;
    r += a.y * b.y;
    r += a.z * b.z;
    return r;
}
// nvVector.h:490-498
inline V3 cross(const V3& l, const V3& r) {
    return V3(l.y * r.z - l.z * r.y, l.z * r.x - l.x * r.z, l.x * r.y - l.y * r.x);
}
inline float length(const V3& a) { return std::sqrt(dot(a, a)); }
// nvVector.h:522-526
inline void normalize(V3& a) {
    float r = length(a);
    if (r > 0) a = a / r;
}

const double REF_PI = 3.14159265358979;  // macros.h:9 (a double literal)
const float COS_TERM_MIN = 1e-6f;         // macros.h:7

// ---- counter RNG (DESIGN.md "RNG") ------------------------------------------------------------
inline uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
struct Stream {
    uint32_t key, ctr;
    Stream(uint32_t seed, uint32_t pixel, uint32_t sample) {
        uint32_t h1 = lowbias32(sample + 0x9E3779B9U * (seed + 1U));
        key = lowbias32(pixel ^ h1);
        ctr = 0;
    }
    uint32_t bits() { return lowbias32(key + 0x9E3779B9U * (ctr++)); }
    // RandGenerator.cpp:21-25 keeps 24 bits: value in [0,1) on a 2^-24 grid
    float genFloat() { return float(bits() >> 8) / float(1 << 24); }
};

// ---- tree (KDTree.h:61-72) ----------------------------------------------------------------------
struct Node {
    Node* left = nullptr;
    Node* right = nullptr;
    std::vector<unsigned> triangleIndices;
    V3 minCoord, maxCoord;
};

struct Tri {
    unsigned v[3];
};

struct KRay {
    V3 origin, direction;
};

}  // namespace

struct OrcScene {
    std::vector<V3> verts;
    std::vector<Tri> tris;
    std::vector<int32_t> triObj, triLocal;
    std::vector<V3> triVn;  // 3 per triangle, may be empty
    std::vector<OrcObject> objects;
    // textures (orc_scene_set_textures): corner texture coordinates, 3 per triangle (empty: none)
    struct Tex {
        int width = 1, height = 1, channels = 3;
        std::vector<float> texels;
        float coordT[16], colorT[16];
        float coordInv[16];  // inverse(coordTransform), for Texture::getGrad
    };
    std::vector<V3> triVt;
    std::vector<Tex> textures;
    std::vector<OrcObjectTex> objectTex;  // empty: no object reads a texture
    Node* root = nullptr;
    std::vector<Node*> objRoots;  // Scene::objKDTrees (Scene.cpp:287-320), volumetric objects only
    // flattened view (pre-order) used by the ordered-traversal restatement
    std::vector<V3> fmin, fmax;
    std::vector<int32_t> fright, fbegin, fend, slotTid;
    int depth = 0;
};

namespace {

void destroyNode(Node* n) {
    if (!n) return;
    destroyNode(n->left);
    destroyNode(n->right);
    delete n;
}

// KDTree.cpp:129-228
void splitBest(OrcScene& S, Node* node, int depth, int& maxDepthSeen) {
    if (depth > maxDepthSeen) maxDepthSeen = depth;
    float bestRatio = 10;
    unsigned bestDim = 0;
    float bestSplitValue = 0;
    for (unsigned dim = 0; dim < 3; dim++) {
        node->left = node->right = nullptr;
        if (node->triangleIndices.size() == 0) return;
        node->minCoord = node->maxCoord = S.verts[S.tris[node->triangleIndices[0]].v[0]];
        std::unordered_set<unsigned> nodeVertexIndices;
        for (size_t i = 0; i < node->triangleIndices.size(); i++) {
            for (unsigned k = 0; k < 3; k++) {
                unsigned vi = S.tris[node->triangleIndices[i]].v[k];
                const V3& p = S.verts[vi];
                for (int c = 0; c < 3; c++) {
                    node->minCoord[c] = p[c] < node->minCoord[c] ? p[c] : node->minCoord[c];
                    node->maxCoord[c] = p[c] > node->maxCoord[c] ? p[c] : node->maxCoord[c];
                }
                nodeVertexIndices.insert(vi);
            }
        }
        // maxDepth = (unsigned)-1 and maxLeafTriangleNum = 1 (KDTree.cpp:22-27)
        if (node->triangleIndices.size() <= 1) return;

        std::vector<float> coordList;
        coordList.reserve(nodeVertexIndices.size());
        for (unsigned vi : nodeVertexIndices) coordList.push_back(S.verts[vi][dim]);
        std::nth_element(coordList.begin(), coordList.begin() + (coordList.size() >> 1),
                         coordList.end());
        float splitValue = coordList[coordList.size() >> 1];

        unsigned leftCount = 0;
        for (size_t i = 0; i < node->triangleIndices.size(); i++) {
            unsigned mask = 0;
            for (unsigned k = 0; k < 3; k++) {
                const V3& p = S.verts[S.tris[node->triangleIndices[i]].v[k]];
                mask += (p[dim] <= splitValue) << k;
            }
            if (mask != 0) leftCount++;
        }
        float ratio = leftCount / float(node->triangleIndices.size());
        // KDTree.cpp:191: float - double literal, std::abs(double)
        if (std::abs(ratio - 0.5) < std::abs(bestRatio - 0.5)) {
            bestRatio = ratio;
            bestDim = dim;
            bestSplitValue = splitValue;
        }
    }
    if (bestRatio == 0 || bestRatio == 1) {
        node->left = node->right = nullptr;
        return;
    }
    node->left = new Node;
    node->right = new Node;
    for (size_t i = 0; i < node->triangleIndices.size(); i++) {
        unsigned mask = 0;
        for (unsigned k = 0; k < 3; k++) {
            const V3& p = S.verts[S.tris[node->triangleIndices[i]].v[k]];
            mask += (p[bestDim] <= bestSplitValue) << k;
        }
        if (mask == 0)
            node->right->triangleIndices.push_back(node->triangleIndices[i]);
        else
            node->left->triangleIndices.push_back(node->triangleIndices[i]);
    }
    splitBest(S, node->left, depth + 1, maxDepthSeen);
    splitBest(S, node->right, depth + 1, maxDepthSeen);
}

// KDTree.cpp:54-127: the split axis is handed down ((dim+1)%3 per level) instead of chosen
void splitLoop(OrcScene& S, Node* node, unsigned dim, int depth, int& maxDepthSeen) {
    if (depth > maxDepthSeen) maxDepthSeen = depth;
    node->left = node->right = nullptr;
    if (node->triangleIndices.size() == 0) return;
    node->minCoord = node->maxCoord = S.verts[S.tris[node->triangleIndices[0]].v[0]];
    std::unordered_set<unsigned> nodeVertexIndices;
    for (size_t i = 0; i < node->triangleIndices.size(); i++) {
        for (unsigned k = 0; k < 3; k++) {
            unsigned vi = S.tris[node->triangleIndices[i]].v[k];
            const V3& p = S.verts[vi];
            for (int c = 0; c < 3; c++) {
                node->minCoord[c] = p[c] < node->minCoord[c] ? p[c] : node->minCoord[c];
                node->maxCoord[c] = p[c] > node->maxCoord[c] ? p[c] : node->maxCoord[c];
            }
            nodeVertexIndices.insert(vi);
        }
    }
    if (node->triangleIndices.size() <= 1) return;  // maxLeafTriangleNum = 1; maxDepth = (unsigned)-1 never binds
    std::vector<float> coordList;
    for (unsigned vi : nodeVertexIndices) coordList.push_back(S.verts[vi][dim]);
    std::nth_element(coordList.begin(), coordList.begin() + (coordList.size() >> 1), coordList.end());
    float splitValue = coordList[coordList.size() >> 1];
    node->left = new Node;
    node->right = new Node;
    unsigned leftCount = 0;
    for (size_t i = 0; i < node->triangleIndices.size(); i++) {
        unsigned mask = 0;
        for (unsigned k = 0; k < 3; k++) {
            const V3& p = S.verts[S.tris[node->triangleIndices[i]].v[k]];
            mask += (p[dim] <= splitValue) << k;
        }
        if (mask == 0)
            node->right->triangleIndices.push_back(node->triangleIndices[i]);
        else {
            node->left->triangleIndices.push_back(node->triangleIndices[i]);
            leftCount++;
        }
    }
    if (leftCount == 0 || leftCount == node->triangleIndices.size()) {
        delete node->left;
        delete node->right;
        node->left = node->right = nullptr;
        return;
    }
    splitLoop(S, node->left, (dim + 1) % 3, depth + 1, maxDepthSeen);
    splitLoop(S, node->right, (dim + 1) % 3, depth + 1, maxDepthSeen);
}

// KDTree.cpp:34-52
float boxIntersect(const V3& minC, const V3& maxC, const KRay& ray) {
    V3 intPoint;
    float minDist = -1;
    for (unsigned i = 0; i < 6; i++) {
        if (ray.direction[i % 3] == 0) continue;
        float plane = i < 3 ? minC[i] : maxC[i - 3];
        float dist = (plane - ray.origin[i % 3]) / ray.direction[i % 3];
        intPoint = ray.origin + dist * ray.direction;
        bool inter = dist >= 0;
        inter &= minC[(i + 1) % 3] <= intPoint[(i + 1) % 3];
        inter &= maxC[(i + 1) % 3] >= intPoint[(i + 1) % 3];
        inter &= minC[(i + 2) % 3] <= intPoint[(i + 2) % 3];
        inter &= maxC[(i + 2) % 3] >= intPoint[(i + 2) % 3];
        minDist = (inter && (dist < minDist || minDist < 0)) ? dist : minDist;
    }
    return minDist;
}

// SimpleShape.cpp:300-339 (world positions already transformed; vns already multiplied by the
// normal matrix on the host, SimpleShape.cpp:310,322)
V3 faceNormalOf(const V3 vps[3]) {
    V3 b1 = vps[1] - vps[0];
    V3 b2 = vps[2] - vps[0];
    V3 faceNormal = cross(b1, b2);
    normalize(faceNormal);
    return faceNormal;
}
V3 interpNormal(const V3 vps[3], const V3 vns[3], const V3& position) {  // SimpleShape.cpp:329-337
    V3 b1 = vps[1] - vps[0];
    V3 b2 = vps[2] - vps[0];
    V3 v = position - vps[0];
    float d12 = dot(b1, b2);
    float l1 = length(b1);
    float l2 = length(b2);
    float u2 = (dot(v, b1) * d12 - dot(v, b2) * l1 * l1) / (d12 * d12 - l1 * l1 * l2 * l2);
    float u1 = (dot(v, b1) - u2 * d12) / (l1 * l1);
    V3 normal = (1 - u1 - u2) * vns[0] + u1 * vns[1] + u2 * vns[2];
    normalize(normal);
    return normal;
}
// SimpleShape::getWorldNormal (the unperturbed normal)
V3 shapeNormal(const OrcScene& S, unsigned tid, const V3& position) {
    V3 vps[3];
    for (int i = 0; i < 3; i++) vps[i] = S.verts[S.tris[tid].v[i]];
    const OrcObject& obj = S.objects[S.triObj[tid]];
    if (!obj.has_vn || S.triVn.empty()) return faceNormalOf(vps);
    return interpNormal(vps, &S.triVn[3 * (size_t)tid], position);
}
V3 worldNormal(const OrcScene& S, unsigned tid, const V3& position);  // SceneObject::getWorldNormal, below

// SimpleShape::getTexCoord (SimpleShape.cpp:274-298): barycentric weights from world positions, like
// the normal interpolation; (0,0,0) for a shape without texture coordinates
V3 texCoord(const OrcScene& S, unsigned tid, const V3& position) {
    if (S.triVt.empty()) return V3(0, 0, 0);
    V3 vps[3];
    for (int i = 0; i < 3; i++) vps[i] = S.verts[S.tris[tid].v[i]];
    const V3* vts = &S.triVt[3 * (size_t)tid];
    V3 b1 = vps[1] - vps[0];
    V3 b2 = vps[2] - vps[0];
    V3 v = position - vps[0];
    float d12 = dot(b1, b2);
    float l1 = length(b1);
    float l2 = length(b2);
    float u2 = (dot(v, b1) * d12 - dot(v, b2) * l1 * l1) / (d12 * d12 - l1 * l1 * l2 * l2);
    float u1 = (dot(v, b1) - u2 * d12) / (l1 * l1);
    return (1 - u1 - u2) * vts[0] + u1 * vts[1] + u2 * vts[2];
}
// matrix4 * vec4(v, w), xyz of the result (nvMatrix.h operator*: row . vector, accumulated left to right)
V3 applyMat(const float m[16], const V3& v, float w) {
    V3 r;
    for (int i = 0; i < 3; i++) r[i] = v.x * m[4 * i] + v.y * m[4 * i + 1] + v.z * m[4 * i + 2] + w * m[4 * i + 3];
    return r;
}
// Texture::getColor (Texture.cpp:43-94). The gray branch reads its fourth texel at (top, LEFT) as
// the reference does (Texture.cpp:86).
V3 textureColor(const OrcScene::Tex& T, const V3& coord) {
    V3 tc = applyMat(T.coordT, coord, 1.f);
    float u = tc.x * T.width - 0.5;
    float v = tc.y * T.height - 0.5;
    int left = (int)floor(u);
    int right = left + 1;
    int bottom = (int)floor(v);
    int top = bottom + 1;
    u -= left;
    v -= bottom;
    left %= T.width;
    right %= T.width;
    top %= T.height;
    bottom %= T.height;
    if (left < 0) left += T.width;
    if (right < 0) right += T.width;
    if (top < 0) top += T.height;
    if (bottom < 0) bottom += T.height;
    if (T.channels == 3) {
        auto at = [&](int y, int x) {
            const float* p = &T.texels[3 * ((size_t)y * T.width + x)];
            return V3(p[0], p[1], p[2]);
        };
        V3 color(0, 0, 0);
        color = color + at(bottom, left) * (1 - u) * (1 - v);
        color = color + at(bottom, right) * u * (1 - v);
        color = color + at(top, left) * (1 - u) * v;
        color = color + at(top, right) * u * v;
        return applyMat(T.colorT, color, 1.f);
    }
    auto at = [&](int y, int x) { return T.texels[(size_t)y * T.width + x]; };
    float color = 0;
    color += at(bottom, left) * (1 - u) * (1 - v);
    color += at(bottom, right) * u * (1 - v);
    color += at(top, left) * (1 - u) * v;
    color += at(top, left) * u * v;
    return applyMat(T.colorT, V3(color, color, color), 1.f);
}

// inverse(matrix4) of nvMatrix.h:233-369: Gauss-Jordan with scaled partial pivoting in fp32; identity for a
// singular matrix. Row-major in and out.
void invertMat(const float m[16], float out[16]) {
    for (int i = 0; i < 16; i++) out[i] = (i % 5 == 0) ? 1.f : 0.f;
    float rows[4][8];
    float* s[4] = {rows[0], rows[1], rows[2], rows[3]};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            s[i][j] = m[4 * i + j];
            s[i][j + 4] = i == j ? 1.f : 0.f;
        }
    float scp[4];
    for (int i = 0; i < 4; i++) {
        scp[i] = float(fabs(s[i][0]));
        for (int j = 1; j < 4; j++)
            if (float(fabs(s[i][j])) > scp[i]) scp[i] = float(fabs(s[i][j]));
        if (scp[i] == 0.0) return;
    }
    for (int i = 0; i < 4; i++) {
        int pivot_to = i;
        float scp_max = float(fabs(s[i][i] / scp[i]));
        for (int p = i + 1; p < 4; p++)
            if (float(fabs(s[p][i] / scp[p])) > scp_max) {
                scp_max = float(fabs(s[p][i] / scp[p]));
                pivot_to = p;
            }
        if (pivot_to != i) {
            std::swap(s[i], s[pivot_to]);
            std::swap(scp[i], scp[pivot_to]);
        }
        for (int j = i + 1; j < 4; j++) {
            float mji = s[j][i] / s[i][i];
            s[j][i] = 0.0;
            for (int jj = i + 1; jj < 8; jj++) s[j][jj] -= mji * s[i][jj];
        }
    }
    if (s[3][3] == 0.0) return;
    for (int i = 3; i > 0; i--)
        for (int j = i - 1; j > -1; j--) {
            float mij = s[j][i] / s[i][i];
            for (int jj = j + 1; jj < 8; jj++) s[j][jj] -= mij * s[i][jj];
        }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) out[4 * i + j] = s[i][j + 4] / s[i][i];
}
// Texture::getGrad (Texture.cpp:96-147)
V3 textureGrad(const OrcScene::Tex& T, const V3& coord) {
    V3 tc = applyMat(T.coordT, coord, 1.f);
    float u = tc.x * T.width - 0.5;
    float v = tc.y * T.height - 0.5;
    int left = (int)floor(u);
    int right = left + 1;
    int bottom = (int)floor(v);
    int top = bottom + 1;
    left %= T.width;
    right %= T.width;
    top %= T.height;
    bottom %= T.height;
    if (left < 0) left += T.width;
    if (right < 0) right += T.width;
    if (top < 0) top += T.height;
    if (bottom < 0) bottom += T.height;
    auto at = [&](int y, int x) { return T.texels[(size_t)T.channels * ((size_t)y * T.width + x)]; };  // .x of an rgb texel
    V3 image_grad(0, 0, 0);
    float lbColor = at(bottom, left);
    image_grad.x = at(bottom, right) - lbColor;
    image_grad.y = at(top, left) - lbColor;
    float var_n = length(image_grad);
    var_n = applyMat(T.colorT, V3(var_n, var_n, var_n), 0.f).x;
    normalize(image_grad);
    V3 uv_grad = applyMat(T.coordInv, image_grad, 0.f);
    uv_grad.z = var_n;
    return uv_grad;
}
// smallFuncs.h:11-24 applied to a direction (w = 0)
V3 rotateDir(const V3& axis, float angle, const V3& v) {
    float c = std::cos(angle), s = std::sin(angle);
    float _c = 1 - c;
    float x = axis.x, y = axis.y, z = axis.z;
    float m00 = c + _c * x * x, m01 = _c * x * y - s * z, m02 = _c * x * z + s * y;
    float m10 = _c * x * y + s * z, m11 = c + _c * y * y, m12 = _c * y * z - s * x;
    float m20 = _c * x * z - s * y, m21 = _c * y * z + s * x, m22 = c + _c * z * z;
    // nvMatrix.h:415-424 with src.w = 0
    return V3(v.x * m00 + v.y * m01 + v.z * m02 + 0.f * 0.f, v.x * m10 + v.y * m11 + v.z * m12 + 0.f * 0.f,
              v.x * m20 + v.y * m21 + v.z * m22 + 0.f * 0.f);
}
// SceneObject::getWorldNormal (SceneObject.cpp:250-283)
V3 worldNormal(const OrcScene& S, unsigned tid, const V3& position) {
    V3 original_normal = shapeNormal(S, tid, position);
    if (S.objectTex.empty() || S.triVt.empty()) return original_normal;
    const int bump = S.objectTex[S.triObj[tid]].bump;
    if (bump < 0 || S.textures[bump].width * S.textures[bump].height <= 1) return original_normal;
    V3 vps[3];
    for (int i = 0; i < 3; i++) vps[i] = S.verts[S.tris[tid].v[i]];
    const V3* vts = &S.triVt[3 * (size_t)tid];
    V3 uv_grad = textureGrad(S.textures[bump], texCoord(S, tid, position));
    V3 b1 = vps[1] - vps[0];
    V3 b2 = vps[2] - vps[0];
    V3 duv1 = vts[1] - vts[0];
    V3 duv2 = vts[2] - vts[0];
    float k2 = (uv_grad.x * duv1.y - uv_grad.y * duv1.x) / (duv1.y * duv2.x - duv1.x * duv2.y);
    float k1 = (uv_grad.y - k2 * duv2.y) / duv1.y;
    normalize(b1);
    normalize(b2);
    V3 dl = k1 * b1 + k2 * b2;
    V3 dh = original_normal * uv_grad.z;
    if (length(dh) * 1000 < length(dl)) return original_normal;
    float angle = atan2(length(dh), length(dl));
    V3 axis = cross(dl, dh);
    normalize(axis);
    return rotateDir(axis, angle, original_normal);
}

// NoSelfIntersectionCondition.cpp:12-27
struct Condition {
    int mode = 0;  // 0 none, 1 last_tid, 2 NoSelfIntersection
    int lastTid = -1;
    int contact = -1, inside = -1;
    bool legal(const OrcScene& S, const KRay& ray, unsigned tid, float dist) const {
        if (mode == 0) return true;
        if (mode == 1) return (int)tid != lastTid;
        int hitObj = S.triObj[tid];
        float d = dot(ray.direction, worldNormal(S, tid, ray.origin + ray.direction * dist));
        bool in = d <= 0;
        bool out = d >= 0;
        if (inside == contact) return !(in && contact == hitObj);
        return !(out && inside != hitObj);
    }
};

// KDTree.cpp:267-293
float triIntersect(const OrcScene& S, const KRay& ray, unsigned tid, const Condition* cond) {
    V3 vs[3], lines[3], r2v, n;
    for (unsigned i = 0; i < 3; i++) vs[i] = S.verts[S.tris[tid].v[i]];
    unsigned mask_pos = 0, mask_neg = 0;
    for (unsigned i = 0; i < 3; i++) {
        lines[i] = vs[(i + 1) % 3] - vs[i];
        r2v = vs[i] - ray.origin;
        n = cross(lines[i], r2v);
        mask_pos += (dot(ray.direction, n) >= 0) << i;
        mask_neg += (dot(ray.direction, n) <= 0) << i;
    }
    n = cross(lines[0], lines[1]);
    normalize(n);
    float orthoDist = dot(n, r2v);
    float d = dot(n, ray.direction);
    if (d == 0 || orthoDist / d < 0 || (mask_pos != 7 && mask_neg != 7)) return -1;
    float dist = orthoDist / d;
    if (cond && !cond->legal(S, ray, tid, dist)) return -1;
    return dist;
}

struct Counters {
    uint64_t box = 0, tri = 0;
};

// KDTree.cpp:295-333
float nodeIntersect(const OrcScene& S, const KRay& ray, const Node* node, unsigned& triangleID,
                    const Condition* cond, Counters* cnt) {
    if (cnt) cnt->box++;
    if (boxIntersect(node->minCoord, node->maxCoord, ray) < 0) return -1;
    if (node->left == nullptr && node->right == nullptr) {
        float minDist = -1;
        for (size_t ti = 0; ti < node->triangleIndices.size(); ti++) {
            if (cnt) cnt->tri++;
            float dist = triIntersect(S, ray, node->triangleIndices[ti], cond);
            if (dist >= 0 && (dist < minDist || minDist < 0)) {
                minDist = dist;
                triangleID = node->triangleIndices[ti];
            }
        }
        return minDist;
    }
    float ld, rd;
    unsigned ltid = 0, rtid = 0;
    ld = nodeIntersect(S, ray, node->left, ltid, cond, cnt);
    rd = nodeIntersect(S, ray, node->right, rtid, cond, cnt);
    if (ld >= 0 && rd >= 0) {
        triangleID = ld < rd ? ltid : rtid;
        return ld < rd ? ld : rd;
    } else if (rd < 0) {
        triangleID = ltid;
        return ld;
    } else if (ld < 0) {
        triangleID = rtid;
        return rd;
    }
    return -1;
}

// KDTree.cpp:346-377 (integers kept as integers; triangles referenced by slot -> global id)
void flattenNode(OrcScene& S, const Node* node) {
    S.fmin.push_back(node->minCoord);
    S.fmax.push_back(node->maxCoord);
    if (node->left && node->right) {
        S.fright.push_back(-1);
        S.fbegin.push_back(-1);
        S.fend.push_back(-1);
        size_t me = S.fright.size() - 1;
        flattenNode(S, node->left);
        S.fright[me] = (int32_t)S.fright.size();
        flattenNode(S, node->right);
    } else {
        int32_t si = (int32_t)S.slotTid.size();
        for (unsigned t : node->triangleIndices) S.slotTid.push_back((int32_t)t);
        S.fright.push_back(-1);
        S.fbegin.push_back(si);
        S.fend.push_back((int32_t)S.slotTid.size());
    }
}

Condition makeCond(int mode, int32_t aux) {
    Condition c;
    c.mode = mode;
    if (mode == 1) c.lastTid = aux;
    if (mode == 2) {
        c.contact = (int)((uint32_t)aux & 0xFFFFu) - 1;
        c.inside = (int)(((uint32_t)aux >> 16) & 0xFFFFu) - 1;
    }
    return c;
}

// ---- the product's traversal order, restated for visit counts (DESIGN.md "intersect kernel") ----
struct BoxRes {
    bool hit;
    float lb;
};
BoxRes boxOrdered(const V3& minC, const V3& maxC, const KRay& ray) {
    BoxRes r;
    r.hit = boxIntersect(minC, maxC, ray) >= 0;
    float lb = 0;
    for (int a = 0; a < 3; a++) {
        if (ray.direction[a] == 0) continue;
        float t0 = (minC[a] - ray.origin[a]) / ray.direction[a];
        float t1 = (maxC[a] - ray.origin[a]) / ray.direction[a];
        float nr = t0 < t1 ? t0 : t1;
        lb = nr > lb ? nr : lb;
    }
    r.lb = lb;
    return r;
}
// (mirt_dev.h exact_prune_slack: how far in front of its leaf's box the reference's t can lie for a ray that barely
// moves along an axis; the same fp32 operations in the same order)
float exactPruneSlack(const OrcScene& S, const KRay& ray) {
    float m = 0;
    for (int a = 0; a < 3; a++) {
        float ia = std::fabs(1.0f / ray.direction[a]);
        if (!(ia < INFINITY)) continue;
        float bound = std::max(std::fabs(S.fmin[0][a]), std::fabs(S.fmax[0][a]));
        float v = (std::fabs(ray.origin[a]) + bound) * ia;
        m = v > m ? v : m;
    }
    return 1.9073486328125e-06f * m;
}
inline bool pruned(float lb, bool have, float best, float slack) { return have && lb > best * 1.0001f + 1e-5f + slack; }

float orderedIntersect(const OrcScene& S, const KRay& ray, const Condition& cond, int& slotOut,
                       Counters& cnt) {
    float best = -1;
    int bestSlot = -1;
    struct E {
        int node;
        float lb;
    };
    std::vector<E> stack;
    const float slack = exactPruneSlack(S, ray);
    cnt.box++;
    if (!boxOrdered(S.fmin[0], S.fmax[0], ray).hit) {
        slotOut = -1;
        return -1;
    }
    int cur = 0;
    for (;;) {
        bool pop = false;
        if (S.fright[cur] < 0) {
            int lb = S.fbegin[cur], le = S.fend[cur];
            for (int s = lb; s < le; s++) {
                cnt.tri++;
                unsigned tid = (unsigned)S.slotTid[s];
                float t = triIntersect(S, ray, tid, nullptr);
                if (!(t >= 0)) continue;
                bool wins = bestSlot < 0 || t < best || (t == best && bestSlot < lb);
                if (wins && cond.legal(S, ray, tid, t)) {
                    best = t;
                    bestSlot = s;
                }
            }
            pop = true;
        } else {
            int L = cur + 1, R = S.fright[cur];
            cnt.box += 2;
            BoxRes bl = boxOrdered(S.fmin[L], S.fmax[L], ray);
            BoxRes br = boxOrdered(S.fmin[R], S.fmax[R], ray);
            bool hl = bl.hit && !pruned(bl.lb, bestSlot >= 0, best, slack);
            bool hr = br.hit && !pruned(br.lb, bestSlot >= 0, best, slack);
            if (hl && hr) {
                if (bl.lb <= br.lb) {
                    stack.push_back({R, br.lb});
                    cur = L;
                } else {
                    stack.push_back({L, bl.lb});
                    cur = R;
                }
            } else if (hl) {
                cur = L;
            } else if (hr) {
                cur = R;
            } else {
                pop = true;
            }
        }
        if (pop) {
            bool found = false;
            while (!stack.empty()) {
                E e = stack.back();
                stack.pop_back();
                if (pruned(e.lb, bestSlot >= 0, best, slack)) continue;
                cur = e.node;
                found = true;
                break;
            }
            if (!found) break;
        }
    }
    slotOut = bestSlot;
    return bestSlot >= 0 ? best : -1;
}

// ---- the product's FAST path, restated (DESIGN.md "fast tree"): near-first pruned traversal of a tree
// GIVEN by the caller whose leaves carry slots of this scene's reference tree (single triangles, or any ranges
// that lie inside one reference leaf), entering every box a
// generously inflated slab test cannot exclude; then the winner's chain of reference boxes (leaf up to the
// root) is checked with the reference box test, and a ray whose chain fails is answered by the exact
// ordered traversal of the reference tree instead. Counts every box / triangle test on the way.
struct GivenTree {
    int n;
    const float* mn;
    const float* mx;
    const int32_t* right;
    const int32_t* lbeg;
    const int32_t* lend;
    V3 lo(int i) const { return V3(mn[3 * i], mn[3 * i + 1], mn[3 * i + 2]); }
    V3 hi(int i) const { return V3(mx[3 * i], mx[3 * i + 1], mx[3 * i + 2]); }
};
// conservative: false only when the ray clearly misses (slabs disjoint by a wide relative margin)
BoxRes boxConservative(const V3& minC, const V3& maxC, const KRay& ray) {
    BoxRes r;
    double tn = -INFINITY, tf = INFINITY;
    bool outside = false;
    for (int a = 0; a < 3; a++) {
        double o = ray.origin[a], d = ray.direction[a];
        if (d == 0) {
            if (o < (double)minC[a] - 1e-4 * (1 + std::fabs(o)) || o > (double)maxC[a] + 1e-4 * (1 + std::fabs(o))) outside = true;
            continue;
        }
        double t0 = (minC[a] - o) / d, t1 = (maxC[a] - o) / d;
        tn = std::max(tn, std::min(t0, t1));
        tf = std::min(tf, std::max(t0, t1));
    }
    const double m = 1e-3 * (1 + std::fabs(tn) + std::fabs(tf));
    r.hit = !outside && !(tn > tf + m) && !(tf < -m);
    r.lb = (float)std::max(0.0, tn - m);
    return r;
}
float fastIntersect(const OrcScene& S, const GivenTree& T, const std::vector<int>& refParent,
                    const std::vector<int>& slotNode, const KRay& ray, const Condition& cond, int& slotOut,
                    Counters& cnt) {
    float best = -1;
    int bestSlot = -1;
    struct E {
        int node;
        float lb;
    };
    std::vector<E> stack;
    cnt.box++;
    bool alive = boxConservative(T.lo(0), T.hi(0), ray).hit;
    int cur = 0;
    while (alive) {
        bool pop = false;
        if (T.right[cur] < 0) {
            int lb = T.lbeg[cur], le = T.lend[cur];
            for (int s = lb; s < le; s++) {
                cnt.tri++;
                unsigned tid = (unsigned)S.slotTid[s];
                float t = triIntersect(S, ray, tid, nullptr);
                if (!(t >= 0)) continue;
                // ties follow the reference's order whatever the given tree's leaves are: between reference leaves
                // the later one wins (KDTree.cpp:319-320), inside one the earlier triangle (KDTree.cpp:305)
                auto rank = [&](int q) { return S.fbegin[slotNode[q]] + (S.fend[slotNode[q]] - 1 - q); };
                bool wins = bestSlot < 0 || t < best || (t == best && rank(s) > rank(bestSlot));
                if (wins && cond.legal(S, ray, tid, t)) {
                    best = t;
                    bestSlot = s;
                }
            }
            pop = true;
        } else {
            int L = cur + 1, R = T.right[cur];
            cnt.box += 2;
            BoxRes bl = boxConservative(T.lo(L), T.hi(L), ray);
            BoxRes br = boxConservative(T.lo(R), T.hi(R), ray);
            bool hl = bl.hit && !pruned(bl.lb, bestSlot >= 0, best, 0.f);
            bool hr = br.hit && !pruned(br.lb, bestSlot >= 0, best, 0.f);
            if (hl && hr) {
                if (bl.lb <= br.lb) {
                    stack.push_back({R, br.lb});
                    cur = L;
                } else {
                    stack.push_back({L, bl.lb});
                    cur = R;
                }
            } else if (hl) {
                cur = L;
            } else if (hr) {
                cur = R;
            } else {
                pop = true;
            }
        }
        if (pop) {
            bool found = false;
            while (!stack.empty()) {
                E e = stack.back();
                stack.pop_back();
                if (pruned(e.lb, bestSlot >= 0, best, 0.f)) continue;
                cur = e.node;
                found = true;
                break;
            }
            if (!found) break;
        }
    }
    if (bestSlot >= 0) {  // the winner's chain of REFERENCE boxes, by the reference test
        for (int node = slotNode[bestSlot]; node >= 0;) {
            cnt.box++;
            const V3 &mn = S.fmin[node], &mx = S.fmax[node];
            if (!(boxIntersect(mn, mx, ray) >= 0))
                return orderedIntersect(S, ray, cond, slotOut, cnt);  // exact re-run
            // The product's k_verify does not test every ancestor: when the reference accepts a box through its exit
            // face F, an ancestor whose plane on F is the same bit for bit and whose other two slabs are no narrower
            // is accepted through F by the very same t and point (KDTree.cpp:40-49), so the walk goes on above the run
            // of such ancestors (NodeRec::jump in renderer_amd/csrc/mirt_dev.h). Restated here with the accepted face
            // of largest t found by the reference arithmetic itself. A flat box (zero extent along an axis) accepted
            // through that axis' plane, which is its min and its max plane at once, hands the acceptance on through
            // either; the product follows the run that ends higher up.
            const bool flat = mn[0] == mx[0] || mn[1] == mx[1] || mn[2] == mx[2];
            bool acc[6];
            float dists[6];
            for (unsigned i = 0; i < 6; i++) {
                acc[i] = false;
                dists[i] = -1;
                if (ray.direction[i % 3] == 0) continue;
                float plane = i < 3 ? mn[i] : mx[i - 3];
                float dist = (plane - ray.origin[i % 3]) / ray.direction[i % 3];
                V3 p = ray.origin + dist * ray.direction;
                acc[i] = dist >= 0 && mn[(i + 1) % 3] <= p[(i + 1) % 3] && mx[(i + 1) % 3] >= p[(i + 1) % 3] &&
                         mn[(i + 2) % 3] <= p[(i + 2) % 3] && mx[(i + 2) % 3] >= p[(i + 2) % 3];
                dists[i] = dist;
            }
            auto runAbove = [&](int face) {  // first ancestor that does not inherit an acceptance through `face`
                int next = refParent[node];
                while (next >= 0) {
                    const int a = face % 3;
                    const float cp = face < 3 ? mn[a] : mx[a], pp = face < 3 ? S.fmin[next][a] : S.fmax[next][a];
                    bool inherits = std::memcmp(&cp, &pp, 4) == 0;
                    for (int b : {(a + 1) % 3, (a + 2) % 3})
                        inherits = inherits && S.fmin[next][b] <= mn[b] && S.fmax[next][b] >= mx[b];
                    if (!inherits) break;
                    next = refParent[next];
                }
                return next;
            };
            int next = refParent[node];
            if (flat) {
                for (int a = 0; a < 3; a++)
                    if (mn[a] == mx[a] && acc[a] && acc[a + 3]) next = std::min(next, std::min(runAbove(a), runAbove(a + 3)));
            } else {
                int face = -1;
                float tExit = -1;
                for (int i = 0; i < 6; i++)
                    if (acc[i] && dists[i] > tExit) tExit = dists[i], face = i;
                if (face >= 0) next = runAbove(face);
            }
            node = next;
        }
    }
    slotOut = bestSlot;
    return bestSlot >= 0 ? best : -1;
}

// ---- path tracing ---------------------------------------------------------------------------------
const int CAMERA_OBJ = -2;  // Camera is a SceneObject used as contactObject (Camera.cpp:25)

// Ray.h:8-61 (fields the PathTracer path reads)
struct PRay {
    V3 origin, direction, color;
    float directionProb, originProb;
    int insideObject, contactObject, intersectObject;  // object ids, -1 = NULL
    unsigned contactObjectTriangleID, intersectObjectTriangleID;
    float intersectDist;
    bool randomDir;  // directionSampleType == Ray::RANDOM (the Ray.h:46 default)
    PRay() {  // Ray.h:37-51
        randomDir = true;
        insideObject = intersectObject = contactObject = -1;
        color = V3(1, 1, 1);
        directionProb = 1;
        originProb = 1;
        contactObjectTriangleID = intersectObjectTriangleID = 0;
        intersectDist = -1;
    }
};

// LocalFrame.h:8-34
struct LocalFrame {
    V3 s, t, n;
    void buildFromNormal(const V3& normal) {
        n = normal;
        V3 tmpT = (std::abs(n.z) > 0.99f) ? V3(1, 0, 0) : V3(0, 0, 1);
        s = cross(n, tmpT);
        normalize(s);
        t = cross(s, n);
        normalize(t);
    }
    V3 toWorld(const V3& a) const { return s * a.x + n * a.y + t * a.z; }
};

struct Tracer {
    const OrcScene& S;
    const OrcCamera& cam;
    unsigned maxDepth;
    uint64_t nRays = 0;
    V3 camPos, camFront;
    Tracer(const OrcScene& s, const OrcCamera& c, unsigned md) : S(s), cam(c), maxDepth(md) {
        camPos = V3(c.position[0], c.position[1], c.position[2]);
        camFront = V3(c.focus[0], c.focus[1], c.focus[2]) - camPos;
        normalize(camFront);
    }

    unsigned globalTid(int obj, unsigned local) const { return S.objects[obj].tri_begin + local; }

    // SceneObject::getWorldNormal, flat=false, no bump texture (SceneObject.cpp:250-254)
    V3 objNormal(int obj, unsigned localTri, const V3& pos) const {
        return worldNormal(S, globalTid(obj, localTri), pos);
    }

    // Ray.cpp:5-12, Camera.cpp:92-97
    V3 contactNormal(const PRay& r) const {
        if (r.contactObject == CAMERA_OBJ) {
            V3 front = V3(cam.focus[0], cam.focus[1], cam.focus[2]) - r.origin;
            normalize(front);
            return front;
        }
        if (r.contactObject >= 0) return objNormal(r.contactObject, r.contactObjectTriangleID, r.origin);
        return V3(0, 0, 0);
    }
    bool hasCosineTerm(int obj) const {
        if (obj == CAMERA_OBJ) return true;
        return S.objects[obj].type != ORC_VPM;  // SceneVPMObject.h: hasCosineTerm() false
    }
    // Ray.cpp:71-78
    float cosineTerm(const PRay& r) const {
        float d = 1.f;
        if (r.contactObject != -1 && hasCosineTerm(r.contactObject) && length(r.direction) > 0.5)
            d = std::abs(dot(contactNormal(r), r.direction));
        d = d > COS_TERM_MIN ? d : COS_TERM_MIN;
        return d;
    }
    // Ray.cpp:66-69, SceneObject.cpp:354-364, RefractiveMaterial.h:15
    V3 radianceDecay(const PRay& r, float dist) const {
        if (r.insideObject < 0) return V3(1, 1, 1);
        const OrcObject& o = S.objects[r.insideObject];
        if (o.type == ORC_REFRACTIVE)
            return V3(powf(o.decay[0], dist), powf(o.decay[1], dist), powf(o.decay[2], dist));
        if (o.type == ORC_VPM) return vpmTransmittance(o, dist);  // SceneVPMObject.cpp:235-237
        return V3(1, 1, 1);
    }
    bool emissive(int obj) const { return obj >= 0 && S.objects[obj].type == ORC_EMISSIVE; }

    // RandGenerator.cpp:152-168 (disk.x drawn first)
    static V3 hemiCos(float expTerm, float dx, float dy) {
        float u1 = 2.f * REF_PI * dx;
        float u2 = std::pow(dy, 1.f / (expTerm + 1.f));
        float u3 = std::sqrt(1.f - u2 * u2);
        V3 res(std::cos(u1) * u3, u2, std::sin(u1) * u3);
        normalize(res);
        return res;
    }
    // CosineSphericalSampler.h:23-34
    static float cosPdf(const LocalFrame& lf, const V3& dir, float coeff) {
        float c = dot(lf.n, dir);
        float cosTheta = c > 0.f ? c : 0.f;
        float res = (powf(cosTheta, coeff) * (coeff + 1.f) / (2.f * REF_PI));
        return res;
    }

    // Scene::findInsideObject (Scene.cpp:354-390): the last volumetric object whose own tree is
    // hit from inside (normal along the ray) both backwards and forwards
    int findInsideObject(const PRay& ray, int currentObject) const {
        KRay front, back;
        front.direction = ray.direction;
        front.origin = ray.origin;
        back.direction = -ray.direction;
        back.origin = ray.origin;
        for (int i = (int)S.objects.size() - 1; i >= 0; i--) {
            if (currentObject == i || S.objects[i].type != ORC_VPM || !S.objRoots[i]) continue;
            unsigned tid = 0;
            float dist = nodeIntersect(S, back, S.objRoots[i], tid, nullptr, nullptr);
            if (dist < 1e-6f) continue;
            V3 normal = worldNormal(S, tid, back.origin + back.direction * dist);
            if (dot(normal, back.direction) < 1e-6f) continue;
            dist = nodeIntersect(S, front, S.objRoots[i], tid, nullptr, nullptr);
            if (dist < 1e-6f) continue;
            normal = worldNormal(S, tid, front.origin + front.direction * dist);
            if (dot(normal, front.direction) < 1e-6f) continue;
            return i;
        }
        return -1;
    }

    // What the materials read at a point: the constant of the object, or its texture at
    // getTexCoord(triangle, at) (DiffuseMaterial.cpp:23,83,99; GlossyMaterial.cpp:11-12,66,74,78)
    const OrcScene::Tex* boundTex(int obj, int which) const {
        if (S.objectTex.empty()) return nullptr;
        const OrcObjectTex& b = S.objectTex[obj];
        const int k = which == 0 ? b.diffuse : (which == 1 ? b.glossy_color : b.glossy_coeff);
        return k >= 0 ? &S.textures[k] : nullptr;
    }
    V3 diffuseColor(int obj, unsigned localTri, const V3& at) const {
        const OrcObject& o = S.objects[obj];
        if (const OrcScene::Tex* T = boundTex(obj, 0)) return textureColor(*T, texCoord(S, globalTid(obj, localTri), at));
        return V3(o.color[0], o.color[1], o.color[2]);
    }
    V3 glossyColor(int obj, unsigned localTri, const V3& at) const {
        const OrcObject& o = S.objects[obj];
        if (const OrcScene::Tex* T = boundTex(obj, 1)) return textureColor(*T, texCoord(S, globalTid(obj, localTri), at));
        return o.type == ORC_PHONG ? V3(o.color2[0], o.color2[1], o.color2[2]) : V3(o.color[0], o.color[1], o.color[2]);
    }
    float glossyCoeff(int obj, unsigned localTri, const V3& at) const {
        if (const OrcScene::Tex* T = boundTex(obj, 2)) return textureColor(*T, texCoord(S, globalTid(obj, localTri), at)).x;
        return S.objects[obj].coeff;
    }

    // DiffuseMaterial.cpp:4-76
    PRay scatterDiffuse(int object, const PRay& in, Stream& rng) const {
        // DiffuseMaterial.cpp:23: the texture at the hit point
        const V3 at = in.origin + in.direction * in.intersectDist;
        return scatterDiffuse(object, diffuseColor(in.intersectObject, in.intersectObjectTriangleID, at), in, rng);
    }
    PRay scatterDiffuse(int object, const V3& color, const PRay& in, Stream& rng) const {
        PRay out;
        out.origin = in.origin + in.direction * in.intersectDist;
        LocalFrame lf;
        lf.buildFromNormal(objNormal(in.intersectObject, in.intersectObjectTriangleID, out.origin));
        V3 normal = lf.n;
        out.insideObject = in.insideObject;
        out.intersectObject = -1;
        out.contactObject = object;
        out.contactObjectTriangleID = in.intersectObjectTriangleID;
        if (dot(in.direction, normal) > 0) {  // hit from behind: pass straight through
            out.direction = in.direction;
            out.directionProb = 1;
            out.color = V3(1, 1, 1);
            out.randomDir = false;  // DiffuseMaterial.cpp:42
            return out;
        }
        float dx = rng.genFloat(), dy = rng.genFloat();
        out.direction = lf.toWorld(hemiCos(1.f, dx, dy));
        float p = maxComp(color);
        if (rng.genFloat() < p) {
            // DiffuseBSDF.h:18-24
            if (dot(lf.n, in.direction) <= 0 && dot(lf.n, out.direction) >= 0)
                out.color = color / float(REF_PI);
            else
                out.color = V3(0, 0, 0);
            out.directionProb = p * cosPdf(lf, out.direction, 1.f);
        } else {
            out.direction = V3(0, 0, 0);
            out.color = V3(0, 0, 0);
            out.directionProb = 1 - p;
        }
        return out;
    }

    // GlossyMaterial.cpp:4-56, GlossyBSDF.h:24-33
    PRay scatterGlossy(int object, const PRay& in, Stream& rng) const {
        // GlossyMaterial.cpp:11-12 look the textures up at getTexCoord(tid, outRay.origin) BEFORE outRay.origin
        // is set (:17): a default-constructed vec3f, (0,0,0) (nvVector.h:304)
        const V3 at(0, 0, 0);
        return scatterGlossy(object, glossyColor(in.intersectObject, in.intersectObjectTriangleID, at),
                             glossyCoeff(in.intersectObject, in.intersectObjectTriangleID, at), in, rng);
    }
    PRay scatterGlossy(int object, const V3& color, float coeff, const PRay& in, Stream& rng) const {
        PRay out;
        V3 position = in.origin + in.direction * in.intersectDist;
        LocalFrame lf;
        lf.buildFromNormal(objNormal(in.intersectObject, in.intersectObjectTriangleID, position));
        V3 normal = lf.n;
        out.insideObject = in.insideObject;
        out.origin = position;
        out.contactObject = object;
        out.contactObjectTriangleID = in.intersectObjectTriangleID;
        LocalFrame out_lf;
        V3 reflDir = -dot(normal, in.direction) * normal * 2 + in.direction;
        out_lf.buildFromNormal(reflDir);
        float dx = rng.genFloat(), dy = rng.genFloat();
        out.direction = out_lf.toWorld(hemiCos(coeff, dx, dy));
        out.color = V3(0, 0, 0);
        if (dot(lf.n, in.direction) <= 0 && dot(lf.n, out.direction) >= 0) {
            V3 r2 = -dot(lf.n, in.direction) * lf.n * 2 + in.direction;
            if (dot(r2, out.direction) > 0) out.color = color * powf(dot(r2, out.direction), coeff);
        }
        out.directionProb = cosPdf(out_lf, out.direction, coeff);
        float p = maxComp(color);
        if (length(out.color) == 0) {
            out.direction = V3(0, 0, 0);
            out.color = V3(0, 0, 0);
            out.directionProb = 1;
            return out;
        }
        if (rng.genFloat() < p) {
            out.directionProb *= p;
        } else {
            out.direction = V3(0, 0, 0);
            out.color = V3(0, 0, 0);
            out.directionProb = 1 - p;
        }
        return out;
    }

    // smallFuncs.h:11-24 applied to a direction (w = 0)
    static V3 rotate(const V3& axis, float angle, const V3& v) { return rotateDir(axis, angle, v); }
    static float clampf(float v, float lo, float hi) { return v < lo ? lo : v > hi ? hi : v; }
    // *std::max_element(&c.x, &c.x + 3) (DiffuseMaterial.cpp:48,84,91, GlossyMaterial.cpp:30,75,84): the first component
    // stays the largest unless a later one compares greater, which is what decides the result for a NaN component
    static float maxComp(const V3& c) {
        float m = c.x;
        if (m < c.y) m = c.y;
        if (m < c.z) m = c.z;
        return m;
    }

    // RefractiveMaterial.cpp:4-104
    PRay scatterRefractive(int object, const PRay& in, Stream& rng) const {
        PRay out;
        V3 position = in.origin + in.direction * in.intersectDist;
        const OrcObject& o = S.objects[object];
        LocalFrame lf;
        lf.buildFromNormal(objNormal(in.intersectObject, in.intersectObjectTriangleID, position));
        V3 normal = lf.n;
        out.color = V3(o.color[0], o.color[1], o.color[2]);
        out.origin = position;
        out.direction = in.direction;
        V3 reflDir = -dot(normal, in.direction) * normal * 2 + in.direction;
        out.contactObject = object;
        out.contactObjectTriangleID = in.intersectObjectTriangleID;
        out.randomDir = false;  // RefractiveMaterial.cpp:31
        if (o.ior < 0) {  // mirror
            out.direction = reflDir;
            out.insideObject = in.insideObject;
            out.directionProb = 1;
            out.color = out.color / cosineTerm(out);
            return out;
        }
        if (in.intersectObject != object && in.insideObject == object)
            return scatter(in.intersectObject, in, rng);

        float theta = std::acos(clampf(dot(in.direction, normal), -1, 1));
        int currentInside = in.insideObject;
        int outSide = object;
        if (in.insideObject == object) outSide = findInsideObject(out, object);
        float current_n = currentInside >= 0 ? refrCoeff(currentInside) : 1;
        float next_n = outSide >= 0 ? refrCoeff(outSide) : 1;
        out.intersectObject = -1;
        out.directionProb = 1;
        float sin_phi = current_n / next_n * std::sin(theta);
        if (sin_phi >= 1) {
            out.direction = reflDir;
            out.insideObject = in.insideObject;
            out.directionProb = 1;
            out.color = out.color / cosineTerm(out);
        } else {
            float phi = std::asin(sin_phi);
            if (theta > REF_PI / 2) phi = REF_PI - phi;
            V3 axis = cross(normal, in.direction);
            normalize(axis);
            out.direction = rotate(axis, phi, normal);
            normalize(out.direction);
            float cos_theta = std::abs(std::cos(theta));
            float cos_phi = std::abs(std::cos(phi));
            float esr = powf(std::abs(current_n * cos_theta - next_n * cos_phi) /
                                 (current_n * cos_theta + next_n * cos_phi), 2);
            float epr = powf(std::abs(next_n * cos_theta - current_n * cos_phi) /
                                 (next_n * cos_theta + current_n * cos_phi), 2);
            float er = (esr + epr) / 2;
            float p = clampf(er, 0.f, 1.f);
            if (rng.genFloat() < p) {
                out.direction = reflDir;
                out.color = out.color * (er / cosineTerm(out));
                out.directionProb = p;
                out.insideObject = in.insideObject;
            } else {
                out.color = out.color * ((1 - er) / cosineTerm(out));
                out.directionProb = 1 - p;
                out.insideObject = outSide;
            }
        }
        normalize(out.direction);
        return out;
    }
    // SceneObject.h:58, SceneRefractiveObject.h:22 (mirrors keep the SceneObject default 1)
    float refrCoeff(int obj) const {
        const OrcObject& o = S.objects[obj];
        if (o.type == ORC_REFRACTIVE && o.ior >= 0) return o.ior;
        if (o.type == ORC_VPM) return o.ior;
        return 1;
    }

    // SceneEmissiveObject.cpp:4-19
    PRay scatterEmissive(int object, const PRay& in) const {
        PRay out;
        V3 position = in.origin + in.direction * in.intersectDist;
        out.intersectObject = -1;
        out.insideObject = -1;
        out.contactObject = object;
        out.contactObjectTriangleID = in.intersectObjectTriangleID;
        out.origin = position;
        out.direction = V3(0, 0, 0);
        const OrcObject& o = S.objects[object];
        out.color = dot(in.direction, contactNormal(out)) <= 0 ? V3(o.color[0], o.color[1], o.color[2])
                                                               : V3(0, 0, 0);
        out.directionProb = 1;
        return out;
    }

    // ---- homogeneous medium, SceneVPMObject.h:22-47 / SceneVPMObject.cpp:4-233 --------------------
    static float lumY(const V3& c) { return 0.212671f * c[0] + 0.715160f * c[1] + 0.072169f * c[2]; }
    static V3 vpmSigmaT(const OrcObject& o) {  // setSigma: dt = ds + da
        return V3(o.sigma_s[0] + o.sigma_a[0], o.sigma_s[1] + o.sigma_a[1], o.sigma_s[2] + o.sigma_a[2]);
    }
    static V3 vpmTransmittance(const OrcObject& o, float dist) {
        V3 dt = vpmSigmaT(o), tr;
        for (int i = 0; i < 3; i++) tr[i] = std::exp(-dt[i] * dist);
        return tr;
    }
    static float vpmPMedium(const OrcObject& o, float s) {
        V3 dt = vpmSigmaT(o), pm;
        for (int i = 0; i < 3; i++) pm[i] = dt[i] * std::exp(-dt[i] * s);
        return lumY(pm);
    }
    static float vpmPSurface(const OrcObject& o, float d) { return lumY(vpmTransmittance(o, d)); }
    // HGPhaseFunc.h:11-17 and HGPhaseSampler.h:63-69 (the same expression)
    static float hgValue(float g, float cosTheta) {
        float temp = 1 + g * g - 2 * g * cosTheta;
        return float(1 / (4 * REF_PI) * (1 - g * g) / (temp * std::sqrt(temp)));
    }
    // HGPhaseSampler.h:13-34 (right-handed frame variant, the one that returns)
    static V3 hgSample(float g, const LocalFrame& lf, Stream& rng) {
        float e1 = rng.genFloat(), e2 = rng.genFloat();
        float cosTheta;
        if (std::abs(g) < 1e-5f) {
            cosTheta = 1 - 2 * e1;
        } else {
            float sqrTerm = (1 - g * g) / (1 - g + 2 * g * e1);
            cosTheta = (1 + g * g - sqrTerm * sqrTerm) / (2 * g);
        }
        float sinTheta = std::sqrt(1 - cosTheta * cosTheta);
        float sinPhi = float(std::sin(2.0 * REF_PI * e2));
        float cosPhi = float(std::cos(2.0 * REF_PI * e2));
        return lf.toWorld(V3(sinTheta * cosPhi, cosTheta, sinTheta * sinPhi));
    }
    // the Fresnel split shared by "go in" and "go out" (SceneVPMObject.cpp:23-86, 151-213); returns
    // false on total internal reflection
    struct Fresnel {
        V3 reflDir, refrDir;
        float er, p;
        bool total;
    };
    Fresnel fresnel(const V3& normal, const V3& inDir, float current_n, float next_n) const {
        Fresnel f;
        f.reflDir = -dot(normal, inDir) * normal * 2 + inDir;
        normalize(f.reflDir);
        float theta = std::acos(dot(inDir, normal));
        float sin_phi = current_n / next_n * std::sin(theta);
        f.total = sin_phi >= 1;
        f.er = f.p = 0;
        if (f.total) return f;
        float phi = std::asin(sin_phi);
        if (theta > REF_PI / 2) phi = REF_PI - phi;
        V3 axis = cross(normal, inDir);
        normalize(axis);
        f.refrDir = rotate(axis, phi, normal);
        normalize(f.refrDir);
        float cos_theta = std::abs(std::cos(theta));
        float cos_phi = std::abs(std::cos(phi));
        float esr = powf(std::abs(current_n * cos_theta - next_n * cos_phi) /
                             (current_n * cos_theta + next_n * cos_phi), 2);
        float epr = powf(std::abs(next_n * cos_theta - current_n * cos_phi) /
                             (next_n * cos_theta + current_n * cos_phi), 2);
        f.er = (esr + epr) * 0.5f;
        f.p = clampf(f.er, 0.f, 1.f);
        return f;
    }
    PRay scatterVPM(int object, const PRay& in, Stream& rng) const {
        const OrcObject& o = S.objects[object];
        PRay out;
        // HomoMediaDistSampler.h:9-11 (drawn before the case split, SceneVPMObject.cpp:8-9)
        float sampDist = float(-std::log(1.0 - rng.genFloat()) / lumY(vpmSigmaT(o)));
        bool go_in_vol = (in.intersectObject == object) && (in.insideObject != object);
        bool be_in_vol = (in.insideObject == object);
        bool out_of_vol = (sampDist >= in.intersectDist);
        if (go_in_vol) {
            V3 position = in.origin + in.direction * in.intersectDist;
            LocalFrame lf;
            lf.buildFromNormal(objNormal(in.intersectObject, in.intersectObjectTriangleID, position));
            V3 normal = lf.n;
            out.origin = position;
            out.direction = in.direction;
            float current_n = in.insideObject >= 0 ? refrCoeff(in.insideObject) : 1;
            float next_n = refrCoeff(object);
            Fresnel f = fresnel(normal, in.direction, current_n, next_n);
            out.intersectObject = -1;
            out.color = V3(1, 1, 1);
            out.directionProb = 1;
            out.contactObject = object;
            out.contactObjectTriangleID = in.intersectObjectTriangleID;
            out.randomDir = false;  // DEFINITE in all three outcomes (SceneVPMObject.cpp:44, 73, 81)
            if (f.total) {
                out.direction = f.reflDir;
                out.insideObject = in.insideObject;
                out.color = out.color / cosineTerm(out);
            } else {
                out.direction = f.refrDir;
                if (rng.genFloat() < f.p) {
                    out.direction = f.reflDir;
                    out.color = out.color * (f.er / cosineTerm(out));
                    out.directionProb = f.p;
                    out.insideObject = in.insideObject;
                } else {
                    out.color = out.color * ((1 - f.er) / cosineTerm(out));
                    out.directionProb = 1 - f.p;
                    out.contactObject = out.insideObject = object;
                }
                normalize(out.direction);
            }
            return out;
        }
        if (be_in_vol && !out_of_vol) {
            LocalFrame lf;
            lf.buildFromNormal(in.direction);
            out.origin = in.origin + in.direction * sampDist;
            out.direction = hgSample(o.g, lf, rng);
            float ph = hgValue(o.g, dot(in.direction, out.direction));
            out.color = V3(ph, ph, ph);
            out.insideObject = object;
            out.contactObjectTriangleID = in.intersectObjectTriangleID;
            out.randomDir = true;  // RANDOM whether the path goes on or ends (SceneVPMObject.cpp:118, 132)
            V3 ds(o.sigma_s[0], o.sigma_s[1], o.sigma_s[2]);
            float albedo = lumY(ds) / lumY(vpmSigmaT(o));
            float rander = rng.genFloat();
            if (rander < albedo) {
                out.contactObject = -1;
                float oPdfW = hgValue(o.g, dot(lf.n, out.direction));
                out.directionProb = albedo * oPdfW;
                out.originProb = vpmPMedium(o, sampDist);
                out.color = out.color * ds;
            } else {
                out.direction = V3(0, 0, 0);
                out.color = V3(0, 0, 0);
                out.directionProb = 1;
                out.originProb = vpmPMedium(o, sampDist);
                out.insideObject = object;
                out.contactObject = -1;
            }
            return out;
        }
        if (be_in_vol && out_of_vol) {
            out = in;
            out.direction = in.direction;
            out.origin = in.origin + in.intersectDist * in.direction;
            out.contactObject = in.intersectObject;
            out.contactObjectTriangleID = in.intersectObjectTriangleID;
            out.insideObject = object;
            out.directionProb = 1;
            out.color = V3(1, 1, 1);
            bool going_out = (in.intersectObject == object);
            if (going_out) {
                LocalFrame lf;
                lf.buildFromNormal(objNormal(in.intersectObject, in.intersectObjectTriangleID, out.origin));
                V3 normal = lf.n;
                int outSide = findInsideObject(out, object);
                float current_n = refrCoeff(object);
                float next_n = outSide >= 0 ? refrCoeff(outSide) : 1;
                Fresnel f = fresnel(normal, in.direction, current_n, next_n);
                out.intersectObject = -1;
                out.originProb = vpmPSurface(o, in.intersectDist);
                out.randomDir = false;  // DEFINITE in all three outcomes (SceneVPMObject.cpp:171, 200, 209)
                if (f.total) {
                    out.direction = f.reflDir;
                    out.insideObject = in.insideObject;
                    out.contactObject = object;
                    out.color = out.color / cosineTerm(out);
                } else {
                    out.direction = f.refrDir;
                    if (rng.genFloat() < f.p) {
                        out.direction = f.reflDir;
                        out.color = out.color * (f.er / cosineTerm(out));
                        out.directionProb = f.p;
                        out.insideObject = in.insideObject;
                    } else {
                        out.color = out.color * ((1 - f.er) / cosineTerm(out));
                        out.directionProb = 1 - f.p;
                        out.insideObject = outSide;
                    }
                    normalize(out.direction);
                }
            } else {
                out.contactObject = -1;
                out.intersectDist = 0;
                if (in.intersectObject >= 0) out = scatter(in.intersectObject, out, rng);
                out.originProb *= vpmPSurface(o, in.intersectDist);
            }
            return out;
        }
        // (the reference falls off the end of the function here: undefined; the path is ended)
        out.direction = V3(0, 0, 0);
        out.color = V3(0, 0, 0);
        out.origin = in.origin;
        return out;
    }

    // SceneObject.cpp:366-373 (one material per object; the rand() it burns feeds nothing)
    PRay scatter(int object, const PRay& in, Stream& rng) const {
        switch (S.objects[object].type) {
            case ORC_DIFFUSE: return scatterDiffuse(object, in, rng);
            case ORC_GLOSSY: return scatterGlossy(object, in, rng);
            case ORC_REFRACTIVE: return scatterRefractive(object, in, rng);
            case ORC_EMISSIVE: return scatterEmissive(object, in);
            case ORC_VPM: return scatterVPM(object, in, rng);
            case ORC_PHONG: {
                // SceneObject.cpp:366-373: materialList[rand() % 2], list = {glossy, diffuse}
                // (ScenePhongObject.h:16-19); the choice costs one draw of the path's stream and its
                // probability is NOT divided out
                if (rng.genFloat() < 0.5f) return scatterGlossy(object, in, rng);
                return scatterDiffuse(object, in, rng);
            }
            default: {
                PRay dead;  // unsupported kind: terminate the path with zero weight
                dead.direction = V3(0, 0, 0);
                dead.color = V3(0, 0, 0);
                dead.origin = in.origin;
                return dead;
            }
        }
    }

    // Scene::intersect (Scene.cpp:341-352)
    float sceneIntersect(const PRay& r, const Condition* cond, int& objID, unsigned& localTri) {
        KRay k;
        k.origin = r.origin;
        k.direction = r.direction;
        unsigned tid = 0;
        nRays++;
        float dist = nodeIntersect(S, k, S.root, tid, cond, nullptr);
        if (dist >= 0) {
            objID = S.triObj[tid];
            localTri = (unsigned)S.triLocal[tid];
        }
        return dist;
    }

    // Camera.cpp:4-48
    PRay generateRay(unsigned pixelID, Stream& rng) {
        V3 front = V3(cam.focus[0], cam.focus[1], cam.focus[2]) - camPos;
        normalize(front);
        V3 up(cam.up[0], cam.up[1], cam.up[2]);
        V3 right = cross(front, up);
        normalize(right);
        V3 orthoUp = cross(right, front);
        PRay ray;
        int x = pixelID % cam.width;
        int y = pixelID / cam.width;
        float u = rng.genFloat();
        float v = rng.genFloat();
        ray.direction = front * float(cam.sight_dist) + right * float(x + u - (cam.width / 2.0)) +
                        orthoUp * float((cam.height / 2.0) - v - y);
        normalize(ray.direction);
        ray.color = V3(1, 1, 1);
        ray.origin = camPos;
        ray.contactObject = CAMERA_OBJ;
        int oid = -1;
        unsigned lt = 0;
        ray.intersectDist = sceneIntersect(ray, nullptr, oid, lt);
        ray.intersectObjectTriangleID = lt;
        ray.intersectObject = ray.intersectDist >= 0 ? oid : -1;
        ray.insideObject = findInsideObject(ray, -1);
        ray.directionProb = 1.f;
        return ray;
    }

    // Camera.cpp:60-78
    V3 eliminateVignetting(const V3& color, unsigned pixelID) const {
        V3 front = camFront;
        V3 up(cam.up[0], cam.up[1], cam.up[2]);
        V3 right = cross(front, up);
        normalize(right);
        V3 orthoUp = cross(right, front);
        int x = pixelID % cam.width;
        int y = pixelID / cam.width;
        V3 dir = front * float(cam.sight_dist) + right * float(x + 0.5 - (cam.width / 2.0)) +
                 orthoUp * float((cam.height / 2.0) - 0.5 - y);
        normalize(dir);
        return color / std::abs(dot(dir, front));
    }

    // MCRenderer.cpp:149-218
    void samplePath(std::vector<PRay>& path, PRay& prevRay, unsigned depth, Stream& rng) {
        PRay termRay;
        termRay.origin = prevRay.origin;
        termRay.direction = V3(0, 0, 0);
        termRay.color = V3(0, 0, 0);
        termRay.directionProb = 1;
        termRay.randomDir = false;  // MCRenderer.cpp:160
        termRay.insideObject = termRay.contactObject = termRay.intersectObject = -1;

        path.push_back(prevRay);
        PRay nextRay;
        if (prevRay.insideObject >= 0) {
            nextRay = scatter(prevRay.insideObject, prevRay, rng);
        } else if (prevRay.intersectObject >= 0) {
            nextRay = scatter(prevRay.intersectObject, prevRay, rng);
        } else {
            path.push_back(termRay);
            return;
        }
        if (length(nextRay.direction) < 0.5) {
            path.push_back(nextRay);
            return;
        }
        if (depth + 1 > maxDepth) {
            path.push_back(termRay);
            return;
        }
        Condition cond;
        cond.mode = 2;
        cond.contact = nextRay.contactObject;
        cond.inside = nextRay.insideObject;
        int oid = -1;
        unsigned lt = 0;
        float dist = sceneIntersect(nextRay, &cond, oid, lt);
        if (dist < 0) {
            path.push_back(nextRay);
            path.push_back(termRay);
            return;
        }
        nextRay.intersectObject = oid;
        nextRay.intersectObjectTriangleID = lt;
        nextRay.intersectDist = dist;
        samplePath(path, nextRay, depth + 1, rng);
    }

    // PathTracer.cpp:20-42 for one (pixel, sample); returns the vignetting-compensated radiance
    V3 samplePixel(unsigned p, unsigned s, uint32_t seed, std::vector<PRay>& eyePath) {
        Stream rng(seed, p, s);
        eyePath.clear();
        PRay start = generateRay(p, rng);
        samplePath(eyePath, start, 0, rng);
        if (!emissive(eyePath.back().contactObject)) return V3(0, 0, 0);
        V3 color(1, 1, 1);
        for (size_t i = 0; i < eyePath.size(); i++) {
            color = color * (eyePath[i].color / eyePath[i].directionProb / eyePath[i].originProb);
            if (i != eyePath.size() - 1) {
                color = color * cosineTerm(eyePath[i]);
                float dist = length(eyePath[i + 1].origin - eyePath[i].origin);
                color = color * radianceDecay(eyePath[i], dist);
            }
        }
        return eliminateVignetting(color, p);
    }

    // ---- PathTracerTest (PathTracerTest.cpp:7-135): path tracing + one light connection per
    // RANDOM vertex, combined with the power-1 balance heuristic. Restated for scenes without media.
    // Emission sampler: Scene::preprocessEmissionSampler (Scene.cpp:193-204),
    // SurfaceSampler::preprocess / genSample (Scene.cpp:19-33, 91-98), SceneObject::
    // preprocessEmissionSampler / emit (SceneObject.cpp:11-21, 137-218), SimpleShape::getTriangleArea /
    // genRandTrianglePosition (SimpleShape.cpp:14-42).
    struct Emitters {
        std::vector<int> objects;             // emissive objects in scene order
        std::vector<float> weightValues;      // running sum of the un-normalised weights (= areas)
        std::vector<std::vector<float>> areaValues;  // per object: running sum of triangle areas
        std::vector<float> totalArea, weight;  // per object; weight after normalizeEmissionWeight
        float totalWeight = 0;
    };
    float triangleArea(unsigned tid) const {
        V3 vps[3];
        for (int i = 0; i < 3; i++) vps[i] = S.verts[S.tris[tid].v[i]];
        V3 l0 = vps[1] - vps[0];
        V3 l1 = vps[2] - vps[1];
        return length(cross(l0, l1)) / 2;
    }
    Emitters preprocessEmissionSampler() const {
        Emitters E;
        for (int i = 0; i < (int)S.objects.size(); i++) {
            const OrcObject& o = S.objects[i];
            if (o.type != ORC_EMISSIVE) continue;
            E.objects.push_back(i);
            float totalArea = 0;
            std::vector<float> av;
            for (int k = 0; k < o.tri_count; k++) {
                totalArea += triangleArea(o.tri_begin + k);
                av.push_back(totalArea);
            }
            E.areaValues.push_back(av);
            E.totalArea.push_back(totalArea);
            E.weight.push_back(totalArea);
            E.totalWeight += totalArea;
            E.weightValues.push_back(E.totalWeight);
        }
        for (size_t i = 0; i < E.weight.size(); i++) E.weight[i] /= E.totalWeight;
        return E;
    }
    static uint32_t connectionKey(uint32_t key) { return lowbias32(key ^ 0x5bd1e995U); }

    // colorByConnectingLights (PathTracerTest.cpp:85-135) for vertex `ray` reached from `lastRay`;
    // six draws of the connection stream starting at 6 * vertexIndex (object, triangle, two for the
    // point, two for the emitted direction, which the connection overwrites)
    V3 colorByConnectingLights(const Emitters& E, const PRay& ray, const PRay& lastRay, Stream& crng) {
        if (E.objects.empty()) return V3(0, 0, 0);
        // SurfaceSampler::genSample
        float rnd = crng.genFloat() * E.totalWeight;
        size_t index = std::lower_bound(E.weightValues.begin(), E.weightValues.end(), rnd) - E.weightValues.begin();
        if (index >= E.weightValues.size()) index = E.weightValues.size() - 1;
        const int lobj = E.objects[index];
        // SceneObject::emit(isUniformOrigin = true, ...)
        PRay lightRay;
        const std::vector<float>& av = E.areaValues[index];
        rnd = crng.genFloat() * E.totalArea[index];
        size_t ti = std::lower_bound(av.begin(), av.end(), rnd) - av.begin();
        if (ti >= av.size()) ti = av.size() - 1;
        lightRay.contactObject = lobj;
        lightRay.contactObjectTriangleID = (unsigned)ti;
        {
            V3 vps[3];
            const unsigned tid = globalTid(lobj, (unsigned)ti);
            for (int i = 0; i < 3; i++) vps[i] = S.verts[S.tris[tid].v[i]];
            float u1 = crng.genFloat();
            float u2 = crng.genFloat();
            float su1 = sqrtf(u1);
            float u = 1.f - su1;
            float v = u2 * su1;
            lightRay.origin = vps[0] * u + vps[1] * v + vps[2] * (1.f - u - v);
        }
        lightRay.originProb = E.weight[index] / E.totalArea[index];
        crng.genFloat();  // the cosine-sampled emission direction (SceneObject.cpp:183) is replaced below;
        crng.genFloat();  // it was flipped to the normal's side, so getBSDF(ray, ray) returned the colour
        const OrcObject& lo = S.objects[lobj];
        lightRay.color = V3(lo.color[0], lo.color[1], lo.color[2]);

        lightRay.direction = ray.origin - lightRay.origin;
        float dist = length(lightRay.direction);
        float dist2 = dist * dist;
        normalize(lightRay.direction);
        PRay outRay = ray;
        outRay.direction = -lightRay.direction;

        // MCRenderer::testVisibility(outRay, lightRay), MCRenderer.cpp:115-117
        {
            Condition cond;
            cond.mode = 2;
            cond.contact = outRay.contactObject;
            cond.inside = outRay.insideObject;
            int oid = -1;
            unsigned lt = 0;
            float d = sceneIntersect(outRay, &cond, oid, lt);
            if (!(d < 0 || (d - length(outRay.origin - lightRay.origin) >= -1e-5f))) return V3(0, 0, 0);
        }
        V3 bsdfFactor = getBSDF(lastRay, outRay);
        if (lumY(bsdfFactor) < 1e-7f) return V3(0, 0, 0);
        float cosAtLight = std::min(std::max(0.f, std::abs(dot(contactNormal(lightRay), lightRay.direction))), 1.f);
        if (cosAtLight < 1e-6f) return V3(0, 0, 0);
        float cosToLight = 1;
        if (ray.contactObject != -1)
            cosToLight = std::min(std::max(0.f, std::abs(dot(contactNormal(ray), -lightRay.direction))), 1.f);
        float bsdfToLightPdf = directionSampleProbDensity(lastRay, outRay);
        float lightOriginPdf = lightRay.originProb;
        float dirPdfW = lightOriginPdf * dist2 / cosAtLight;
        float mis = dirPdfW / (dirPdfW + bsdfToLightPdf);
        V3 tmp = lightRay.color * cosAtLight * bsdfFactor * cosToLight / (lightRay.originProb * dist2);
        return tmp * mis;
    }
    // Ray::getBSDF (Ray.cpp:14-26) -> SceneObject::getBSDF (SceneObject.cpp:341-352) -> the material's:
    // DiffuseMaterial.cpp:94-101 / DiffuseBSDF.h:18-24, GlossyMaterial.cpp:69-79 / GlossyBSDF.h:24-33
    // (scaled by max(colour), as the reference does), Material.h:20 (zero) for the refractive ones,
    // SceneEmissiveObject.cpp:21-26
    V3 getBSDF(const PRay& inRay, const PRay& outRay) const {
        if (outRay.contactObject < 0) {
            // Ray.cpp:21-24: a vertex without a contact object asks its inside object. For a medium that is
            // SceneVPMObject::getBSDF (SceneVPMObject.cpp:239-246): ds * phase function of the two directions
            if (outRay.insideObject >= 0 && S.objects[outRay.insideObject].type == ORC_VPM) {
                const OrcObject& m = S.objects[outRay.insideObject];
                const float ph = hgValue(m.g, dot(inRay.direction, outRay.direction));
                return V3(m.sigma_s[0], m.sigma_s[1], m.sigma_s[2]) * ph;
            }
            return V3(0, 0, 0);  // (the camera is never a vertex >= 1)
        }
        const OrcObject& o = S.objects[outRay.contactObject];
        if (o.type == ORC_EMISSIVE)
            return dot(contactNormal(outRay), outRay.direction) < 0 ? V3(0, 0, 0) : V3(o.color[0], o.color[1], o.color[2]);
        // the materials' textures at the vertex itself (DiffuseMaterial.cpp:99, GlossyMaterial.cpp:74,78)
        const V3 color = diffuseColor(outRay.contactObject, outRay.contactObjectTriangleID, outRay.origin);
        const float gcoeff = glossyCoeff(outRay.contactObject, outRay.contactObjectTriangleID, outRay.origin);
        LocalFrame lf;
        lf.buildFromNormal(objNormal(outRay.contactObject, outRay.contactObjectTriangleID, outRay.origin));
        if (o.type == ORC_DIFFUSE) {
            if (dot(lf.n, inRay.direction) <= 0 && dot(lf.n, outRay.direction) >= 0) return color / float(REF_PI);
            return V3(0, 0, 0);
        }
        auto glossy = [&](const V3& gc) {
            float p = maxComp(gc);
            V3 c(0, 0, 0);
            if (dot(lf.n, inRay.direction) <= 0 && dot(lf.n, outRay.direction) >= 0) {
                V3 r2 = -dot(lf.n, inRay.direction) * lf.n * 2 + inRay.direction;
                if (dot(r2, outRay.direction) > 0) c = gc * powf(dot(r2, outRay.direction), gcoeff);
            }
            return p * c;
        };
        const V3 gcolor = glossyColor(outRay.contactObject, outRay.contactObjectTriangleID, outRay.origin);
        if (o.type == ORC_GLOSSY) return glossy(gcolor);
        if (o.type == ORC_PHONG) {  // SceneObject.cpp:341-352: the materials' BSDFs add (glossy first)
            V3 bsdfColor(0, 0, 0);
            bsdfColor = bsdfColor + glossy(gcolor);
            V3 dcol(0, 0, 0);
            if (dot(lf.n, inRay.direction) <= 0 && dot(lf.n, outRay.direction) >= 0) dcol = color / float(REF_PI);
            bsdfColor = bsdfColor + dcol;
            return bsdfColor;
        }
        return V3(0, 0, 0);
    }
    // Ray::getDirectionSampleProbDensity (Ray.cpp:28-40) -> SceneObject.cpp:227-248 -> the material's:
    // DiffuseMaterial.cpp:78-86, GlossyMaterial.cpp:58-67, Material.h:18 (1) for refractive objects;
    // an emissive object has no material (0)
    float directionSampleProbDensity(const PRay& inRay, const PRay& outRay) const {
        if (outRay.contactObject < 0) {
            // Ray.cpp:35-38 -> SceneVPMObject::getDirectionSampleProbDensity (SceneVPMObject.cpp:257-267)
            if (outRay.insideObject >= 0 && S.objects[outRay.insideObject].type == ORC_VPM) {
                if (!outRay.randomDir) return 0;
                const OrcObject& m = S.objects[outRay.insideObject];
                LocalFrame lf;
                lf.buildFromNormal(inRay.direction);
                const float albedo = lumY(V3(m.sigma_s[0], m.sigma_s[1], m.sigma_s[2])) / lumY(vpmSigmaT(m));
                return albedo * hgValue(m.g, dot(lf.n, outRay.direction));
            }
            return 0;
        }
        const OrcObject& o = S.objects[outRay.contactObject];
        // DiffuseMaterial.cpp:83, GlossyMaterial.cpp:66: textures at the vertex
        const V3 dcolor = diffuseColor(outRay.contactObject, outRay.contactObjectTriangleID, outRay.origin);
        const float gcoeff = glossyCoeff(outRay.contactObject, outRay.contactObjectTriangleID, outRay.origin);
        if (o.type == ORC_DIFFUSE) {
            LocalFrame lf;
            lf.buildFromNormal(objNormal(outRay.contactObject, outRay.contactObjectTriangleID, outRay.origin));
            float p = maxComp(dcolor);
            return p * cosPdf(lf, outRay.direction, 1.f);
        }
        if (o.type == ORC_GLOSSY) {
            LocalFrame lf;
            V3 normal = contactNormal(outRay);
            V3 reflDir = -dot(normal, inRay.direction) * normal * 2 + inRay.direction;
            lf.buildFromNormal(reflDir);
            return cosPdf(lf, outRay.direction, gcoeff);
        }
        if (o.type == ORC_REFRACTIVE) return 1;
        if (o.type == ORC_PHONG) {  // SceneObject.cpp:239-245: the mean of the materials' densities
            float prob = 0;
            {
                LocalFrame lf;
                V3 normal = contactNormal(outRay);
                V3 reflDir = -dot(normal, inRay.direction) * normal * 2 + inRay.direction;
                lf.buildFromNormal(reflDir);
                prob += cosPdf(lf, outRay.direction, gcoeff);
            }
            {
                LocalFrame lf;
                lf.buildFromNormal(objNormal(outRay.contactObject, outRay.contactObjectTriangleID, outRay.origin));
                float p = maxComp(dcolor);
                prob += p * cosPdf(lf, outRay.direction, 1.f);
            }
            return prob / 2;
        }
        return 0;
    }

    // PathTracerTest.cpp:20-78 for one (pixel, sample)
    V3 samplePixelTest(const Emitters& E, unsigned p, unsigned s, uint32_t seed, std::vector<PRay>& eyePath,
                       uint64_t* shadowRays) {
        Stream rng(seed, p, s);
        eyePath.clear();
        PRay start = generateRay(p, rng);
        samplePath(eyePath, start, 0, rng);
        V3 throughput(1, 1, 1), color(0, 0, 0);
        bool lastSpecular = true;
        float lastPdfW = 1.f;
        for (size_t i = 1; i < eyePath.size(); i++) {
            const PRay& v = eyePath[i];
            if (emissive(v.contactObject)) {
                const OrcObject& eo = S.objects[v.contactObject];
                V3 contrib(eo.color[0], eo.color[1], eo.color[2]);
                size_t ei = std::find(E.objects.begin(), E.objects.end(), v.contactObject) - E.objects.begin();
                float dirPdfA = E.weight[ei] / E.totalArea[ei];
                float mis = 1.f;
                if (i > 1 && !lastSpecular) {
                    float cosine = dot(contactNormal(v), -eyePath[i - 1].direction);
                    float dist = length(v.origin - eyePath[i - 1].origin);
                    float dirPdfW = dirPdfA * dist * dist / std::abs(cosine);
                    mis = lastPdfW / (lastPdfW + dirPdfW);
                }
                color = color + throughput * contrib * mis;
                break;
            }
            if (v.randomDir) {
                Stream crng(seed, p, s);
                crng.key = connectionKey(crng.key);
                crng.ctr = 6 * (uint32_t)i;
                if (shadowRays) (*shadowRays)++;
                color = color + throughput * colorByConnectingLights(E, v, eyePath[i - 1], crng);
            }
            if (i + 1 >= eyePath.size() || (v.origin.x == eyePath[i + 1].origin.x && v.origin.y == eyePath[i + 1].origin.y &&
                                            v.origin.z == eyePath[i + 1].origin.z))
                break;
            lastSpecular = !v.randomDir;
            lastPdfW = v.directionProb;
            throughput = throughput * (v.color * cosineTerm(v) / v.directionProb);
            float dist = std::max(length(v.origin - eyePath[i - 1].origin), 1e-5f);
            throughput = throughput * radianceDecay(eyePath[i - 1], dist);
        }
        return color;
    }
};

}  // namespace

// ---- C ABI ----------------------------------------------------------------------------------------
extern "C" {

OrcScene* orc_scene_create(int32_t n_verts, const float* verts, int32_t n_tris, const uint32_t* tris,
                           const int32_t* tri_obj, const int32_t* tri_local, const float* tri_vn,
                           int32_t n_objects, const OrcObject* objects) {
    OrcScene* S = new OrcScene;
    S->verts.resize(n_verts);
    for (int i = 0; i < n_verts; i++) S->verts[i] = V3(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);
    S->tris.resize(n_tris);
    for (int i = 0; i < n_tris; i++)
        for (int k = 0; k < 3; k++) S->tris[i].v[k] = tris[3 * i + k];
    S->triObj.assign(tri_obj, tri_obj + n_tris);
    S->triLocal.assign(tri_local, tri_local + n_tris);
    if (tri_vn) {
        S->triVn.resize(3 * (size_t)n_tris);
        for (size_t i = 0; i < 3 * (size_t)n_tris; i++)
            S->triVn[i] = V3(tri_vn[3 * i], tri_vn[3 * i + 1], tri_vn[3 * i + 2]);
    }
    S->objects.assign(objects, objects + n_objects);
    // KDTree::build(BEST), KDTree.cpp:230-246
    S->root = new Node;
    for (int i = 0; i < n_tris; i++) S->root->triangleIndices.push_back(i);
    int md = 0;
    splitBest(*S, S->root, 0, md);
    S->depth = md;
    if (n_tris > 0) flattenNode(*S, S->root);
    // Scene::buildObjKDTrees (Scene.cpp:287-320): one tree per object over that object's own
    // triangles; only the volumetric ones are ever queried (Scene.cpp:369)
    S->objRoots.assign(n_objects, nullptr);
    for (int oi = 0; oi < n_objects; oi++) {
        if (objects[oi].type != ORC_VPM) continue;
        Node* r = new Node;
        for (int t = 0; t < objects[oi].tri_count; t++) r->triangleIndices.push_back(objects[oi].tri_begin + t);
        int d2 = 0;
        splitBest(*S, r, 0, d2);
        S->objRoots[oi] = r;
    }
    return S;
}

void orc_scene_set_textures(OrcScene* S, const float* tri_vt, int32_t n_textures, const OrcTexture* textures,
                            const OrcObjectTex* object_tex) {
    S->triVt.clear();
    if (tri_vt) {
        S->triVt.resize(3 * S->tris.size());
        for (size_t i = 0; i < S->triVt.size(); i++) S->triVt[i] = V3(tri_vt[3 * i], tri_vt[3 * i + 1], tri_vt[3 * i + 2]);
    }
    S->textures.assign(n_textures, OrcScene::Tex());
    for (int k = 0; k < n_textures; k++) {
        OrcScene::Tex& T = S->textures[k];
        T.width = textures[k].width, T.height = textures[k].height, T.channels = textures[k].channels;
        T.texels.assign(textures[k].texels, textures[k].texels + (size_t)T.width * T.height * T.channels);
        memcpy(T.coordT, textures[k].coord_transform, sizeof T.coordT);
        memcpy(T.colorT, textures[k].color_transform, sizeof T.colorT);
        invertMat(T.coordT, T.coordInv);
    }
    S->objectTex.clear();
    if (object_tex && n_textures > 0) S->objectTex.assign(object_tex, object_tex + S->objects.size());
}
void orc_texture_color(const OrcTexture* tex, const float coord[3], float out[3]) {
    OrcScene::Tex T;
    T.width = tex->width, T.height = tex->height, T.channels = tex->channels;
    T.texels.assign(tex->texels, tex->texels + (size_t)T.width * T.height * T.channels);
    memcpy(T.coordT, tex->coord_transform, sizeof T.coordT);
    memcpy(T.colorT, tex->color_transform, sizeof T.colorT);
    const V3 c = textureColor(T, V3(coord[0], coord[1], coord[2]));
    out[0] = c.x, out[1] = c.y, out[2] = c.z;
}
// the matrix arithmetic of the texture path (nvMatrix.h:233-369, 415-424), for the known-answer tests
void orc_mat_inverse(const float m[16], float out[16]) { invertMat(m, out); }
void orc_mat_apply(const float m[16], const float v[3], float w, float out[3]) {
    const V3 r = applyMat(m, V3(v[0], v[1], v[2]), w);
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
void orc_tex_coord(const OrcScene* s, int32_t tid, const float pos[3], float out[3]) {
    const V3 c = texCoord(*s, (unsigned)tid, V3(pos[0], pos[1], pos[2]));
    out[0] = c.x, out[1] = c.y, out[2] = c.z;
}

// KDTree::build(Strategy), KDTree.cpp:230-246, on the scene's main tree: 0 = LOOP, 1 = BEST (KDTree.h:24)
void orc_scene_rebuild(OrcScene* S, int32_t strategy) {
    destroyNode(S->root);
    S->root = new Node;
    for (size_t i = 0; i < S->tris.size(); i++) S->root->triangleIndices.push_back((unsigned)i);
    int md = 0;
    if (strategy == 0)
        splitLoop(*S, S->root, 0, 0, md);
    else
        splitBest(*S, S->root, 0, md);
    S->depth = md;
    S->fmin.clear(), S->fmax.clear(), S->fright.clear(), S->fbegin.clear(), S->fend.clear(), S->slotTid.clear();
    if (!S->tris.empty()) flattenNode(*S, S->root);
}

void orc_scene_destroy(OrcScene* s) {
    if (!s) return;
    destroyNode(s->root);
    for (Node* r : s->objRoots) destroyNode(r);
    delete s;
}

int32_t orc_tree_num_nodes(const OrcScene* s) { return (int32_t)s->fright.size(); }
int32_t orc_tree_depth(const OrcScene* s) { return s->depth; }

void orc_tree_flatten(const OrcScene* s, float* node_min, float* node_max, int32_t* node_right,
                      int32_t* leaf_begin, int32_t* leaf_end, int32_t* slot_tid) {
    size_t n = s->fright.size();
    for (size_t i = 0; i < n; i++) {
        for (int c = 0; c < 3; c++) {
            node_min[3 * i + c] = s->fmin[i][c];
            node_max[3 * i + c] = s->fmax[i][c];
        }
        node_right[i] = s->fright[i];
        leaf_begin[i] = s->fbegin[i];
        leaf_end[i] = s->fend[i];
    }
    std::copy(s->slotTid.begin(), s->slotTid.end(), slot_tid);
}

static inline KRay loadRay(const float* o, const float* d, size_t n, size_t i) {
    KRay r;
    r.origin = V3(o[i], o[n + i], o[2 * n + i]);
    r.direction = V3(d[i], d[n + i], d[2 * n + i]);
    return r;
}

void orc_intersect(const OrcScene* s, const float* origins, const float* directions,
                   const int32_t* aux, size_t n, int mode, float* t_out, int32_t* tid_out,
                   int threads) {
    (void)threads;
#pragma omp parallel for schedule(dynamic, 256) num_threads(threads > 0 ? threads : 1)
    for (long long i = 0; i < (long long)n; i++) {
        KRay r = loadRay(origins, directions, n, (size_t)i);
        Condition c = makeCond(mode, aux ? aux[i] : 0);
        unsigned tid = 0;
        float t = s->root ? nodeIntersect(*s, r, s->root, tid, mode ? &c : nullptr, nullptr) : -1;
        t_out[i] = t >= 0 ? t : -1.f;
        tid_out[i] = t >= 0 ? (int32_t)tid : -1;
    }
}

void orc_visibility(const OrcScene* s, const float* origins, const float* directions,
                    const int32_t* aux, const float* max_dist, size_t n, int mode,
                    uint8_t* visible, int threads) {
    (void)threads;
#pragma omp parallel for schedule(dynamic, 256) num_threads(threads > 0 ? threads : 1)
    for (long long i = 0; i < (long long)n; i++) {
        KRay r = loadRay(origins, directions, n, (size_t)i);
        Condition c = makeCond(mode, aux ? aux[i] : 0);
        unsigned tid = 0;
        float dist = s->root ? nodeIntersect(*s, r, s->root, tid, mode ? &c : nullptr, nullptr) : -1;
        // MCRenderer.cpp:117: srInfo.dist<0 || (srInfo.dist - length >= -EPSILON)
        visible[i] = (dist < 0 || (dist - max_dist[i] >= -1e-5f)) ? 1 : 0;
    }
}

void orc_intersect_fast(const OrcScene* s, int32_t n_nodes, const float* node_min, const float* node_max,
                        const int32_t* node_right, const int32_t* leaf_begin, const int32_t* leaf_end,
                        const float* origins, const float* directions, const int32_t* aux, size_t n, int mode,
                        float* t_out, int32_t* tid_out, uint64_t* n_box, uint64_t* n_tri) {
    GivenTree T{n_nodes, node_min, node_max, node_right, leaf_begin, leaf_end};
    const int N = (int)s->fright.size();
    std::vector<int> parent(N, -1), slotNode(s->slotTid.size(), 0);
    for (int i = 0; i < N; i++)
        if (s->fright[i] >= 0) {
            parent[i + 1] = i;
            parent[s->fright[i]] = i;
        } else {
            for (int k = s->fbegin[i]; k < s->fend[i]; k++) slotNode[k] = i;
        }
    Counters cnt;
    for (size_t i = 0; i < n; i++) {
        KRay r = loadRay(origins, directions, n, i);
        Condition c = makeCond(mode, aux ? aux[i] : 0);
        int slot = -1;
        float t = (s->root && n_nodes > 0) ? fastIntersect(*s, T, parent, slotNode, r, c, slot, cnt) : -1;
        if (t_out) t_out[i] = t;
        if (tid_out) tid_out[i] = slot >= 0 ? s->slotTid[slot] : -1;
    }
    if (n_box) *n_box = cnt.box;
    if (n_tri) *n_tri = cnt.tri;
}

void orc_intersect_ref_counts(const OrcScene* s, const float* origins, const float* directions,
                              const int32_t* aux, size_t n, int mode, uint64_t* n_box,
                              uint64_t* n_tri) {
    Counters cnt;
    for (size_t i = 0; i < n; i++) {
        KRay r = loadRay(origins, directions, n, i);
        Condition c = makeCond(mode, aux ? aux[i] : 0);
        unsigned tid = 0;
        if (s->root) nodeIntersect(*s, r, s->root, tid, mode ? &c : nullptr, &cnt);
    }
    *n_box = cnt.box;
    *n_tri = cnt.tri;
}

void orc_intersect_ordered(const OrcScene* s, const float* origins, const float* directions,
                           const int32_t* aux, size_t n, int mode, float* t_out, int32_t* tid_out,
                           uint64_t* n_box, uint64_t* n_tri) {
    Counters cnt;
    for (size_t i = 0; i < n; i++) {
        KRay r = loadRay(origins, directions, n, i);
        Condition c = makeCond(mode, aux ? aux[i] : 0);
        int slot = -1;
        float t = s->root ? orderedIntersect(*s, r, c, slot, cnt) : -1;
        if (t_out) t_out[i] = slot >= 0 ? t : -1.f;
        if (tid_out) tid_out[i] = slot >= 0 ? s->slotTid[slot] : -1;
    }
    if (n_box) *n_box = cnt.box;
    if (n_tri) *n_tri = cnt.tri;
}

float orc_box_intersect(const float bmin[3], const float bmax[3], const float o[3], const float d[3]) {
    KRay r;
    r.origin = V3(o[0], o[1], o[2]);
    r.direction = V3(d[0], d[1], d[2]);
    return boxIntersect(V3(bmin[0], bmin[1], bmin[2]), V3(bmax[0], bmax[1], bmax[2]), r);
}

float orc_tri_intersect(const float v0[3], const float v1[3], const float v2[3], const float o[3],
                        const float d[3]) {
    OrcScene S;
    S.verts = {V3(v0[0], v0[1], v0[2]), V3(v1[0], v1[1], v1[2]), V3(v2[0], v2[1], v2[2])};
    Tri t;
    t.v[0] = 0;
    t.v[1] = 1;
    t.v[2] = 2;
    S.tris.push_back(t);
    KRay r;
    r.origin = V3(o[0], o[1], o[2]);
    r.direction = V3(d[0], d[1], d[2]);
    return triIntersect(S, r, 0, nullptr);
}

void orc_world_normal(const OrcScene* s, int32_t tid, const float pos[3], float n_out[3]) {
    V3 n = worldNormal(*s, (unsigned)tid, V3(pos[0], pos[1], pos[2]));
    n_out[0] = n.x;
    n_out[1] = n.y;
    n_out[2] = n.z;
}

int orc_legal(const OrcScene* s, int32_t tid, const float o[3], const float d[3], float dist,
              int32_t contact, int32_t inside) {
    Condition c;
    c.mode = 2;
    c.contact = contact;
    c.inside = inside;
    KRay r;
    r.origin = V3(o[0], o[1], o[2]);
    r.direction = V3(d[0], d[1], d[2]);
    return c.legal(*s, r, (unsigned)tid, dist) ? 1 : 0;
}

uint32_t orc_rng_draw(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t dim) {
    Stream st(seed, pixel, sample);
    st.ctr = dim;
    return st.bits();
}

void orc_camera_ray(const OrcCamera* cam, uint32_t pixel, float u, float v, float dir_out[3]) {
    V3 pos(cam->position[0], cam->position[1], cam->position[2]);
    V3 front = V3(cam->focus[0], cam->focus[1], cam->focus[2]) - pos;
    normalize(front);
    V3 up(cam->up[0], cam->up[1], cam->up[2]);
    V3 right = cross(front, up);
    normalize(right);
    V3 orthoUp = cross(right, front);
    int x = pixel % cam->width;
    int y = pixel / cam->width;
    V3 dir = front * float(cam->sight_dist) + right * float(x + u - (cam->width / 2.0)) +
             orthoUp * float((cam->height / 2.0) - v - y);
    normalize(dir);
    dir_out[0] = dir.x;
    dir_out[1] = dir.y;
    dir_out[2] = dir.z;
}

static void put(const V3& v, float out[3]) { out[0] = v.x, out[1] = v.y, out[2] = v.z; }
static V3 get(const float v[3]) { return V3(v[0], v[1], v[2]); }

void orc_local_frame(const float n[3], float s_out[3], float t_out[3]) {
    LocalFrame lf;
    lf.buildFromNormal(get(n));
    put(lf.s, s_out);
    put(lf.t, t_out);
}
void orc_to_world(const float n[3], const float a[3], float out[3]) {
    LocalFrame lf;
    lf.buildFromNormal(get(n));
    put(lf.toWorld(get(a)), out);
}
float orc_hg_eval(float g, const float in[3], const float outd[3]) {
    return Tracer::hgValue(g, dot(get(in), get(outd)));
}
void orc_face_normal(const float p0[3], const float p1[3], const float p2[3], float out[3]) {
    V3 vps[3] = {get(p0), get(p1), get(p2)};
    put(faceNormalOf(vps), out);
}
void orc_interp_normal(const float p0[3], const float p1[3], const float p2[3], const float n0[3],
                       const float n1[3], const float n2[3], const float position[3], float out[3]) {
    V3 vps[3] = {get(p0), get(p1), get(p2)};
    V3 vns[3] = {get(n0), get(n1), get(n2)};
    put(interpNormal(vps, vns, get(position)), out);
}
void orc_rotate(const float axis[3], float angle, const float v[3], float out[3]) {
    put(Tracer::rotate(get(axis), angle, get(v)), out);
}
void orc_normalize(const float v[3], float out[3]) {
    V3 r = get(v);
    normalize(r);
    put(r, out);
}
void orc_div(const float v[3], float d, float out[3]) { put(get(v) / d, out); }
float orc_dot(const float a[3], const float b[3]) { return dot(get(a), get(b)); }
float orc_length(const float a[3]) { return length(get(a)); }

void orc_hemi_cos(float expTerm, float u1, float u2, float out[3]) {
    V3 r = Tracer::hemiCos(expTerm, u1, u2);
    out[0] = r.x;
    out[1] = r.y;
    out[2] = r.z;
}

void orc_render_pt(const OrcScene* s, const OrcCamera* cam, uint32_t spp_begin, uint32_t spp_count,
                   uint32_t max_depth, uint32_t seed, float* rgb_sum, uint64_t* n_rays, int threads) {
    int npix = cam->width * cam->height;
    std::fill(rgb_sum, rgb_sum + 3 * (size_t)npix, 0.f);
    uint64_t total = 0;
#pragma omp parallel num_threads(threads > 0 ? threads : 1) reduction(+ : total)
    {
        Tracer tr(*s, *cam, max_depth);
        std::vector<PRay> path;
#pragma omp for schedule(dynamic, 64)
        for (int p = 0; p < npix; p++) {
            float acc[3] = {0, 0, 0};
            for (uint32_t k = 0; k < spp_count; k++) {
                V3 c = tr.samplePixel((unsigned)p, spp_begin + k, seed, path);
                acc[0] += c.x;
                acc[1] += c.y;
                acc[2] += c.z;
            }
            rgb_sum[3 * p] = acc[0];
            rgb_sum[3 * p + 1] = acc[1];
            rgb_sum[3 * p + 2] = acc[2];
        }
        total += tr.nRays;
    }
    if (n_rays) *n_rays = total;
}

int orc_render_pttest(const OrcScene* s, const OrcCamera* cam, uint32_t spp_begin, uint32_t spp_count,
                      uint32_t max_depth, uint32_t seed, float* rgb_sum, uint64_t* n_rays,
                      uint64_t* n_shadow_rays, int threads) {
    // (scenes with a medium included: PathTracerTest.cpp:70-74 never divides by originProb, so inside a medium its
    // estimate is not the path tracer's; it is restated as the reference computes it)
    int npix = cam->width * cam->height;
    std::fill(rgb_sum, rgb_sum + 3 * (size_t)npix, 0.f);
    uint64_t total = 0, shadows = 0;
#pragma omp parallel num_threads(threads > 0 ? threads : 1) reduction(+ : total, shadows)
    {
        Tracer tr(*s, *cam, max_depth);
        const Tracer::Emitters E = tr.preprocessEmissionSampler();
        std::vector<PRay> path;
#pragma omp for schedule(dynamic, 64)
        for (int p = 0; p < npix; p++) {
            float acc[3] = {0, 0, 0};
            for (uint32_t k = 0; k < spp_count; k++) {
                V3 c = tr.samplePixelTest(E, (unsigned)p, spp_begin + k, seed, path, &shadows);
                acc[0] += c.x;
                acc[1] += c.y;
                acc[2] += c.z;
            }
            rgb_sum[3 * p] = acc[0];
            rgb_sum[3 * p + 1] = acc[1];
            rgb_sum[3 * p + 2] = acc[2];
        }
        total += tr.nRays;
    }
    if (n_rays) *n_rays = total - shadows;
    if (n_shadow_rays) *n_shadow_rays = shadows;
    return 0;
}

int32_t orc_trace_path(const OrcScene* s, const OrcCamera* cam, uint32_t pixel, uint32_t sample,
                       uint32_t max_depth, uint32_t seed, float* verts, int32_t cap, float rgb[3]) {
    Tracer tr(*s, *cam, max_depth);
    std::vector<PRay> path;
    V3 c = tr.samplePixel(pixel, sample, seed, path);
    rgb[0] = c.x;
    rgb[1] = c.y;
    rgb[2] = c.z;
    int32_t n = (int32_t)path.size();
    for (int32_t i = 0; i < n && i < cap; i++) {
        float* o = verts + 10 * i;
        o[0] = path[i].origin.x; o[1] = path[i].origin.y; o[2] = path[i].origin.z;
        o[3] = path[i].direction.x; o[4] = path[i].direction.y; o[5] = path[i].direction.z;
        o[6] = path[i].color.x; o[7] = path[i].color.y; o[8] = path[i].color.z;
        o[9] = path[i].directionProb;
    }
    return n;
}

int32_t orc_find_inside(const OrcScene* s, const float o[3], const float d[3], int32_t current) {
    OrcCamera cam;
    memset(&cam, 0, sizeof cam);
    cam.focus[2] = -1;
    Tracer tr(*s, cam, 0);
    PRay r;
    r.origin = V3(o[0], o[1], o[2]);
    r.direction = V3(d[0], d[1], d[2]);
    return tr.findInsideObject(r, current);
}

}  // extern "C"
