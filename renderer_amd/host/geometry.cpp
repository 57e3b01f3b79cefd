// geometry.cpp — SimpleShape (.obj loading, unitize, world normals), SceneObject, KDTree build.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <stdexcept>

#include "host.h"

namespace mirt_host {

// ---- SimpleShape ------------------------------------------------------------------------------------
void SimpleShape::getBoundingBox(vec3f& lo, vec3f& hi) const {
    if (vertexList.empty()) return;
    lo = hi = vertexList[0];
    for (size_t i = 1; i < vertexList.size(); i++)
        for (int c = 0; c < 3; c++) {
            float v = vertexList[i][c];
            lo[c] = v < lo[c] ? v : lo[c];
            hi[c] = v > hi[c] ? v : hi[c];
        }
}

// SimpleShape.cpp:257-272: recentre on the box centre, scale the longest extent to [-1, 1].
// Returns the matrix the reference returns. Its 16-argument constructor fills COLUMN-major memory
// (nvMatrix.h:98-106), so the "translation" arguments land in the bottom row: applied to a point the
// matrix only scales by 2/maxLen. <GroupObject> sub-shapes are unitized through this matrix
// (SimpleShape.cpp:186-194) and therefore are scaled but not recentred; kept as it is.
matrix4 SimpleShape::unitize() {
    matrix4 u;
    if (vertexList.empty()) return u;
    vec3f lo, hi;
    getBoundingBox(lo, hi);
    vec3f center = (lo + hi) / 2;
    vec3f delta = hi - lo;
    float maxLen = delta.x > delta.y ? delta.x : delta.y;
    maxLen = maxLen > delta.z ? maxLen : delta.z;
    for (auto& v : vertexList) {
        v = v - center;
        v = v / (maxLen / 2);
    }
    u(0, 0) = u(1, 1) = u(2, 2) = 2 / maxLen;
    u(3, 0) = -2 / maxLen * center.x;
    u(3, 1) = -2 / maxLen * center.y;
    u(3, 2) = -2 / maxLen * center.z;
    return u;
}

// SimpleShape.cpp:222-236
void SimpleShape::applyTransform() {
    matrix4 normalMat = transform.inverse().transposed();
    for (auto& v : vertexList) v = transform.apply(v, 1.f);
    for (auto& n : vertexNormalList) n = normalMat.apply(n, 0.f);
    transform.make_identity();
}

// Face token forms a, a/b, a/b/c, a//c with 1-based indices (SimpleShape.cpp:116-139). Returns the
// number of integers the "%d/%d/%d" pattern matched and, when that is 1, retries "%d//%d".
static void parseFaceToken(const char* tok, unsigned& v, unsigned& t, unsigned& n, int& got3, int& got2) {
    int a = 0, b = 0, c = 0;
    got3 = sscanf(tok, "%d/%d/%d", &a, &b, &c);
    got2 = 0;
    if (got3 >= 1) v = (unsigned)a;
    if (got3 >= 2) t = (unsigned)b;
    if (got3 >= 3) n = (unsigned)c;
    if (got3 == 1) {
        int a2 = 0, c2 = 0;
        got2 = sscanf(tok, "%d//%d", &a2, &c2);
        if (got2 >= 1) v = (unsigned)a2;
        if (got2 == 2) n = (unsigned)c2;
    }
}

// SimpleShape.cpp:44-220 for ".obj". Rules kept: '#' lines are skipped; a
// line counts only if it has at least 4 whitespace-separated tokens and only the first 4 are read
// (faces must be triangles; "vt" needs 3 numbers); no material handling.
void SimpleShape::loadShape(const std::string& fileName, bool normalize, std::vector<SimpleShape*>* splitShapes) {
    const bool split = splitShapes != nullptr;
    SimpleShape* shape = nullptr;
    unsigned current_fi = 0, current_fni = 0, current_fti = 0;
    vertexList.clear();
    faceVertexIndexList.clear();
    vertexNormalList.clear();
    vertexTexCoordList.clear();
    faceVertexNormalIndexList.clear();
    faceVertexTexCoordIndexList.clear();
    const std::string suffix = fileName.size() >= 4 ? fileName.substr(fileName.size() - 4) : std::string();
    if (suffix == ".shp") {
        // SimpleShape.cpp:52-64, the binary dump SimpleShape::saveShape writes (:341-357): an int count of FLOATS and
        // the vertex coordinates, an int count of INDICES and the face indices. The reference resizes both lists to
        // those counts, i.e. to three times the number of vertices / faces, and reads into their heads: the tails stay
        // zero - extra vertices at the origin and extra faces (0, 0, 0). Kept as it is (they are degenerate triangles
        // on vertex 0 that the tree and the triangle test handle like any other).
        FILE* f = fopen(fileName.c_str(), "rb");
        if (!f) throw std::runtime_error("loadShape: cannot open " + fileName);
        int size = 0;
        bool ok = fread(&size, sizeof(int), 1, f) == 1 && size >= 0 && size < (1 << 28);
        if (ok) {
            vertexList.assign((size_t)size, vec3f(0, 0, 0));
            ok = size == 0 || fread(reinterpret_cast<float*>(vertexList.data()), sizeof(float), (size_t)size, f) == (size_t)size;
        }
        if (ok) ok = fread(&size, sizeof(int), 1, f) == 1 && size >= 0 && size < (1 << 28);
        if (ok) {
            faceVertexIndexList.assign((size_t)size, vec3ui{0, 0, 0});
            ok = size == 0 || fread(reinterpret_cast<unsigned*>(faceVertexIndexList.data()), sizeof(unsigned), (size_t)size, f) == (size_t)size;
        }
        fclose(f);
        if (!ok) throw std::runtime_error("loadShape: truncated or malformed .shp file " + fileName);
        for (const vec3ui& fi : faceVertexIndexList)
            for (int k = 0; k < 3; k++)
                if (fi[k] >= vertexList.size()) throw std::runtime_error("loadShape: face index out of range in " + fileName);
        if (split) throw std::runtime_error("loadShape: a .shp file has no groups to split: " + fileName);
        // the common tail below (SimpleShape.cpp:183-196: unitize when asked, bounding box) is shared with .obj
    } else {
    if (suffix != ".obj")
        throw std::runtime_error("loadShape: only .obj and .shp meshes are supported: " + fileName);
    FILE* file = fopen(fileName.c_str(), "r");
    if (!file) throw std::runtime_error("loadShape: cannot open " + fileName);
    char line[1024], attrib[1024], parms[3][1024];
    while (fgets(line, sizeof line, file)) {
        if (line[0] == '#') continue;
        int num = sscanf(line, "%1023s %1023s %1023s %1023s", attrib, parms[0], parms[1], parms[2]);
        if (num >= 2 && strcmp("g", attrib) == 0 && split) {  // SimpleShape.cpp:85-96
            if (shape) splitShapes->push_back(shape);
            shape = new SimpleShape;
            shape->name = parms[0];
            current_fi = (unsigned)vertexList.size();
            current_fni = (unsigned)vertexNormalList.size();
            current_fti = (unsigned)vertexTexCoordList.size();
        }
        if (num != 4) continue;
        if (strcmp("v", attrib) == 0 || strcmp("vn", attrib) == 0 || strcmp("vt", attrib) == 0) {
            vec3f v;
            sscanf(parms[0], "%f", &v.x);
            sscanf(parms[1], "%f", &v.y);
            sscanf(parms[2], "%f", &v.z);
            if (attrib[1] == 0) {
                vertexList.push_back(v);
                if (shape) shape->vertexList.push_back(v);
            } else if (attrib[1] == 'n') {
                vertexNormalList.push_back(v);
                if (shape) shape->vertexNormalList.push_back(v);
            } else {
                vertexTexCoordList.push_back(v);
                if (shape) shape->vertexTexCoordList.push_back(v);
            }
        } else if (strcmp("f", attrib) == 0) {
            vec3ui tri{0, 0, 0}, vn{0, 0, 0}, vt{0, 0, 0};
            int got3 = 0, got2 = 0;
            for (int k = 0; k < 3; k++) parseFaceToken(parms[k], tri[k], vt[k], vn[k], got3, got2);
            // the form of the LAST token decides what the face carries (SimpleShape.cpp:118-139)
            bool has_n = got3 == 3 || (got3 == 1 && got2 == 2);
            bool has_t = got3 >= 2;
            for (int k = 0; k < 3; k++) {
                tri[k] -= 1;
                vn[k] -= 1;
                vt[k] -= 1;
            }
            faceVertexIndexList.push_back(tri);
            if (has_n) faceVertexNormalIndexList.push_back(vn);
            if (has_t) faceVertexTexCoordIndexList.push_back(vt);
            if (shape) {
                shape->faceVertexIndexList.push_back(vec3ui{tri.x - current_fi, tri.y - current_fi, tri.z - current_fi});
                if (has_n)
                    shape->faceVertexNormalIndexList.push_back(vec3ui{vn.x - current_fni, vn.y - current_fni, vn.z - current_fni});
                if (has_t)
                    shape->faceVertexTexCoordIndexList.push_back(vec3ui{vt.x - current_fti, vt.y - current_fti, vt.z - current_fti});
            }
        }
    }
    fclose(file);
    if (shape) splitShapes->push_back(shape);
    auto dropSplit = [&]() {
        if (!split) return;
        for (SimpleShape* p : *splitShapes) delete p;
        splitShapes->clear();
    };
    if (split)  // a face that refers to a vertex of an earlier group cannot be rebased: refuse it
        for (SimpleShape* p : *splitShapes) {
            bool bad = false;
            for (auto& f : p->faceVertexIndexList)
                for (int k = 0; k < 3; k++) bad |= f[k] >= p->vertexList.size();
            for (auto& f : p->faceVertexNormalIndexList)
                for (int k = 0; k < 3; k++) bad |= f[k] >= p->vertexNormalList.size();
            if (bad) {
                std::string nm = p->name;
                dropSplit();
                throw std::runtime_error("loadShape: group '" + nm + "' in " + fileName +
                                         " uses vertices declared before its g line");
            }
        }
    for (auto& f : faceVertexIndexList)
        for (int k = 0; k < 3; k++)
            if (f[k] >= vertexList.size())
            {
                dropSplit();
                throw std::runtime_error("loadShape: face index out of range in " + fileName);
            }
    for (auto& f : faceVertexNormalIndexList)
        for (int k = 0; k < 3; k++)
            if (f[k] >= vertexNormalList.size())
            {
                dropSplit();
                throw std::runtime_error("loadShape: normal index out of range in " + fileName);
            }
    }  // .obj
    if (normalize) {
        const matrix4 unitizeMat = unitize();
        if (split)  // SimpleShape.cpp:186-194
            for (SimpleShape* p : *splitShapes) {
                p->setTransform(unitizeMat);
                p->applyTransform();
                p->setTransform(transform);
                p->getBoundingBox(p->minCoord, p->maxCoord);
            }
    }
    getBoundingBox(minCoord, maxCoord);
}

bool SimpleShape::hasVertexNormals() const {
    return !vertexNormalList.empty() && faceVertexNormalIndexList.size() == faceVertexIndexList.size();
}

bool SimpleShape::getWorldCornerNormals(unsigned fi, vec3f out[3]) const {
    if (!hasVertexNormals()) return false;
    matrix4 normalMat = transform.inverse().transposed();
    for (int i = 0; i < 3; i++) out[i] = normalMat.apply(vertexNormalList[faceVertexNormalIndexList[fi][i]], 0.f);
    return true;
}

// ---- SceneObject ------------------------------------------------------------------------------------
SceneObject::SceneObject(Scene* s) : scene(s), objectIndex(-1) {
    memset(&material, 0, sizeof material);
    material.type = MIRT_OBJ_DIFFUSE;
    material.ior = 1.f;
    for (int c = 0; c < 3; c++) material.color[c] = material.decay[c] = 1.f;
}

float SceneObject::getRefrCoeff() const {
    if ((material.type == MIRT_OBJ_REFRACTIVE && material.ior >= 0) || material.type == MIRT_OBJ_VPM)
        return material.ior;
    return 1.f;  // SceneObject.h:58
}

// ---- KDTree -----------------------------------------------------------------------------------------
void KDTree::destroy() {
    triangleList.clear();
    vertexPositionList.clear();
    nodeMin.clear();
    nodeMax.clear();
    nodeRight.clear();
    nodeLeafBegin.clear();
    nodeLeafEnd.clear();
    slotTid.clear();
    depth = 0;
}

vec3f KDTree::getDiagonal() const {
    if (nodeMin.empty()) return vec3f();
    return vec3f(nodeMax[0] - nodeMin[0], nodeMax[1] - nodeMin[1], nodeMax[2] - nodeMin[2]);
}

// One node of the median-split tree (KDTree::splitBest, KDTree.cpp:129-228), emitted in pre-order.
// `order[begin,end)` holds the node's triangles in the reference's order (a stable partition keeps
// it), so a leaf's slot range is simply [begin,end).
// loopDim >= 0: KDTree::splitLoop (KDTree.cpp:54-127) instead: the split axis is given (it cycles with the depth), no
// choice among the axes; everything else (median of the distinct vertices' coordinates, "any vertex <= split goes
// left", a split that separates nothing makes a leaf) is the same rule.
void KDTree::buildNode(std::vector<unsigned>& order, size_t begin, size_t end, int level,
                       std::vector<unsigned>& stamp, unsigned& stampGen, std::vector<unsigned>& uniq,
                       std::vector<float>& coords, int loopDim) {
    const size_t me = nodeRight.size();
    nodeRight.push_back(-1);
    nodeLeafBegin.push_back(-1);
    nodeLeafEnd.push_back(-1);
    if (level > depth) depth = level;
    // tight bounds of the node's triangles (KDTree.cpp:139-153) + its distinct vertex ids
    vec3f lo = vertexPositionList[triangleList[order[begin]].vertexIndices[0]], hi = lo;
    stampGen++;
    uniq.clear();
    for (size_t i = begin; i < end; i++)
        for (int k = 0; k < 3; k++) {
            unsigned vi = triangleList[order[i]].vertexIndices[k];
            const vec3f& p = vertexPositionList[vi];
            for (int c = 0; c < 3; c++) {
                lo[c] = p[c] < lo[c] ? p[c] : lo[c];
                hi[c] = p[c] > hi[c] ? p[c] : hi[c];
            }
            if (stamp[vi] != stampGen) {
                stamp[vi] = stampGen;
                uniq.push_back(vi);
            }
        }
    for (int c = 0; c < 3; c++) {
        nodeMin.push_back(lo[c]);
        nodeMax.push_back(hi[c]);
    }
    auto makeLeaf = [&]() {
        nodeLeafBegin[me] = (int32_t)begin;
        nodeLeafEnd[me] = (int32_t)end;
    };
    const size_t count = end - begin;
    if (count <= 1) {  // maxLeafTriangleNum = 1 (KDTree.cpp:24)
        makeLeaf();
        return;
    }
    // per axis: split at the median of the distinct vertices' coordinates; a triangle goes left if
    // ANY vertex is <= split; keep the axis whose left fraction is closest to 1/2 (first wins ties)
    float bestRatio = 10;
    int bestDim = 0;
    float bestSplit = 0;
    for (int dim = 0; dim < 3; dim++) {
        if (loopDim >= 0 && dim != loopDim) continue;
        coords.resize(uniq.size());
        for (size_t i = 0; i < uniq.size(); i++) coords[i] = vertexPositionList[uniq[i]][dim];
        size_t mid = coords.size() >> 1;
        std::nth_element(coords.begin(), coords.begin() + mid, coords.end());
        float split = coords[mid];
        size_t left = 0;
        for (size_t i = begin; i < end; i++) {
            const vec3ui& v = triangleList[order[i]].vertexIndices;
            left += (vertexPositionList[v.x][dim] <= split) | (vertexPositionList[v.y][dim] <= split) |
                    (vertexPositionList[v.z][dim] <= split);
        }
        float ratio = (unsigned)left / float(count);
        // the reference compares in double: abs(ratio - 0.5) with a double literal (KDTree.cpp:191)
        if (loopDim >= 0 || std::fabs((double)ratio - 0.5) < std::fabs((double)bestRatio - 0.5)) {
            bestRatio = ratio;
            bestDim = dim;
            bestSplit = split;
        }
    }
    if (bestRatio == 0 || bestRatio == 1) {  // KDTree.cpp:199-203: no useful split, multi-triangle leaf
        makeLeaf();
        return;
    }
    auto goesLeft = [&](unsigned t) {
        const vec3ui& v = triangleList[t].vertexIndices;
        return vertexPositionList[v.x][bestDim] <= bestSplit || vertexPositionList[v.y][bestDim] <= bestSplit ||
               vertexPositionList[v.z][bestDim] <= bestSplit;
    };
    size_t mid = std::stable_partition(order.begin() + begin, order.begin() + end, goesLeft) - order.begin();
    const int nextDim = loopDim >= 0 ? (loopDim + 1) % 3 : -1;  // KDTree.cpp:125-126
    buildNode(order, begin, mid, level + 1, stamp, stampGen, uniq, coords, nextDim);
    nodeRight[me] = (int32_t)nodeRight.size();
    buildNode(order, mid, end, level + 1, stamp, stampGen, uniq, coords, nextDim);
}

void KDTree::build(Strategy strategy) {
    nodeMin.clear();
    nodeMax.clear();
    nodeRight.clear();
    nodeLeafBegin.clear();
    nodeLeafEnd.clear();
    slotTid.clear();
    depth = 0;
    if (triangleList.empty()) return;
    for (auto& t : triangleList)
        for (int k = 0; k < 3; k++)
            if (t.vertexIndices[k] >= vertexPositionList.size())
                throw std::runtime_error("KDTree::build: vertex index out of range");
    // The build orders vertex coordinates (std::nth_element / sort in the reference, KDTree.cpp:54-127, 160-196): with a
    // NaN among them the ordering is not one, and what the reference builds depends on its library. Refused.
    for (auto& t : triangleList)
        for (int k = 0; k < 3; k++) {
            const vec3f& p = vertexPositionList[t.vertexIndices[k]];
            if (p.x != p.x || p.y != p.y || p.z != p.z)
                throw std::runtime_error("KDTree::build: a triangle corner has a NaN coordinate");
        }
    std::vector<unsigned> order(triangleList.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = (unsigned)i;
    std::vector<unsigned> stamp(vertexPositionList.size(), 0), uniq;
    std::vector<float> coords;
    unsigned gen = 0;
    buildNode(order, 0, order.size(), 0, stamp, gen, uniq, coords, strategy == LOOP ? 0 : -1);  // KDTree.cpp:237-244
    slotTid.assign(order.begin(), order.end());
}

}  // namespace mirt_host
