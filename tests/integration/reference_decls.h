// Minimal LOCAL declarations of the reference's types, just enough to syntax-check the binding printed in
// INTEGRATION.md section 1 against include/mirt.h (tests/test_integration_snippet.py). Declarations only, written for
// this check: names, members and signatures follow the reference headers cited per block; nothing here is compiled
// into the product. The snippet's own #include lines are stripped by the test; this header stands in for them.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <stdint.h>
using namespace std;

// nvVector.h / nvMatrix.h: component access by name and by index, the constructors the snippet uses
template <class T> struct vec3 {
    T x, y, z;
    vec3() {}
    vec3(T x, T y, T z);
    template <class V4> explicit vec3(const V4& v);
    T& operator[](int i);
    const T& operator[](int i) const;
};
template <class T> struct vec4 {
    T x, y, z, w;
    vec4() {}
    vec4(T x, T y, T z, T w);
    vec4(const vec3<T>& v, T w);
};
typedef vec3<float> vec3f;
typedef vec3<unsigned> vec3ui;
typedef vec4<float> vec4f;
template <class T> struct matrix4 {};
template <class T> matrix4<T> inverse(const matrix4<T>& m);
template <class T> matrix4<T> transpose(const matrix4<T>& m);
template <class T> vec4<T> operator*(const matrix4<T>& m, const vec4<T>& v);

// KDTree.h:10-72
class KDTree {
public:
    struct Ray { vec3f origin, direction; };
    struct Triangle { vec3ui vertexIndices; void* sourceInformation; };
    struct Node;
    Node* root;
    vector<vec3f> vertexPositionList;
    vector<Triangle> triangleList;
};
struct KDTree::Node {
    Node* left;
    Node* right;
    vector<unsigned> triangleIndices;
    struct BoundingBox { vec3f minCoord, maxCoord; } boundingBox;
};

// Texture.h:37
class Texture { public: vec3f getColor(const vec3f& coord) const; };

// SimpleShape.h:18-61, SceneObject.h:21-82 (the members the snippet reads)
class SimpleShape {
public:
    string name;
    vector<vec3f> vertexNormalList;
    vector<vec3ui> faceVertexNormalIndexList;
    matrix4<float> getTransform() const;
    unsigned getTriangleNum() const;
    virtual ~SimpleShape() {}
};
class SceneObject : public SimpleShape {
public:
    int objectIndex;
    virtual bool isVolumetric();
};
// DiffuseMaterial.h:12, GlossyMaterial.h:14-15, RefractiveMaterial.h:6-13 (+ the two getters INTEGRATION.md asks for)
class DiffuseMaterial { public: Texture tex; };
class GlossyMaterial { public: Texture colorTex, coeffTex; };
class RefractiveMaterial { public: float getRefrCoeff() const; vec3f getSurfColor() const; vec3f getDecayColor() const; };
// Scene*Object.h (+ getMaterial() on the refractive / reflective ones, as INTEGRATION.md asks)
class SceneDiffuseObject : public SceneObject { public: DiffuseMaterial* getMaterial(); };
class SceneGlossyObject : public SceneObject { public: GlossyMaterial* getMaterial(); };
class ScenePhongObject : public SceneObject { public: GlossyMaterial* getGlossyMaterial(); DiffuseMaterial* getDiffuseMaterial(); };
class SceneRefractiveObject : public SceneObject { public: RefractiveMaterial* getMaterial(); };
class SceneReflectiveObject : public SceneObject { public: RefractiveMaterial* getMaterial(); };
class SceneEmissiveObject : public SceneObject { public: vec3f getColor() const; };
class SceneVPMObject : public SceneObject { public: vec3f ds, da, dt; float g; float IOR; };

// Scene.h:10-107
class Scene {
public:
    struct ObjSourceInformation { unsigned objID; unsigned triangleID; };
    KDTree tree;
    vector<unsigned> objTriangleOffsetMap;
    vector<SceneObject*> objects;
};

// IntersectionGPU.h:12-46: the public interface exactly as the reference declares it, + the two static members
class IntersectionGPU {
public:
    struct Ray {
        vec3f origin;
        vec3f direction;
        int last_tid;
        Ray() {}
        Ray(const vec3f& origin, const vec3f& direction);
    };
    static void loadKDTreeToShader(const KDTree& tree);
    static void init();
    static vector<vec4f> query(const vector<Ray>& rayList);
    static const Scene* scene;  // added
    static int mode;            // added
};
