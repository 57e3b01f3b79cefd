// ref_math.cpp — driver over the REFERENCE's own header-only math, compiled in place from
// /root/reference (never copied): nvVector.h, nvMatrix.h, LocalFrame.h, BSDF.h, HGPhaseFunc.h.
// These are the only reference sources that build in this image without stand-in headers (every
// .cpp of the reference starts with the MSVC precompiled header). Built by `make ref` into
// oracle/_ref/libref_math.so; used by scripts/gen_ref_math_golden.py to write tests/golden/ref_math.npz
// and, when present, by tests/test_ref_math.py for a live comparison. TEST INFRASTRUCTURE ONLY.
//
// Each entry point evaluates the expression the named reference call site evaluates, with the
// reference's own types and operators; only the marshalling is ours.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include <algorithm>

// Name lookup as at the reference's own include sites: HGPhaseFunc.h and rotMat are always seen
// after smallFuncs.h:9 `using namespace std;` (SceneVPMObject.h:2-6 via RandGenerator.h:6), so an
// unqualified sqrt / cos / sin of a float picks the float overload there (and on MSVC in any case).
using namespace std;

#include "LocalFrame.h"
#include "nvMatrix.h"
#undef M_PI
#include "HGPhaseFunc.h"

static matrix4<float> load(const float m[16]) {  // m[r*4+c] = element(r,c)
    matrix4<float> M;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) M.element(r, c) = m[r * 4 + c];
    return M;
}

extern "C" {

// SimpleShape.h:46  vec3f(transform*vec4<float>(vertexList[vi], 1))
void ref_world_vertex(const float m[16], const float v[3], float out[3]) {
    vec3f r = vec3f(load(m) * vec4<float>(vec3f(v[0], v[1], v[2]), 1));
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}

// SimpleShape.cpp:310,322  normalMat = transpose(inverse(transform)); vec3f(normalMat * vec4<float>(vn, 0))
void ref_normal_transform(const float m[16], const float vn[3], float out[3]) {
    matrix4<float> normalMat = transpose(inverse(load(m)));
    vec3f r = vec3f(normalMat * vec4<float>(vec3f(vn[0], vn[1], vn[2]), 0));
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}

// nvMatrix.h inverse(), all 16 elements, element(r,c) order
void ref_inverse(const float m[16], float out[16]) {
    matrix4<float> I = inverse(load(m));
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) out[r * 4 + c] = I.element(r, c);
}

// LocalFrame.h:14-23
void ref_local_frame(const float n[3], float s_out[3], float t_out[3]) {
    LocalFrame lf;
    lf.buildFromNormal(vec3f(n[0], n[1], n[2]));
    s_out[0] = lf.s.x, s_out[1] = lf.s.y, s_out[2] = lf.s.z;
    t_out[0] = lf.t.x, t_out[1] = lf.t.y, t_out[2] = lf.t.z;
}

// LocalFrame.h:25-28 after buildFromNormal(n)
void ref_to_world(const float n[3], const float a[3], float out[3]) {
    LocalFrame lf;
    lf.buildFromNormal(vec3f(n[0], n[1], n[2]));
    vec3f r = lf.toWorld(vec3f(a[0], a[1], a[2]));
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}

// HGPhaseFunc.h:11-17
float ref_hg_eval(float g, const float in[3], const float outd[3]) {
    HGPhaseFunc f(g);
    LocalFrame lf;
    return f.evaluate(lf, vec3f(in[0], in[1], in[2]), vec3f(outd[0], outd[1], outd[2])).x;
}

// SimpleShape.cpp:311-314  faceNormal = (vps[1]-vps[0]).cross(vps[2]-vps[0]); normalize()
void ref_face_normal(const float p0[3], const float p1[3], const float p2[3], float out[3]) {
    vec3f a(p0[0], p0[1], p0[2]), b(p1[0], p1[1], p1[2]), c(p2[0], p2[1], p2[2]);
    vec3f b1 = b - a;
    vec3f b2 = c - a;
    vec3f faceNormal = b1.cross(b2);
    faceNormal.normalize();
    out[0] = faceNormal.x, out[1] = faceNormal.y, out[2] = faceNormal.z;
}

// SimpleShape.cpp:329-337 (the statements, re-typed over the reference's vector operators):
// barycentric interpolation of already-transformed corner normals at `position`
void ref_interp_normal(const float p0[3], const float p1[3], const float p2[3], const float n0[3],
                       const float n1[3], const float n2[3], const float position[3], float out[3]) {
    vec3f vps[3] = {vec3f(p0[0], p0[1], p0[2]), vec3f(p1[0], p1[1], p1[2]), vec3f(p2[0], p2[1], p2[2])};
    vec3f vns[3] = {vec3f(n0[0], n0[1], n0[2]), vec3f(n1[0], n1[1], n1[2]), vec3f(n2[0], n2[1], n2[2])};
    vec3f b1 = vps[1] - vps[0];
    vec3f b2 = vps[2] - vps[0];
    vec3f v = vec3f(position[0], position[1], position[2]) - vps[0];
    float d12 = b1.dot(b2);
    float l1 = b1.length();
    float l2 = b2.length();
    float u2 = (v.dot(b1) * d12 - v.dot(b2) * l1 * l1) / (d12 * d12 - l1 * l1 * l2 * l2);
    float u1 = (v.dot(b1) - u2 * d12) / (l1 * l1);
    vec3f normal = (1 - u1 - u2) * vns[0] + u1 * vns[1] + u2 * vns[2];
    normal.normalize();
    out[0] = normal.x, out[1] = normal.y, out[2] = normal.z;
}

// smallFuncs.h:11-24 rotMat(axis, angle) * vec4(v, 0), the way RefractiveMaterial.cpp / SceneVPMObject.cpp
// use it. smallFuncs.h itself does not compile here (its isIllegal() calls MSVC's _isnan), so the
// entries are re-typed; what this pins is the reference's 16-argument matrix4 constructor
// (column-major fill, nvMatrix.h:98-106), transpose() and matrix * vec4 (nvMatrix.h:415-424).
void ref_rotate(const float axis[3], float angle, const float v[3], float out[3]) {
    float c = cos(angle);
    float s = sin(angle);
    float _c = 1 - c;
    float x = axis[0], y = axis[1], z = axis[2];
    matrix4<float> R = transpose(matrix4<float>(c + _c * x * x, _c * x * y - s * z, _c * x * z + s * y, 0,
                                                _c * x * y + s * z, c + _c * y * y, _c * y * z - s * x, 0,
                                                _c * x * z - s * y, _c * y * z + s * x, c + _c * z * z, 0,
                                                0, 0, 0, 1));
    vec3f r = vec3f(R * vec4<float>(vec3f(v[0], v[1], v[2]), 0));
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}

// nvVector.h: normalize() (zero vectors stay), operator/ (division by zero is a no-op)
void ref_normalize(const float v[3], float out[3]) {
    vec3f r(v[0], v[1], v[2]);
    r.normalize();
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
void ref_div(const float v[3], float d, float out[3]) {
    vec3f r(v[0], v[1], v[2]);
    r /= d;
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
float ref_dot(const float a[3], const float b[3]) { return vec3f(a[0], a[1], a[2]).dot(vec3f(b[0], b[1], b[2])); }
float ref_length(const float a[3]) { return vec3f(a[0], a[1], a[2]).length(); }

}  // extern "C"
