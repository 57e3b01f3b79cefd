// vecmath.h — small fp32 vector / 4x4 matrix types for the host surface.
// Semantics follow the reference's nv::vec3f / nv::matrix4<float> where they are observable:
// dividing by zero and normalising a zero vector are no-ops (nvVector.h:376-380, 522-526), dot
// products accumulate x, y, z in that order (nvVector.h:501-507), M*v sums columns 0..3 in order
// (nvMatrix.h:415-424). Storage is row-major m[r][c] (the reference stores column-major; only
// element(r,c) is observable).
#pragma once
#include <cmath>

namespace mirt_host {

struct vec3f {
    float x, y, z;
    vec3f() : x(0), y(0), z(0) {}
    vec3f(float a, float b, float c) : x(a), y(b), z(c) {}
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
    vec3f operator+(const vec3f& o) const { return vec3f(x + o.x, y + o.y, z + o.z); }
    vec3f operator-(const vec3f& o) const { return vec3f(x - o.x, y - o.y, z - o.z); }
    vec3f operator-() const { return vec3f(-x, -y, -z); }
    vec3f operator*(float s) const { return vec3f(x * s, y * s, z * s); }
    vec3f operator/(float d) const { return d == 0 ? *this : vec3f(x / d, y / d, z / d); }
    float dot(const vec3f& o) const {
        float r = 0;
        r += x * o.x;
        r += y * o.y;
        r += z * o.z;
        return r;
    }
    vec3f cross(const vec3f& r) const {
        return vec3f(y * r.z - z * r.y, z * r.x - x * r.z, x * r.y - y * r.x);
    }
    float length() const { return std::sqrt(dot(*this)); }
    void normalize() {
        float r = length();
        if (r > 0) *this = *this / r;
    }
};

struct vec3ui {
    unsigned x, y, z;
    unsigned& operator[](int i) { return (&x)[i]; }
    const unsigned& operator[](int i) const { return (&x)[i]; }
};

struct matrix4 {
    float m[4][4];
    matrix4() { make_identity(); }
    void make_identity() {
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) m[r][c] = r == c ? 1.f : 0.f;
    }
    float& operator()(int r, int c) { return m[r][c]; }
    float operator()(int r, int c) const { return m[r][c]; }
    // M * (v, w)
    vec3f apply(const vec3f& v, float w) const {
        vec3f r;
        for (int i = 0; i < 3; i++) r[i] = v.x * m[i][0] + v.y * m[i][1] + v.z * m[i][2] + w * m[i][3];
        return r;
    }
    matrix4 transposed() const {
        matrix4 t;
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) t.m[r][c] = m[c][r];
        return t;
    }
    matrix4 operator*(const matrix4& b) const {
        matrix4 r;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                float s = 0;
                for (int c = 0; c < 4; c++) s += m[i][c] * b.m[c][j];
                r.m[i][j] = s;
            }
        return r;
    }
    // Gauss-Jordan with scaled partial pivoting, fp32 (same scheme as nvMatrix.h:233-369).
    // A singular matrix returns identity.
    matrix4 inverse() const {
        float s[4][8];
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                s[i][j] = m[i][j];
                s[i][j + 4] = i == j ? 1.f : 0.f;
            }
        float scp[4];
        for (int i = 0; i < 4; i++) {
            scp[i] = std::fabs(s[i][0]);
            for (int j = 1; j < 4; j++)
                if (std::fabs(s[i][j]) > scp[i]) scp[i] = std::fabs(s[i][j]);
            if (scp[i] == 0.f) return matrix4();
        }
        int row[4] = {0, 1, 2, 3};
        for (int i = 0; i < 4; i++) {
            int pivot = i;
            float best = std::fabs(s[row[i]][i] / scp[row[i]]);
            for (int p = i + 1; p < 4; p++) {
                float v = std::fabs(s[row[p]][i] / scp[row[p]]);
                if (v > best) {
                    best = v;
                    pivot = p;
                }
            }
            if (best == 0.f) return matrix4();
            if (pivot != i) {
                int t = row[i];
                row[i] = row[pivot];
                row[pivot] = t;
            }
            for (int p = i + 1; p < 4; p++) {
                float f = s[row[p]][i] / s[row[i]][i];
                s[row[p]][i] = 0.f;
                for (int j = i + 1; j < 8; j++) s[row[p]][j] -= f * s[row[i]][j];
            }
        }
        if (s[row[3]][3] == 0.f) return matrix4();
        for (int i = 3; i > 0; i--)
            for (int j = i - 1; j > -1; j--) {
                float f = s[row[j]][i] / s[row[i]][i];
                for (int jj = j + 1; jj < 8; jj++) s[row[j]][jj] -= f * s[row[i]][jj];
            }
        matrix4 inv;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) inv.m[i][j] = s[row[i]][j + 4] / s[row[i]][i];
        return inv;
    }
};

}  // namespace mirt_host
