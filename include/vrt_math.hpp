// vrt_math.hpp -- the handful of fp32 vector/matrix routines the camera path
// needs, written so that every result has the same bits as glm 1.0.0 (the
// library the reference uses: glm::lookAt / perspective / inverse / normalize /
// cross / radians, src/main.cpp:808-813, include/Camera.hpp:44-47,86-97).
// Column-major storage, m[c][r]; compile with -ffp-contract=off.
#ifndef VRT_MATH_HPP
#define VRT_MATH_HPP
#include <cmath>

namespace vrtm {

struct vec3 {
    float x, y, z;
    vec3() : x(0.0f), y(0.0f), z(0.0f) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    vec3 &operator+=(const vec3 &o) { x += o.x; y += o.y; z += o.z; return *this; }
    vec3 &operator-=(const vec3 &o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
};
inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
inline vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }

struct vec4 {
    float x, y, z, w;
    vec4() : x(0.0f), y(0.0f), z(0.0f), w(0.0f) {}
    vec4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
    vec4(vec3 v, float w_) : x(v.x), y(v.y), z(v.z), w(w_) {}
    float &operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};

struct mat4 {
    vec4 col[4];
    mat4() {}
    explicit mat4(float d) { col[0].x = d; col[1].y = d; col[2].z = d; col[3].w = d; }
    vec4 &operator[](int c) { return col[c]; }
    const vec4 &operator[](int c) const { return col[c]; }
    const float *data() const { return &col[0].x; }
};

inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }
inline float dot(vec3 a, vec3 b) { float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z; return tx + ty + tz; }
inline vec3 cross(vec3 a, vec3 b) { return vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
inline vec3 normalize(vec3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }

// glm::lookAt (right-handed)
inline mat4 lookAt(vec3 eye, vec3 center, vec3 up) {
    const vec3 f = normalize(center - eye);
    const vec3 s = normalize(cross(f, up));
    const vec3 u = cross(s, f);
    mat4 r(1.0f);
    r[0][0] = s.x; r[1][0] = s.y; r[2][0] = s.z;
    r[0][1] = u.x; r[1][1] = u.y; r[2][1] = u.z;
    r[0][2] = -f.x; r[1][2] = -f.y; r[2][2] = -f.z;
    r[3][0] = -dot(s, eye);
    r[3][1] = -dot(u, eye);
    r[3][2] = dot(f, eye);
    return r;
}

// glm::perspective (right-handed, depth -1..1)
inline mat4 perspective(float fovy, float aspect, float zNear, float zFar) {
    const float t = std::tan(fovy / 2.0f);
    mat4 r(0.0f);
    r[0][0] = 1.0f / (aspect * t);
    r[1][1] = 1.0f / t;
    r[2][2] = -(zFar + zNear) / (zFar - zNear);
    r[2][3] = -1.0f;
    r[3][2] = -(2.0f * zFar * zNear) / (zFar - zNear);
    return r;
}

// glm::inverse(mat4): cofactor expansion in glm's grouping
inline mat4 inverse(const mat4 &m) {
    const float c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3], c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    const float c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3], c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    const float c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3], c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    const float c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2], c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    const float c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2], c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    const float c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3], c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    const float c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2], c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    const float c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2], c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    const float c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1], c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    const float F0[4] = {c00, c00, c02, c03}, F1[4] = {c04, c04, c06, c07}, F2[4] = {c08, c08, c10, c11};
    const float F3[4] = {c12, c12, c14, c15}, F4[4] = {c16, c16, c18, c19}, F5[4] = {c20, c20, c22, c23};
    const float V0[4] = {m[1][0], m[0][0], m[0][0], m[0][0]}, V1[4] = {m[1][1], m[0][1], m[0][1], m[0][1]};
    const float V2[4] = {m[1][2], m[0][2], m[0][2], m[0][2]}, V3[4] = {m[1][3], m[0][3], m[0][3], m[0][3]};
    const float SA[4] = {+1.0f, -1.0f, +1.0f, -1.0f}, SB[4] = {-1.0f, +1.0f, -1.0f, +1.0f};
    mat4 inv;
    for (int i = 0; i < 4; ++i) {
        inv[0][i] = ((V1[i] * F0[i] - V2[i] * F1[i]) + V3[i] * F2[i]) * SA[i];
        inv[1][i] = ((V0[i] * F0[i] - V2[i] * F3[i]) + V3[i] * F4[i]) * SB[i];
        inv[2][i] = ((V0[i] * F1[i] - V1[i] * F3[i]) + V3[i] * F5[i]) * SA[i];
        inv[3][i] = ((V0[i] * F2[i] - V1[i] * F4[i]) + V2[i] * F5[i]) * SB[i];
    }
    const float d0 = m[0][0] * inv[0][0], d1 = m[0][1] * inv[1][0], d2 = m[0][2] * inv[2][0], d3 = m[0][3] * inv[3][0];
    const float ood = 1.0f / ((d0 + d1) + (d2 + d3));
    mat4 r;
    for (int c = 0; c < 4; ++c)
        for (int k = 0; k < 4; ++k) r[c][k] = inv[c][k] * ood;
    return r;
}

}  // namespace vrtm
#endif
