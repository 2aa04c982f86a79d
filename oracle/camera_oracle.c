/*
 * camera_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Restates include/Camera.hpp and the four glm 1.0.0 routines the reference's
 * frame loop uses (src/main.cpp:808-813), in glm's exact fp32 operation order.
 * Pinned by tests/golden/camera.json (generated from the reference's own
 * Camera.hpp + vendored glm by oracle/_ref/ref_camera).
 */
#include "oracle.h"
#include <math.h>
#include <string.h>

/* glm/trigonometric: radians(deg) = deg * 0.01745329251994329576923690768489 */
float o_radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

static float dot3(const float *a, const float *b) { /* glm compute_dot<vec3>: (x + y) + z */
    float tx = a[0] * b[0], ty = a[1] * b[1], tz = a[2] * b[2];
    return tx + ty + tz;
}
static void normalize3(const float *v, float *o) { /* v * (1/sqrt(dot)) */
    float inv = 1.0f / sqrtf(dot3(v, v));
    o[0] = v[0] * inv; o[1] = v[1] * inv; o[2] = v[2] * inv;
}
static void cross3(const float *x, const float *y, float *o) { /* glm/detail/func_geometric.inl compute_cross */
    float r0 = x[1] * y[2] - y[1] * x[2];
    float r1 = x[2] * y[0] - y[2] * x[0];
    float r2 = x[0] * y[1] - y[0] * x[1];
    o[0] = r0; o[1] = r1; o[2] = r2;
}

/* Camera.hpp:35-42 ctor, :86-97 updateCameraVectors */
void o_camera_init(o_camera *c, const float pos[3], float yaw, float pitch) {
    memcpy(c->position, pos, 12);
    c->world_up[0] = 0.0f; c->world_up[1] = 1.0f; c->world_up[2] = 0.0f;
    c->yaw = yaw; c->pitch = pitch;
    float f[3];
    /* Camera.hpp:90-92 calls the unqualified ::cos/::sin => the double overloads:
     * float radians are promoted, the product is formed in double and rounded
     * once on assignment (verified against the reference build, camera.json). */
    f[0] = (float)(cos((double)o_radians(yaw)) * cos((double)o_radians(pitch)));
    f[1] = (float)sin((double)o_radians(pitch));
    f[2] = (float)(sin((double)o_radians(yaw)) * cos((double)o_radians(pitch)));
    normalize3(f, c->front);
    float t[3];
    cross3(c->front, c->world_up, t);
    normalize3(t, c->right);
    cross3(c->right, c->front, t);
    normalize3(t, c->up);
}

/* Camera.hpp:44-47 -> glm/ext/matrix_transform.inl:153-174 lookAtRH */
void o_camera_view(const o_camera *c, float m[16]) {
    float center[3] = {c->position[0] + c->front[0], c->position[1] + c->front[1], c->position[2] + c->front[2]};
    float d[3] = {center[0] - c->position[0], center[1] - c->position[1], center[2] - c->position[2]};
    float f[3], s[3], u[3], t[3];
    normalize3(d, f);
    cross3(f, c->up, t);
    normalize3(t, s);
    cross3(s, f, u);
    memset(m, 0, 64);
    m[15] = 1.0f;
    m[0 * 4 + 0] = s[0]; m[1 * 4 + 0] = s[1]; m[2 * 4 + 0] = s[2];
    m[0 * 4 + 1] = u[0]; m[1 * 4 + 1] = u[1]; m[2 * 4 + 1] = u[2];
    m[0 * 4 + 2] = -f[0]; m[1 * 4 + 2] = -f[1]; m[2 * 4 + 2] = -f[2];
    m[3 * 4 + 0] = -dot3(s, c->position);
    m[3 * 4 + 1] = -dot3(u, c->position);
    m[3 * 4 + 2] = dot3(f, c->position);
}

/* glm/ext/matrix_clip_space.inl:249-262 perspectiveRH_NO */
void o_perspective(float fovy, float aspect, float zn, float zf, float m[16]) {
    float th = tanf(fovy / 2.0f);
    memset(m, 0, 64);
    m[0 * 4 + 0] = 1.0f / (aspect * th);
    m[1 * 4 + 1] = 1.0f / th;
    m[2 * 4 + 2] = -(zf + zn) / (zf - zn);
    m[2 * 4 + 3] = -1.0f;
    m[3 * 4 + 2] = -(2.0f * zf * zn) / (zf - zn);
}

/* glm/detail/func_matrix.inl:347-405 compute_inverse<4,4> */
void o_mat4_inverse(const float a[16], float inv[16]) {
#define M(c, r) a[(c) * 4 + (r)]
    float c00 = M(2,2) * M(3,3) - M(3,2) * M(2,3);
    float c02 = M(1,2) * M(3,3) - M(3,2) * M(1,3);
    float c03 = M(1,2) * M(2,3) - M(2,2) * M(1,3);
    float c04 = M(2,1) * M(3,3) - M(3,1) * M(2,3);
    float c06 = M(1,1) * M(3,3) - M(3,1) * M(1,3);
    float c07 = M(1,1) * M(2,3) - M(2,1) * M(1,3);
    float c08 = M(2,1) * M(3,2) - M(3,1) * M(2,2);
    float c10 = M(1,1) * M(3,2) - M(3,1) * M(1,2);
    float c11 = M(1,1) * M(2,2) - M(2,1) * M(1,2);
    float c12 = M(2,0) * M(3,3) - M(3,0) * M(2,3);
    float c14 = M(1,0) * M(3,3) - M(3,0) * M(1,3);
    float c15 = M(1,0) * M(2,3) - M(2,0) * M(1,3);
    float c16 = M(2,0) * M(3,2) - M(3,0) * M(2,2);
    float c18 = M(1,0) * M(3,2) - M(3,0) * M(1,2);
    float c19 = M(1,0) * M(2,2) - M(2,0) * M(1,2);
    float c20 = M(2,0) * M(3,1) - M(3,0) * M(2,1);
    float c22 = M(1,0) * M(3,1) - M(3,0) * M(1,1);
    float c23 = M(1,0) * M(2,1) - M(2,0) * M(1,1);
    float f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11};
    float f3[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
    float v0[4] = {M(1,0), M(0,0), M(0,0), M(0,0)}, v1[4] = {M(1,1), M(0,1), M(0,1), M(0,1)};
    float v2[4] = {M(1,2), M(0,2), M(0,2), M(0,2)}, v3[4] = {M(1,3), M(0,3), M(0,3), M(0,3)};
    static const float sa[4] = {+1, -1, +1, -1}, sb[4] = {-1, +1, -1, +1};
    float I[16];
    for (int i = 0; i < 4; i++) {
        float i0 = (v1[i] * f0[i] - v2[i] * f1[i]) + v3[i] * f2[i];
        float i1 = (v0[i] * f0[i] - v2[i] * f3[i]) + v3[i] * f4[i];
        float i2 = (v0[i] * f1[i] - v1[i] * f3[i]) + v3[i] * f5[i];
        float i3 = (v0[i] * f2[i] - v1[i] * f4[i]) + v2[i] * f5[i];
        I[0 * 4 + i] = i0 * sa[i];
        I[1 * 4 + i] = i1 * sb[i];
        I[2 * 4 + i] = i2 * sa[i];
        I[3 * 4 + i] = i3 * sb[i];
    }
    float d0 = M(0,0) * I[0 * 4 + 0], d1 = M(0,1) * I[1 * 4 + 0];
    float d2 = M(0,2) * I[2 * 4 + 0], d3 = M(0,3) * I[3 * 4 + 0];
    float det = (d0 + d1) + (d2 + d3);
    float ood = 1.0f / det;
    for (int i = 0; i < 16; i++) inv[i] = I[i] * ood;
#undef M
}

/* src/main.cpp:808-813 */
void o_camera_ubo(const o_camera *c, int width, int height, float inv_proj[16], float inv_view[16], float cam_pos[4]) {
    float view[16], proj[16];
    o_camera_view(c, view);
    o_perspective(o_radians(45.0f), (float)width / (float)height, 0.1f, 1000.0f, proj);
    o_mat4_inverse(proj, inv_proj);
    o_mat4_inverse(view, inv_view);
    cam_pos[0] = c->position[0]; cam_pos[1] = c->position[1]; cam_pos[2] = c->position[2]; cam_pos[3] = 1.0f;
}
