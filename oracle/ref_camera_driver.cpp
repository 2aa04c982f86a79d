// ref_camera_driver.cpp -- TEST INFRASTRUCTURE ONLY.
// Harness around the REFERENCE's own include/Camera.hpp + vendored glm 1.0.0
// (compiled in place from /root/reference/include; nothing is copied).
// Emits, as JSON, the bit patterns of the Camera UBO the reference's frame loop
// builds (src/main.cpp:808-813) for a list of poses/frame sizes, plus the
// light direction of src/main.cpp:483. Output -> tests/golden/camera.json.
#include <Camera.hpp>
#include <cstdio>
#include <cstring>
#include <cstdint>

static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static void dump16(const char* name, const glm::mat4& m, bool comma) {
    std::printf("    \"%s\": [", name);
    for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++)
        std::printf("%u%s", bits(m[c][r]), (c == 3 && r == 3) ? "" : ", ");
    std::printf("]%s\n", comma ? "," : "");
}

struct Pose { float x, y, z, yaw, pitch; int w, h; };

int main() {
    const Pose poses[] = {
        {34.0f, 60.0f, 34.0f, -90.0f, 0.0f, 1280, 720},      // reference default (main.cpp:70)
        {34.0f, 60.0f, 34.0f, -90.0f, 0.0f, 256, 256},
        {63.5f, 60.5f, 140.5f, -90.0f, -10.0f, 1920, 1080},  // dragon framing pose
        {48.5f, 60.5f, 170.5f, -90.0f, -12.0f, 1280, 720},   // monu9
        {60.5f, 80.5f, 200.5f, -90.0f, -20.0f, 3840, 2160},  // nature
        {32.5f, 40.5f, 150.5f, -90.0f, -8.0f, 256, 256},     // config 1 synthetic scene
        {512.5f, 420.5f, 1000.5f, -90.0f, -20.0f, 1920, 1080}, // config 4 terrain
        {10.25f, 33.75f, -20.5f, 37.0f, 21.5f, 640, 360},    // off-axis
        {-100.5f, 200.25f, 300.125f, 135.0f, -45.0f, 800, 600},
        {0.0f, 0.0f, 0.0f, 0.0f, 89.0f, 64, 36},
    };
    std::printf("{\n  \"generator\": \"oracle/ref_camera_driver.cpp built against the reference's Camera.hpp + glm 1.0.0\",\n  \"cases\": [\n");
    const int n = (int)(sizeof(poses) / sizeof(poses[0]));
    for (int i = 0; i < n; i++) {
        const Pose& p = poses[i];
        Camera cam(glm::vec3(p.x, p.y, p.z), glm::vec3(0.0f, 1.0f, 0.0f), p.yaw, p.pitch);
        glm::mat4 view = cam.GetViewMatrix();
        glm::mat4 proj = glm::perspective(glm::radians(45.0f), (float)p.w / (float)p.h, 0.1f, 1000.0f);
        glm::mat4 ip = glm::inverse(proj), iv = glm::inverse(view);
        std::printf("  {\n    \"pos\": [%u, %u, %u], \"yaw\": %u, \"pitch\": %u, \"width\": %d, \"height\": %d,\n",
                    bits(p.x), bits(p.y), bits(p.z), bits(p.yaw), bits(p.pitch), p.w, p.h);
        std::printf("    \"front\": [%u, %u, %u], \"right\": [%u, %u, %u], \"up\": [%u, %u, %u],\n",
                    bits(cam.Front.x), bits(cam.Front.y), bits(cam.Front.z), bits(cam.Right.x), bits(cam.Right.y),
                    bits(cam.Right.z), bits(cam.Up.x), bits(cam.Up.y), bits(cam.Up.z));
        dump16("view", view, true);
        dump16("proj", proj, true);
        dump16("inv_proj", ip, true);
        dump16("inv_view", iv, false);
        std::printf("  }%s\n", i + 1 < n ? "," : "");
    }
    glm::vec3 l = glm::normalize(glm::vec3(0.3481553f, 0.870388f, 0.3481553f));
    std::printf("  ],\n  \"light_dir\": [%u, %u, %u]\n}\n", bits(l.x), bits(l.y), bits(l.z));
    return 0;
}
