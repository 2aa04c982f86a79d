// Camera.hpp -- the yaw/pitch fly camera of the host API. Class, member, enum and constant names are those of the
// reference's include/Camera.hpp (:18-98) so its main loop compiles against this header; vectors and matrices come
// from vrt_math.hpp, whose functions reproduce the arithmetic of the glm calls the reference makes bit for bit
// (tests/test_host.py pins that on golden camera blocks).
//
// Conventions: right-handed, y up, angles in degrees. Yaw = -90, Pitch = 0 looks down -z. The camera never rolls:
// Right and Up are rebuilt from Front and WorldUp after every change of angle.
#ifndef VRT_CAMERA_HPP
#define VRT_CAMERA_HPP
#include <cmath>
#include <vrt_math.hpp>

// directions ProcessKeyboard understands (window-system independent)
enum Camera_Movement { FORWARD, BACKWARD, LEFT, RIGHT };

// defaults: initial angles (degrees), fly speed (world cells per second), mouse degrees per pixel
const float YAW = -90.0f, PITCH = 0.0f, SPEED = 20.5f, SENSITIVITY = 0.1f;

class Camera {
public:
    vrtm::vec3 Position, Front, Up, Right, WorldUp;  // eye point and the orthonormal frame derived from the angles
    float Yaw, Pitch;                                // degrees
    float MovementSpeed, MouseSensitivity;

    Camera(vrtm::vec3 position = vrtm::vec3(0.0f, 0.0f, 0.0f), vrtm::vec3 up = vrtm::vec3(0.0f, 1.0f, 0.0f),
           float yaw = YAW, float pitch = PITCH)
        : Position(position), Front(0.0f, 0.0f, -1.0f), WorldUp(up), Yaw(yaw), Pitch(pitch), MovementSpeed(SPEED),
          MouseSensitivity(SENSITIVITY) {
        updateCameraVectors();
    }

    // world -> eye transform for the current pose
    vrtm::mat4 GetViewMatrix() const { return vrtm::lookAt(Position, Position + Front, Up); }

    // moves the eye by MovementSpeed * deltaTime along (or against) Front / Right
    void ProcessKeyboard(Camera_Movement direction, float deltaTime) {
        const float step = MovementSpeed * deltaTime;
        switch (direction) {
        case FORWARD: Position += Front * step; break;
        case BACKWARD: Position -= Front * step; break;
        case LEFT: Position -= Right * step; break;
        case RIGHT: Position += Right * step; break;
        }
    }

    // turns by a mouse delta in pixels; with constrainPitch the pitch stays within +-89 degrees so the frame
    // never degenerates at the poles
    void ProcessMouseMovement(float xoffset, float yoffset, bool constrainPitch = true) {
        Yaw += xoffset * MouseSensitivity;
        Pitch += yoffset * MouseSensitivity;
        if (constrainPitch) Pitch = Pitch > 89.0f ? 89.0f : (Pitch < -89.0f ? -89.0f : Pitch);
        updateCameraVectors();
    }

    // Not in the reference's class: the 144-byte camera block vrt_dispatch() consumes, computed the way the
    // reference's frame loop does before its dispatch (src/main.cpp:808-813) --
    // inverse(perspective(45 deg, w / h, 0.1, 1000)), inverse(view), vec4(Position, 1) -- column-major.
    void FillDispatchBlock(int width, int height, float inv_projection[16], float inv_view[16], float camera_pos[4]) const {
        const vrtm::mat4 ip = vrtm::inverse(vrtm::perspective(vrtm::radians(45.0f), (float)width / (float)height, 0.1f, 1000.0f));
        const vrtm::mat4 iv = vrtm::inverse(GetViewMatrix());
        for (int i = 0; i < 16; ++i) {
            inv_projection[i] = ip.data()[i];
            inv_view[i] = iv.data()[i];
        }
        const float eye[4] = {Position.x, Position.y, Position.z, 1.0f};
        for (int i = 0; i < 4; ++i) camera_pos[i] = eye[i];
    }

private:
    // Front from the two angles, then Right and Up. The products cos(yaw) * cos(pitch) and sin(yaw) * cos(pitch) are
    // formed in double and rounded to float once, which is what the reference's expression evaluates to
    // (Camera.hpp:90-92 resolves to the double overloads of cos / sin); doing them in float changes the last bit
    // of Front and with it whole frames.
    void updateCameraVectors() {
        const double yaw = (double)vrtm::radians(Yaw), pitch = (double)vrtm::radians(Pitch);
        const double cp = std::cos(pitch);
        Front = vrtm::normalize(vrtm::vec3((float)(std::cos(yaw) * cp), (float)std::sin(pitch), (float)(std::sin(yaw) * cp)));
        Right = vrtm::normalize(vrtm::cross(Front, WorldUp));
        Up = vrtm::normalize(vrtm::cross(Right, Front));
    }
};

#endif
