// Camera.hpp -- yaw/pitch fly camera of the host API (reference: include/Camera.hpp:18-98).
// Same public members and methods; vector/matrix types come from vrt_math.hpp
// (bit-compatible with the glm calls the reference makes).
#ifndef VRT_CAMERA_HPP
#define VRT_CAMERA_HPP
#include <cmath>
#include <vrt_math.hpp>

enum Camera_Movement { FORWARD, BACKWARD, LEFT, RIGHT };

const float YAW = -90.0f;
const float PITCH = 0.0f;
const float SPEED = 20.5f;
const float SENSITIVITY = 0.1f;

class Camera {
public:
    vrtm::vec3 Position, Front, Up, Right, WorldUp;
    float Yaw, Pitch;
    float MovementSpeed, MouseSensitivity;

    Camera(vrtm::vec3 position = vrtm::vec3(0.0f, 0.0f, 0.0f), vrtm::vec3 up = vrtm::vec3(0.0f, 1.0f, 0.0f),
           float yaw = YAW, float pitch = PITCH)
        : Front(vrtm::vec3(0.0f, 0.0f, -1.0f)), MovementSpeed(SPEED), MouseSensitivity(SENSITIVITY) {
        Position = position;
        WorldUp = up;
        Yaw = yaw;
        Pitch = pitch;
        updateCameraVectors();
    }

    vrtm::mat4 GetViewMatrix() const { return vrtm::lookAt(Position, Position + Front, Up); }

    void ProcessKeyboard(Camera_Movement direction, float deltaTime) {
        const float velocity = MovementSpeed * deltaTime;
        if (direction == FORWARD) Position += Front * velocity;
        if (direction == BACKWARD) Position -= Front * velocity;
        if (direction == LEFT) Position -= Right * velocity;
        if (direction == RIGHT) Position += Right * velocity;
    }

    void ProcessMouseMovement(float xoffset, float yoffset, bool constrainPitch = true) {
        Yaw += xoffset * MouseSensitivity;
        Pitch += yoffset * MouseSensitivity;
        if (constrainPitch) {
            if (Pitch > 89.0f) Pitch = 89.0f;
            if (Pitch < -89.0f) Pitch = -89.0f;
        }
        updateCameraVectors();
    }

    // The 144-byte Camera block the dispatch consumes (reference src/main.cpp:808-813):
    // inverse(perspective(45deg, w/h, 0.1, 1000)), inverse(view), vec4(Position, 1).
    void FillDispatchBlock(int width, int height, float inv_projection[16], float inv_view[16], float camera_pos[4]) const {
        const vrtm::mat4 proj = vrtm::perspective(vrtm::radians(45.0f), (float)width / (float)height, 0.1f, 1000.0f);
        const vrtm::mat4 ip = vrtm::inverse(proj), iv = vrtm::inverse(GetViewMatrix());
        for (int i = 0; i < 16; ++i) { inv_projection[i] = ip.data()[i]; inv_view[i] = iv.data()[i]; }
        camera_pos[0] = Position.x; camera_pos[1] = Position.y; camera_pos[2] = Position.z; camera_pos[3] = 1.0f;
    }

private:
    void updateCameraVectors() {
        // the reference evaluates these products through the double-precision
        // ::cos/::sin overloads and rounds once (Camera.hpp:90-92)
        const double yr = (double)vrtm::radians(Yaw), pr = (double)vrtm::radians(Pitch);
        vrtm::vec3 front((float)(std::cos(yr) * std::cos(pr)), (float)std::sin(pr), (float)(std::sin(yr) * std::cos(pr)));
        Front = vrtm::normalize(front);
        Right = vrtm::normalize(vrtm::cross(Front, WorldUp));
        Up = vrtm::normalize(vrtm::cross(Right, Front));
    }
};

#endif
