// Camera.h — ray-basis part of the reference's Camera (src/Camera.h:8-63, src/Camera.cpp:3-13).
// The view/projection matrices (gizmo only) are out of scope.
#pragma once
#include "VecMath.h"

class Camera
{
public:
    Camera()
        : m_eye(make_float3(0.0f)), m_lookat(make_float3(0.0f)), m_up(make_float3(0.0f)), m_fovY(0.0f), m_aspectRatio(1.0f) {}

    const float3& eye() const { return m_eye; }
    const float3& lookat() const { return m_lookat; }

    void setEye(const float3& val) { m_eye = val; }
    void setLookat(const float3& val) { m_lookat = val; }
    void setUp(const float3& val) { m_up = val; }
    void setFovY(float val) { m_fovY = val; }
    void setAspectRatio(float val) { m_aspectRatio = val; }
    void setMoveSpeed(const float& val) { m_moveSpeed = val; }

    void UVWFrame(float3& U, float3& V, float3& W) const;

private:
    float3 m_eye, m_lookat, m_up;
    float  m_fovY;
    float  m_aspectRatio;
    float  m_moveSpeed = 1.0f;
};
