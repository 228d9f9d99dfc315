#include "Camera.h"

#include "../../include/grt.h"

// src/Camera.cpp:3-13 — delegated to the C ABI's host helper so ctypes and C++ users share one definition.
void Camera::UVWFrame(float3& U, float3& V, float3& W) const
{
    const float e[3] = {m_eye.x, m_eye.y, m_eye.z}, l[3] = {m_lookat.x, m_lookat.y, m_lookat.z},
                u[3] = {m_up.x, m_up.y, m_up.z};
    float uu[3], vv[3], ww[3];
    grt_host_uvw_frame(e, l, u, m_fovY, m_aspectRatio, uu, vv, ww);
    U = make_float3(uu[0], uu[1], uu[2]);
    V = make_float3(vv[0], vv[1], vv[2]);
    W = make_float3(ww[0], ww[1], ww[2]);
}
