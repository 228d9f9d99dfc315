// Primitives.h — reflective mesh primitives (reference src/geometry/Primitives.h:13-65, .cpp:6-216):
// procedural plane / UV sphere and an OBJ loader with the reference's Y flip (Primitives.cpp:175,179).  The geometry
// comes from the C ABI (grt_host_primitive_*, grt_host_obj_*: csrc/grt_host.cpp); this class keeps the bookkeeping.
#pragma once
#include <cstddef>
#include <string>
#include <vector>

#include "VecMath.h"

// column-major 4x4 like glm::mat4: m[c][r]
struct Mat4
{
    float m[4][4];
    float* operator[](int c) { return m[c]; }
    const float* operator[](int c) const { return m[c]; }
    static Mat4 identity()
    {
        Mat4 r{};
        for (int i = 0; i < 4; i++) r.m[i][i] = 1.0f;
        return r;
    }
    static Mat4 translate(float3 t)
    {
        Mat4 r = identity();
        r.m[3][0] = t.x; r.m[3][1] = t.y; r.m[3][2] = t.z;
        return r;
    }
    // glm mat4 * vec4(v,1): m[0]*v.x + m[1]*v.y + m[2]*v.z + m[3]
    float3 point(const float3& v) const
    {
        return make_float3(m[0][0] * v.x + m[1][0] * v.y + m[2][0] * v.z + m[3][0],
                           m[0][1] * v.x + m[1][1] * v.y + m[2][1] * v.z + m[3][1],
                           m[0][2] * v.x + m[1][2] * v.y + m[2][2] * v.z + m[3][2]);
    }
    // glm::mat3(m) * n (src/GaussianTracer.cpp:659-662: NOT the inverse transpose)
    float3 dir(const float3& v) const
    {
        return make_float3(m[0][0] * v.x + m[1][0] * v.y + m[2][0] * v.z,
                           m[0][1] * v.x + m[1][1] * v.y + m[2][1] * v.z,
                           m[0][2] * v.x + m[1][2] * v.y + m[2][2] * v.z);
    }
};

struct Primitive
{
    size_t      index;
    std::string type;
    size_t      instanceIndex;

    std::vector<float3>       vertices;
    std::vector<unsigned int> indices;
    std::vector<float3>       normals;

    size_t vertex_count;

    Mat4 transform;
};

class Primitives
{
public:
    Primitive createPlane(float3 position);
    Primitive createSphere(float3 position);
    Primitive createLoadMesh(std::string filename, float3 position);

    std::vector<Primitive>& getPrimitives() { return m_primitives; }
    void clearPrimitives()
    {
        m_primitives.clear();
        numberOfMesh = numberOfPlane = numberOfSphere = numberOfLoaded = 0;
    }

    size_t& getMeshCount() { return numberOfMesh; }
    size_t& getPlaneCount() { return numberOfPlane; }
    size_t& getSphereCount() { return numberOfSphere; }
    size_t& getLoadedCount() { return numberOfLoaded; }

private:
    Mat4 getInitialTransform(float3 position) { return Mat4::translate(position); } // rotations 0, scale 1
    Primitive registerPrimitive(Primitive p, const char* type, size_t& counter, float3 position);
    Primitive createProcedural(int kind, const char* type, size_t& counter, float3 position);

    std::vector<Primitive> m_primitives;
    size_t numberOfMesh = 0, numberOfPlane = 0, numberOfSphere = 0, numberOfLoaded = 0;
};
