// Parameters.h — launch-parameter ABI of the renderer, field-for-field the reference's struct Params
// (src/Parameters.h:42-74) so gui.cpp / main.cpp code that touches tracer.params keeps compiling
// (gui.cpp:433 writes params.mode_fisheye).  OptiX handles become opaque 64-bit values owned by the HIP
// library; d_particles / d_meshes / traceState remain as (unused) fields for source compatibility.
#pragma once
#include <cstdint>

#include "VecMath.h"

// src/Parameters.h:10-23
#define SH_C0     0.28209479177387814f
#define SH_C1     0.4886025119029199f
#define SH_C2_0   1.0925484305920792f
#define SH_C2_1  -1.0925484305920792f
#define SH_C2_2   0.31539156525252005f
#define SH_C2_3  -1.0925484305920792f
#define SH_C2_4   0.5462742152960396f
#define SH_C3_0  -0.5900435899266435f
#define SH_C3_1   2.890611442640554f
#define SH_C3_2  -0.4570457994644658f
#define SH_C3_3   0.3731763325901154f
#define SH_C3_4  -0.4570457994644658f
#define SH_C3_5   1.445305721320277f
#define SH_C3_6  -0.5900435899266435f

struct GaussianParticle;

struct Mesh
{
    uint3*  faces;
    float3* vertex_normals;
};

typedef uint64_t GrtTraversableHandle; // was OptixTraversableHandle; non-zero == "a BVH exists in the library"

struct Params
{
    uchar3* output_buffer;

    unsigned int width;
    unsigned int height;
    unsigned int sh_degree_max;

    float3 eye;
    float3 U;
    float3 V;
    float3 W;

    float t_min;
    float t_max;
    float minTransmittance;
    float alpha_min;

    GrtTraversableHandle handle;
    GaussianParticle* d_particles; // unused: the library owns the (SoA) device copy

    // Mesh
    GrtTraversableHandle mesh_handle;

    // FishEye
    bool mode_fisheye;

    Mesh* d_meshes;                // unused

    int32_t type;

    unsigned int* traceState;      // unused: the per-ray state lives in a register

    // extension (reference constant MAX_BOUNCES, shaders/tracer.cuh:13)
    unsigned int max_bounces;
};

enum MeshType
{
    MIRROR = 0,
    NORMAL = 1,
    GLASS  = 2
};

enum TraceState
{
    TraceLastGaussianPass = 0,
    TraceGaussianPass     = 1,
    TraceMeshPass         = 2,
    TraceTerminate        = 3
};
