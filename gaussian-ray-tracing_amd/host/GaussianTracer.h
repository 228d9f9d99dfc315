// GaussianTracer.h — the reference's renderer class (src/GaussianTracer.h:27-111), same public surface,
// backed by libgrt_hip.so through the C ABI (include/grt.h) instead of OptiX.
#pragma once
#include <string>
#include <vector>

#include "Camera.h"
#include "GaussianData.h"
#include "HIPOutputBuffer.h"
#include "Parameters.h"
#include "Primitives.h"

struct grt_ctx;

enum ReflectionPrimitiveType { PLANE = 0, SPHERE, CUSTOM };

class GaussianTracer
{
public:
    GaussianTracer(const std::string& filename);
    ~GaussianTracer();

    void initializeOptix();                 // name kept for drop-in; builds the HIP scene + LBVH
    void initialize() { initializeOptix(); }

    void render(CUDAOutputBuffer& output_buffer); // CUDAOutputBuffer = HIPOutputBuffer (HIPOutputBuffer.h)

    void updateCamera(Camera& camera, bool& camera_changed);
    void updateInstanceTransforms(Primitive& p);

    std::vector<Primitive>& getPrimitives() { return primitives->getPrimitives(); }
    void removePrimitive();

    void setRenderType(unsigned int renderType);
    void setSize(unsigned int width, unsigned int height);
    float3 getGaussianCenter() { return m_gsData.getCenter(); }

    Params params;
    void*  stream; // hipStream_t (CUstream in the reference)

    Primitives* primitives = new Primitives();
    void createPlane();
    void createSphere();
    void createLoadMesh(std::string filename);

    // extras for headless use
    float lastKernelMs();
    grt_ctx* context() { return m_ctx; }

private:
    void initializeParams();
    void createGaussianParticlesBVH();
    void uploadMeshes(bool same_topology = false);
    float3 primitivePosition() const;
    void check(int rc, const char* what);

    grt_ctx* m_ctx = nullptr;
    GaussianData m_gsData;
    size_t particle_count;
    float alpha_min;
    float3 current_lookat;
};
