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
    // one tracer per GPU (the CLI's --gpus N): the device this tracer's scene lives on (before
    // initializeOptix), and the tracer's share of a frame — tiles first + j * stride (j < count) of the tile_w x tile_h
    // grid into the compact device buffer [count][tile_h][tile_w][3] (asynchronous on `stream`)
    void setDevice(int device) { m_device = device; }
    int device() const { return m_device; }
    void renderTiles(unsigned char* d_tiles, unsigned int tile_w, unsigned int tile_h, unsigned int first, unsigned int stride,
                     unsigned int count);
    // rank 0: the ranks' compact buffers [world][max_count][tile_h][tile_w][3] (device memory of this tracer's GPU) -> frame
    void assembleTiles(const unsigned char* d_gathered, unsigned int world, unsigned int max_count, unsigned int tile_w,
                       unsigned int tile_h, CUDAOutputBuffer& output_buffer);
    void sync(); // waits for this tracer's stream and throws when a wave had to give up on live rays (grt_sync)

private:
    void initializeParams();
    void createGaussianParticlesBVH();
    void uploadMeshes(bool same_topology = false);
    float3 primitivePosition() const;
    void check(int rc, const char* what);

    // (void*: grt_params by value would drag include/grt.h into this header)
    void fillParams(void* grt_params_out) const;
    grt_ctx* m_ctx = nullptr;
    int m_device = 0;
    GaussianData m_gsData;
    size_t particle_count;
    float alpha_min;
    float3 current_lookat;
};
