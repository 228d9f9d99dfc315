#include "GaussianTracer.h"

#include <stdexcept>

#include "../../include/grt.h"
#include "HipGlue.h"

// every failure throws std::runtime_error, as the reference's CUDA_CHECK / OPTIX_CHECK do (src/Exception.h:19-80)
void GaussianTracer::check(int rc, const char* what)
{
    if (rc != GRT_OK) throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + grt_last_error(m_ctx));
}

GaussianTracer::GaussianTracer(const std::string& filename) : m_gsData(filename) // src/GaussianTracer.cpp:20-52
{
    stream = nullptr;
    params = {};
    particle_count = m_gsData.getVertexCount();
    alpha_min = 0.01f;
    current_lookat = make_float3(0.0f);
}

GaussianTracer::~GaussianTracer()
{
    if (m_ctx) grt_destroy(m_ctx);
    hipglue::streamDestroy(stream);
    delete primitives;
}

void GaussianTracer::initializeOptix() // src/GaussianTracer.cpp:72-83
{
    // (grt_create makes m_device this thread's current device: the stream of initializeParams is created on it)
    if (grt_create(&m_ctx, m_device) != GRT_OK) throw std::runtime_error(grt_last_error(nullptr));
    createGaussianParticlesBVH();
    initializeParams();
}

void GaussianTracer::createGaussianParticlesBVH() // src/GaussianTracer.cpp:297-317 -> upload + device LBVH
{
    const size_t n = particle_count;
    std::vector<float> pos(n * 3), scale(n * 3), quat(n * 4), op(n), sh(n * 48);
    for (size_t i = 0; i < n; i++) {
        const GaussianParticle& p = m_gsData.particles[i];
        pos[i * 3] = p.position.x; pos[i * 3 + 1] = p.position.y; pos[i * 3 + 2] = p.position.z;
        scale[i * 3] = p.scale.x; scale[i * 3 + 1] = p.scale.y; scale[i * 3 + 2] = p.scale.z;
        for (int k = 0; k < 4; k++) quat[i * 4 + k] = p.rotation[k];
        op[i] = p.opacity;
        for (int k = 0; k < 16; k++) { sh[i * 48 + k * 3] = p.sh[k].x; sh[i * 48 + k * 3 + 1] = p.sh[k].y; sh[i * 48 + k * 3 + 2] = p.sh[k].z; }
    }
    grt_gaussians g{pos.data(), scale.data(), quat.data(), op.data(), sh.data()};
    check(grt_upload_gaussians(m_ctx, &g, n), "grt_upload_gaussians");
    check(grt_build_bvh(m_ctx, alpha_min), "grt_build_bvh");
}

void GaussianTracer::initializeParams() // src/GaussianTracer.cpp:475-506
{
    params.output_buffer    = nullptr;
    params.handle           = 1;
    params.t_min            = 1e-3f;
    params.t_max            = 1e5f;
    params.minTransmittance = 0.001f;
    params.alpha_min        = alpha_min;
    params.sh_degree_max    = 0;
    params.mesh_handle      = 0;
    params.mode_fisheye     = false;
    params.type             = MIRROR;
    params.max_bounces      = 32;
    stream = hipglue::streamCreate();
}

void GaussianTracer::render(CUDAOutputBuffer& output_buffer) // src/GaussianTracer.cpp:508-538
{
    // (the reference pre-clears the frame in fisheye mode because its raygen leaves r > 1 pixels unwritten;
    //  the HIP kernel writes 0 there itself)
    uchar3* result = output_buffer.map();
    params.output_buffer = result;
    grt_params p{};
    fillParams(&p);
    check(grt_render(m_ctx, &p, reinterpret_cast<uint8_t*>(result), nullptr, 0, 0, params.width, params.height, stream), "grt_render");
    output_buffer.unmap();
    hipglue::streamSync(stream); // CUDA_SYNC_CHECK, :537
    // the frame is done: a wave that had to give up on live rays (watchdog, stack guard, stalled passes) left its reason
    // in the context's sticky error word — thrown here, as OptiX exceptions are in the reference (:114-119, Exception.h:31-80)
    check(grt_sync(m_ctx), "render");
}

void GaussianTracer::fillParams(void* out) const // Params (src/Parameters.h:42-74) -> the C ABI's grt_params
{
    grt_params& p = *static_cast<grt_params*>(out);
    p = grt_params{};
    p.width = params.width; p.height = params.height; p.sh_degree_max = params.sh_degree_max;
    p.eye[0] = params.eye.x; p.eye[1] = params.eye.y; p.eye[2] = params.eye.z;
    p.U[0] = params.U.x; p.U[1] = params.U.y; p.U[2] = params.U.z;
    p.V[0] = params.V.x; p.V[1] = params.V.y; p.V[2] = params.V.z;
    p.W[0] = params.W.x; p.W[1] = params.W.y; p.W[2] = params.W.z;
    p.t_min = params.t_min; p.t_max = params.t_max; p.minTransmittance = params.minTransmittance; p.alpha_min = params.alpha_min;
    p.mode_fisheye = params.mode_fisheye ? 1 : 0;
    p.type = params.type;
    p.max_bounces = params.max_bounces;
}

void GaussianTracer::renderTiles(unsigned char* d_tiles, unsigned int tile_w, unsigned int tile_h, unsigned int first,
                                 unsigned int stride, unsigned int count)
{
    hipglue::setDevice(m_device);
    grt_params p{};
    fillParams(&p);
    check(grt_render_tiles(m_ctx, &p, d_tiles, nullptr, tile_w, tile_h, first, stride, count, stream), "grt_render_tiles");
}

void GaussianTracer::assembleTiles(const unsigned char* d_gathered, unsigned int world, unsigned int max_count, unsigned int tile_w,
                                   unsigned int tile_h, CUDAOutputBuffer& output_buffer)
{
    hipglue::setDevice(m_device);
    uchar3* result = output_buffer.map();
    check(grt_assemble_tiles(m_ctx, d_gathered, world, max_count, tile_w, tile_h, params.width, params.height,
                             reinterpret_cast<uint8_t*>(result), stream), "grt_assemble_tiles");
    output_buffer.unmap();
}

void GaussianTracer::sync()
{
    hipglue::setDevice(m_device);
    hipglue::streamSync(stream);
    check(grt_sync(m_ctx), "render");
}

void GaussianTracer::updateCamera(Camera& camera, bool& camera_changed) // src/GaussianTracer.cpp:540-551
{
    if (!camera_changed) return;
    camera_changed = false;
    camera.setAspectRatio(static_cast<float>(params.width) / static_cast<float>(params.height));
    params.eye = camera.eye();
    camera.UVWFrame(params.U, params.V, params.W);
    current_lookat = camera.lookat();
}

void GaussianTracer::removePrimitive() // src/GaussianTracer.cpp:553-566
{
    primitives->clearPrimitives();
    check(grt_set_meshes(m_ctx, nullptr, 0), "grt_set_meshes");
    params.mesh_handle = 0;
}

void GaussianTracer::setRenderType(unsigned int renderType) { params.type = renderType; }

float3 GaussianTracer::primitivePosition() const // src/GaussianTracer.cpp:580-588
{
    const float3 cameraPosition = params.eye;
    const float cameraWeight = 0.75f, gaussianWeight = 1.0f - cameraWeight;
    return make_float3(current_lookat.x * gaussianWeight + cameraPosition.x * cameraWeight,
                       current_lookat.y * gaussianWeight + cameraPosition.y * cameraWeight,
                       current_lookat.z * gaussianWeight + cameraPosition.z * cameraWeight);
}

// createGAS + createIAS + sendGeometryAttributesToDevice (src/GaussianTracer.cpp:592-600,653-709): every
// primitive is placed in world space by its transform, normals by mat3(transform), and handed over whole;
// the library rebuilds the mesh LBVH (no per-update leak, unlike :672-709,728).
void GaussianTracer::uploadMeshes(bool same_topology)
{
    std::vector<Primitive>& ps = primitives->getPrimitives();
    std::vector<std::vector<float>> v(ps.size()), n(ps.size());
    std::vector<grt_mesh> ms(ps.size());
    for (size_t k = 0; k < ps.size(); k++) {
        const Primitive& p = ps[k];
        v[k].resize(p.vertex_count * 3);
        n[k].resize(p.vertex_count * 3);
        for (size_t i = 0; i < p.vertex_count; i++) {
            const float3 w = p.transform.point(p.vertices[i]), nn = p.transform.dir(p.normals[i]);
            v[k][i * 3] = w.x; v[k][i * 3 + 1] = w.y; v[k][i * 3 + 2] = w.z;
            n[k][i * 3] = nn.x; n[k][i * 3 + 1] = nn.y; n[k][i * 3 + 2] = nn.z;
        }
        ms[k] = grt_mesh{v[k].data(), n[k].data(), (uint32_t)p.vertex_count, p.indices.data(), (uint32_t)(p.indices.size() / 3)};
    }
    // a moved primitive keeps its topology: re-fit the mesh LBVH; anything else (or a refused refit) rebuilds it
    if (!same_topology || grt_update_meshes(m_ctx, ms.data(), (uint32_t)ms.size()) != GRT_OK)
        check(grt_set_meshes(m_ctx, ms.data(), (uint32_t)ms.size()), "grt_set_meshes");
    params.mesh_handle = ps.empty() ? 0 : 1;
}

void GaussianTracer::createPlane() { primitives->createPlane(primitivePosition()); uploadMeshes(); }
void GaussianTracer::createSphere() { primitives->createSphere(primitivePosition()); uploadMeshes(); }
void GaussianTracer::createLoadMesh(std::string filename) { primitives->createLoadMesh(filename, primitivePosition()); uploadMeshes(); }

void GaussianTracer::updateInstanceTransforms(Primitive& p) // src/GaussianTracer.cpp:711-736
{
    std::vector<Primitive>& ps = primitives->getPrimitives();
    if (p.instanceIndex < ps.size() && &ps[p.instanceIndex] != &p) ps[p.instanceIndex].transform = p.transform;
    uploadMeshes(true);
}

void GaussianTracer::setSize(unsigned int width, unsigned int height) // src/GaussianTracer.cpp:796-800
{
    params.width = width;
    params.height = height;
}

float GaussianTracer::lastKernelMs()
{
    float ms = 0.f;
    check(grt_last_kernel_ms(m_ctx, &ms), "grt_last_kernel_ms");
    return ms;
}

// ---- HIPOutputBuffer (src/CUDAOutputBuffer.cpp:3-64); the GL-interop form is HIPOutputBufferGL.cpp ----
#ifdef GRT_WITH_GL
HIPOutputBuffer::HIPOutputBuffer(int32_t width, int32_t height, bool gl_interop) : m_gl(gl_interop) { resize(width, height); }
#else
HIPOutputBuffer::HIPOutputBuffer(int32_t width, int32_t height) { resize(width, height); }
void HIPOutputBuffer::resizeGL(int32_t, int32_t) {}
void HIPOutputBuffer::releaseGL() {}
uchar3* HIPOutputBuffer::mapGL() { return nullptr; }
void HIPOutputBuffer::unmapGL() {}
#endif
HIPOutputBuffer::~HIPOutputBuffer()
{
    if (m_gl) releaseGL();
    hipglue::deviceFree(m_device);
    hipglue::hostFreePinned(m_host);
}
void HIPOutputBuffer::resize(int32_t width, int32_t height)
{
    if (m_width == width && m_height == height && (m_device || m_pbo)) return;
    if (m_gl) { resizeGL(width, height); return; }
    hipglue::deviceFree(m_device);
    hipglue::hostFreePinned(m_host);
    m_device = nullptr; m_host = nullptr;
    m_width = width; m_height = height;
    const size_t bytes = (size_t)width * height * 3;
    m_device = static_cast<uchar3*>(hipglue::deviceAlloc(bytes));
    m_host = static_cast<uchar3*>(hipglue::hostAllocPinned(bytes));
}
uchar3* HIPOutputBuffer::map() { return m_gl ? mapGL() : m_device; }
void HIPOutputBuffer::unmap()
{
    if (m_gl) unmapGL();
    else hipglue::copyToHostAsync(m_host, m_device, (size_t)m_width * m_height * 3, m_stream);
}
const std::vector<unsigned char>& HIPOutputBuffer::download()
{
    m_copy.resize((size_t)m_width * m_height * 3);
    const uchar3* src = m_gl ? mapGL() : m_device; // (GL: the PBO is mapped for the copy and handed back — also when the copy throws)
    struct Unmap { HIPOutputBuffer* b; ~Unmap() { if (b) b->unmapGL(); } } guard{m_gl ? this : nullptr};
    hipglue::copyToHost(m_copy.data(), src, m_copy.size());
    return m_copy;
}
