#include "Primitives.h"

#include <stdexcept>

#include "../../include/grt.h"

// The geometry itself (lattice primitives, OBJ soup with the reference's Y flip) lives behind the C ABI
// (csrc/grt_host.cpp: grt_host_primitive_*, grt_host_obj_*), shared with the ctypes binding; this file only keeps the
// reference's bookkeeping: per-type counters, instance index = order of creation, initial transform = translation
// (src/geometry/Primitives.cpp:46-59,124-138,193-201).
namespace {
void fill_from_arrays(Primitive& p, const std::vector<float>& v, const std::vector<float>& n, const std::vector<uint32_t>& f)
{
    p.vertices.resize(v.size() / 3);
    p.normals.resize(n.size() / 3);
    for (size_t i = 0; i < p.vertices.size(); i++) {
        p.vertices[i] = make_float3(v[3 * i], v[3 * i + 1], v[3 * i + 2]);
        p.normals[i] = make_float3(n[3 * i], n[3 * i + 1], n[3 * i + 2]);
    }
    p.indices.assign(f.begin(), f.end());
    p.vertex_count = p.vertices.size();
}
} // namespace

Primitive Primitives::registerPrimitive(Primitive p, const char* type, size_t& counter, float3 position)
{
    p.index = counter++;
    p.type = type;
    p.instanceIndex = numberOfMesh++;
    p.transform = getInitialTransform(position);
    m_primitives.push_back(p);
    return p;
}

Primitive Primitives::createProcedural(int kind, const char* type, size_t& counter, float3 position)
{
    uint32_t nv = 0, nf = 0;
    if (grt_host_primitive_counts(kind, &nv, &nf) != GRT_OK) throw std::runtime_error(grt_host_last_error());
    std::vector<float> v((size_t)nv * 3), n((size_t)nv * 3);
    std::vector<uint32_t> f((size_t)nf * 3);
    if (grt_host_primitive_fill(kind, v.data(), n.data(), f.data()) != GRT_OK) throw std::runtime_error(grt_host_last_error());
    Primitive p;
    fill_from_arrays(p, v, n, f);
    return registerPrimitive(p, type, counter, position);
}

Primitive Primitives::createPlane(float3 position) { return createProcedural(GRT_PRIM_PLANE, "Plane", numberOfPlane, position); }
Primitive Primitives::createSphere(float3 position) { return createProcedural(GRT_PRIM_SPHERE, "Sphere", numberOfSphere, position); }

// A parse failure throws (the reference calls exit(1), src/geometry/Primitives.cpp:149-154).
Primitive Primitives::createLoadMesh(std::string filename, float3 position)
{
    uint32_t nv = 0, nf = 0;
    if (grt_host_obj_count(filename.c_str(), &nv, &nf) != GRT_OK) throw std::runtime_error(grt_host_last_error());
    std::vector<float> v((size_t)nv * 3), n((size_t)nv * 3);
    std::vector<uint32_t> f(nv);
    if (grt_host_obj_read(filename.c_str(), nv, v.data(), n.data(), f.data()) != GRT_OK) throw std::runtime_error(grt_host_last_error());
    Primitive p;
    fill_from_arrays(p, v, n, f);
    return registerPrimitive(p, "LoadedMesh", numberOfLoaded, position);
}
