// grt_mesh.h — mesh side of the bounce loop, shared by the per-lane kernels (grt_render.hip) and the one-ray-per-wave mode
// of the tile kernel: closest triangle hit on the mesh LBVH, barycentric normal, closest-hit / miss shading.
// Citations (file:line) are into Ray-Studio2/gaussian-ray-tracing.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "grt_device.h"
#include "grt_internal.h"

namespace grt {

// traceMesh: closest triangle in (tmin, tmax) — shaders/tracer.cuh:266-287
struct MeshHit { bool hit; float t, u, v; uint32_t face; };

// STRIDE = distance between two stack levels in `stk` (per-lane stacks interleaved in LDS: the workgroup size; a stack
// of its own: 1)
template <bool COUNT, int STRIDE>
__device__ __forceinline__ MeshHit mesh_closest_t(const RenderArgs& a, uint32_t* __restrict__ stk, f3 o, f3 d,
                                                  float tmin, float tmax, uint32_t& iters, uint32_t& node_visits)
{
    MeshHit best{false, 0.f, 0.f, 0.f, 0u};
    if (a.mroot == kNoRoot) return best; // mesh_handle == 0 => miss
    const rayinv ri = mk_rayinv(o, d);
    uint32_t sp = 0, cur = a.mroot;
    while (true) {
        iters++;
        if (cur & kLeafBit) {
            const uint32_t first = leaf_first(cur), cnt = leaf_count(cur);
            for (uint32_t j = 0; j < cnt; j++) {
                const float4* __restrict__ tr = a.tri + (size_t)(first + j) * 3;
                const float4 t0 = tr[0], t1 = tr[1], t2 = tr[2];
                const uint32_t face = __float_as_uint(t0.w);
                float t, u, v;
                if (tri_hit(mk3(t0.x, t0.y, t0.z), mk3(t1.x, t1.y, t1.z), mk3(t2.x, t2.y, t2.z), o, d, t, u, v)) {
                    const bool inside = (t > tmin) && (t < tmax);
                    const bool tie = best.hit && (t == best.t) && (face < best.face);
                    if (inside || tie) {
                        best.hit = true; best.t = t; best.u = u; best.v = v; best.face = face;
                        tmax = t;
                    }
                }
            }
            if (sp == 0) break;
            cur = stk[(--sp) * STRIDE];
        } else {
            const float4* __restrict__ q = a.mnodes + (size_t)cur * 4;
            const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            if (COUNT) node_visits++;
            float n0, f0, n1, f1;
            box_interval(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, ri, n0, f0);
            box_interval(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, ri, n1, f1);
            const bool h0 = (n0 <= f0) && (f0 >= tmin) && (n0 <= tmax);
            const bool h1 = (n1 <= f1) && (f1 >= tmin) && (n1 <= tmax);
            const uint32_t c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y);
            if (h0 && h1) {
                const bool first0 = n0 <= n1;
                stk[(sp++) * STRIDE] = first0 ? c1 : c0;
                cur = first0 ? c0 : c1;
            } else if (h0) {
                cur = c0;
            } else if (h1) {
                cur = c1;
            } else {
                if (sp == 0) break;
                cur = stk[(--sp) * STRIDE];
            }
        }
    }
    return best;
}

// ---- traceMesh for the 64 rays of a wave TOGETHER (camera rays of an 8x8 tile: almost parallel, they meet the same few triangles).
// The wave walks the tree once: a node — both children's boxes in one 64-B record — comes by a scalar load, every lane tests the two
// boxes against ITS ray and its own current closest hit, a child is entered when any lane wants it (the nearer one first, as the first
// voting lane sees it), the far one waits on ONE stack per wave (`stk`: wave-private, height + 2 entries, entries wave-uniform); a
// leaf's triangles come by scalar loads and every lane runs the Moeller-Trumbore test and the closest-hit rule of mesh_closest_t.  A
// lane visits a superset of the nodes its own walk would visit; the closest hit and its tie rule (lowest face at equal t) do not
// depend on the order, (t, u, v) come from the same arithmetic on the same triangle: the same MeshHit as mesh_closest_t, bit for bit.
typedef float v4f_ __attribute__((ext_vector_type(4)));
typedef float v16f_ __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float4 sload16_(const float4* base, uint32_t byte_off)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) char* cptr1;
    typedef const __attribute__((address_space(4))) v4f_* cptr4;
    const v4f_ v = *(cptr4)((cptr1)(uintptr_t)base + byte_off);
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *(const float4*)((const char*)base + byte_off);
#endif
}
__device__ __forceinline__ void sload64_(const float4* base, uint32_t byte_off, float4& q0, float4& q1, float4& q2, float4& q3)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) char* cptr1;
    typedef const __attribute__((address_space(4))) v16f_* cptr16;
    const v16f_ v = *(cptr16)((cptr1)(uintptr_t)base + byte_off);
    q0 = make_float4(v.s0, v.s1, v.s2, v.s3); q1 = make_float4(v.s4, v.s5, v.s6, v.s7);
    q2 = make_float4(v.s8, v.s9, v.sa, v.sb); q3 = make_float4(v.sc, v.sd, v.se, v.sf);
#else
    const float4* p = (const float4*)((const char*)base + byte_off);
    q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = p[3];
#endif
}
__device__ __forceinline__ uint32_t uni_u32(uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
#else
    return v;
#endif
}


template <bool COUNT>
__device__ __forceinline__ MeshHit mesh_closest_wave(const RenderArgs& a, uint32_t* __restrict__ stk, bool have_ray, f3 o, f3 d,
                                                     float tmin, float tmax, uint32_t& node_visits)
{
    MeshHit best{false, 0.f, 0.f, 0.f, 0u};
    if (a.mroot == kNoRoot || __ballot(have_ray) == 0ull) return best; // (mesh_handle == 0 => miss)
    const uint32_t lane = threadIdx.x & 63u;
    const rayinv ri = mk_rayinv(o, d);
    uint32_t sp = 0, cur = a.mroot;
    while (true) {
        cur = uni_u32(cur);
        if (cur & kLeafBit) {
            const uint32_t first = leaf_first(cur), cnt = leaf_count(cur);
            for (uint32_t j = 0; j < cnt; j++) {
                const uint32_t off = (first + j) * 48u;
                const float4 t0 = sload16_(a.tri, off), t1 = sload16_(a.tri, off + 16u), t2 = sload16_(a.tri, off + 32u);
                const uint32_t face = __float_as_uint(t0.w);
                float t, u, v;
                if (have_ray && tri_hit(mk3(t0.x, t0.y, t0.z), mk3(t1.x, t1.y, t1.z), mk3(t2.x, t2.y, t2.z), o, d, t, u, v)) {
                    const bool inside = (t > tmin) && (t < tmax);
                    const bool tie = best.hit && (t == best.t) && (face < best.face);
                    if (inside || tie) {
                        best.hit = true; best.t = t; best.u = u; best.v = v; best.face = face;
                        tmax = t;
                    }
                }
            }
            if (sp == 0) break;
            cur = stk[--sp];
        } else {
            float4 q0, q1, q2, q3;
            sload64_(a.mnodes, cur << 6, q0, q1, q2, q3);
            if (COUNT && have_ray) node_visits++;
            float n0, f0, n1, f1;
            box_interval(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, ri, n0, f0);
            box_interval(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, ri, n1, f1);
            const bool h0 = have_ray && (n0 <= f0) && (f0 >= tmin) && (n0 <= tmax);
            const bool h1 = have_ray && (n1 <= f1) && (f1 >= tmin) && (n1 <= tmax);
            const uint64_t m0 = __ballot(h0), m1 = __ballot(h1);
            const uint32_t c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y);
            if (m0 && m1) {
                const int l0 = (int)__builtin_ctzll(m0 | m1);
                const bool first0 = __shfl((int)((h0 && (!h1 || n0 <= n1)) ? 1 : 0), l0) != 0;
                if (lane == 0u) stk[sp] = first0 ? c1 : c0;
                sp++;
                cur = first0 ? c0 : c1;
            } else if (m0) {
                cur = c0;
            } else if (m1) {
                cur = c1;
            } else {
                if (sp == 0) break;
                cur = stk[--sp];
            }
        }
    }
    return best;
}

// getBarycentricNormal — shaders/tracer.cuh:167-185
__device__ __forceinline__ f3 bary_normal(const RenderArgs& a, const MeshHit& h)
{
    const uint32_t i0 = a.faces[h.face * 3], i1 = a.faces[h.face * 3 + 1], i2 = a.faces[h.face * 3 + 2];
    const f3 n0 = mk3(a.vnormals[i0 * 3], a.vnormals[i0 * 3 + 1], a.vnormals[i0 * 3 + 2]);
    const f3 n1 = mk3(a.vnormals[i1 * 3], a.vnormals[i1 * 3 + 1], a.vnormals[i1 * 3 + 2]);
    const f3 n2 = mk3(a.vnormals[i2 * 3], a.vnormals[i2 * 3 + 1], a.vnormals[i2 * 3 + 2]);
    const float w0 = 1.0f - h.u - h.v, w1 = h.u, w2 = h.v;
    return normalize3(add3(add3(mul3s(n0, w0), mul3s(n1, w1)), mul3s(n2, w2)));
}

enum { LastGaussianPass = 0, GaussianPass = 1, MeshPass = 2, Terminate = 3 }; // src/Parameters.h:85-91

// RayPayload + RayData (shaders/tracer.cuh:24-56) as far as the bounce loop carries them
struct RayState {
    f3 curO, curD, accumColor;
    float accumAlpha, blocking, density;
    uint32_t numBounces, timeout;
};

// __closesthit__closesthit / __miss__miss for one mesh trace (shaders/tracer.cu:112-122,155-187): decides the state,
// the upper end of this iteration's Gaussian segment and the next ray
__device__ __forceinline__ void mesh_shade(const RenderArgs& a, const MeshHit& mh, f3 ray_o, f3 ray_d, int& state,
                                           float& seg_tmax, f3& normal, f3& curO, f3& curD, uint32_t& numBounces)
{
    normal = mk3(0, 0, 0);
    seg_tmax = a.p.t_max; // LastGaussianPass traces to t_max (shaders/tracer.cu:70-75)
    if (mh.hit) {
        float t_hit = mh.t;
        normal = bary_normal(a, mh);
        f3 newDir = mk3(0, 0, 0);
        state = GaussianPass;
        seg_tmax = t_hit;
        if (a.p.type == GRT_MIRROR) { // renderMirror, shaders/tracer.cuh:396-404
            newDir = reflect3(ray_d, normal);
            numBounces += 1;
        } else if (a.p.type == GRT_NORMAL) { // renderNormal traces [t_min, t_hit] itself, shaders/tracer.cuh:406-429
            state = Terminate;
        } else if (a.p.type == GRT_GLASS) { // renderGlass, shaders/tracer.cuh:466-482
            const float n1 = 1.0003f, n2 = 1.5f;
            if (refract_dir(ray_d, normal, n2 / n1, newDir)) t_hit += kRefractionEpsShift;
            else numBounces += 1;
            seg_tmax = t_hit; // payload.t_hit carries the shifted value (shaders/tracer.cu:180)
        }
        curO = add3(ray_o, mul3s(ray_d, t_hit));
        curD = newDir;
    } else { // __miss__miss
        curO = mk3(0, 0, 0);
        curD = mk3(0, 0, 0);
        state = LastGaussianPass;
    }
}


}  // namespace grt
