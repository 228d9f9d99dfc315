// grt_api.hip — C ABI of libgrt_hip.so (include/grt.h): context, scene upload, BVH build driver,
// render entry points.  No CPU fallback: every entry point that needs the GPU fails loudly without one.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string.h>

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#include <cstring>
#include <string>
#include <vector>

#include "grt_device.h"
#include "grt_internal.h"

using namespace grt;

static thread_local std::string g_create_err;

#define CHK(ctx, x)                                                                                   \
    do {                                                                                              \
        hipError_t e_ = (x);                                                                          \
        if (e_ != hipSuccess) {                                                                       \
            (ctx)->err = std::string(#x) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"; \
            return GRT_ERR_HIP;                                                                       \
        }                                                                                             \
    } while (0)

static void free_slot_state(grt_ctx* c);
static inline grt_ctx* scene_of(grt_ctx* c) { return c->parent ? c->parent : c; }
static inline const grt_ctx* scene_of(const grt_ctx* c) { return c->parent ? c->parent : c; }
// scene calls on a view are refused: the scene belongs to the parent
#define NOT_A_VIEW(c, what)                                                                           \
    do {                                                                                              \
        if ((c)->parent) { (c)->err = what ": this context is a view; change the scene through its parent"; return GRT_ERR_INVALID; } \
    } while (0)

// ------------------------------------------------------------------------------------------------
// scene kernels
// ------------------------------------------------------------------------------------------------

// World AABB of the proxy icosahedron M = T * (R * diag(scale*s)) (src/GaussianTracer.cpp:304-311,
// src/geometry/Icosahedron.h:13-37).  opacity <= alpha_min gives s = NaN/0 in the reference, i.e. an
// unhittable instance: such particles get an inverted box and are left out of the BVH.
__global__ void k_proxy_boxes(const float* __restrict__ pos, const float* __restrict__ scale,
                              const float* __restrict__ quat, const float* __restrict__ s_arr, uint32_t n,
                              float4* __restrict__ lo, float4* __restrict__ hi)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float s = s_arr[i];
    if (!(s > 0.0f)) {
        lo[i] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
        hi[i] = make_float4(-1.0f, -1.0f, -1.0f, 0.0f);
        return;
    }
    const float rr = (3.0f + sqrtf(5.0f)) / (2.0f * sqrtf(3.0f));
    const float ss = 1.0f / rr;
    const float tt = (1.0f + sqrtf(5.0f)) / (2.0f * rr);
    const float V[12][3] = {{-ss, tt, 0}, {ss, tt, 0}, {-ss, -tt, 0}, {ss, -tt, 0}, {0, -ss, tt}, {0, ss, tt},
                            {0, -ss, -tt}, {0, ss, -tt}, {tt, 0, -ss}, {tt, 0, ss}, {-tt, 0, -ss}, {-tt, 0, ss}};
    float Rg[9];
    mat3_cast(quat[i * 4], quat[i * 4 + 1], quat[i * 4 + 2], quat[i * 4 + 3], Rg);
    const float sx = scale[i * 3] * s, sy = scale[i * 3 + 1] * s, sz = scale[i * 3 + 2] * s;
    float l[3] = {INFINITY, INFINITY, INFINITY}, h[3] = {-INFINITY, -INFINITY, -INFINITY};
    float r2 = 0.0f; // largest squared distance of a vertex from the centre
#pragma unroll
    for (int v = 0; v < 12; v++) {
        const float lx = sx * V[v][0], ly = sy * V[v][1], lz = sz * V[v][2];
        float q2 = 0.0f;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const float wl = (Rg[0 * 3 + r] * lx + Rg[1 * 3 + r] * ly) + Rg[2 * 3 + r] * lz;
            const float w = wl + pos[i * 3 + r];
            l[r] = fminf(l[r], w);
            h[r] = fmaxf(h[r], w);
            q2 += wl * wl;
        }
        r2 = fmaxf(r2, q2);
    }
    float emax = 0.0f;
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const float e = 1e-5f * (1.0f + fmaxf(fabsf(l[r]), fabsf(h[r])));
        l[r] -= e;
        h[r] += e;
        emax = fmaxf(emax, e);
    }
    // hi.w: radius of a sphere about the BOX CENTRE that holds the proxy (the vertices come in +- pairs, so the box centre is
    // the particle's position up to the rounding the box margin e covers many times over): what the tile kernel's leaf step
    // culls with besides the box, which for a round proxy reaches 1.5 x as far along an oblique plane normal
    lo[i] = make_float4(l[0], l[1], l[2], 0.0f);
    hi[i] = make_float4(h[0], h[1], h[2], sqrtf(r2) * (1.0f + 1e-5f) + 3.0f * emax);
}

// ---- spatial splits of large anisotropic proxies -------------------------------------------------------------------
// The LBVH bounds every proxy by its world AABB.  A needle or a sheet that is not axis-aligned fills a vanishing part of
// that box: a tile's thin frustum crosses thousands of such boxes without ever touching the proxies inside, and the
// traversal runs with a frontier that never clears (the C3a scene: 608 child boxes culled and 774 proxies slab-tested
// per ray for 67 composited events).  OptiX meets the same scene with an ORIENTED proxy per particle (an instance
// transform over 20 triangles, src/GaussianTracer.cpp:297-317,401-420).  Here a large proxy whose box is mostly empty
// enters the tree as several PIECES: the proxy-local box [-tt s, tt s]^3 (tt = 1.0705: the icosahedron's extent along
// its principal axes) is cut into p1 x p2 x p3 cells, each bounded by the world AABB of its cell, clipped to the
// proxy's own AABB.  Every piece refers to the WHOLE particle (the record is the particle's; the exact test is
// unchanged), the cells cover the proxy, so the piece that contains a ray's entry (exit) point is reached no later than
// that event: the hits are the same.  A ray that crosses several pieces of one particle meets it several times with
// bit-identical keys (t, id, entry/exit).  An EVENT BELONGS TO THE PIECE WHOSE CELL HOLDS ITS POINT (piece_owns,
// grt_device.h: the cell index of o_g + t d_g, from the descriptor in the record's last word), so each event is
// reported once; where the wave-per-tile kernels carry the exit with the entry a repeat is still possible (the entry
// is composited, then the exit's own piece turns up) and they drop it: an event at or before the last composited key is
// not inserted, and of equal keys that meet in a window only the first is composited.  Pure acceleration-structure
// work, as splitting is inside OptiX: pixels, hit counters and the oracle (which builds its own BVH) are untouched.
constexpr float kIcoTT = 1.0704663f; // (1 + sqrt 5) / (2 rr), rr = (3 + sqrt 5) / (2 sqrt 3): src/geometry/Icosahedron.h:15-17
constexpr uint32_t kMaxPieces = 512u;

struct PieceGrid { uint32_t p[3]; };

// how particle i is cut: pieces per principal axis (1,1,1 = not split).  tau = the piece length aimed at.
__device__ __forceinline__ PieceGrid piece_grid(const float* __restrict__ scale, const float* __restrict__ quat, float s, uint32_t i,
                                                float tau, float4 lo, float4 hi, float volf)
{
    PieceGrid g{{1u, 1u, 1u}};
    if (!(s > 0.0f) || !(tau > 0.0f)) return g;
    const float e[3] = {scale[i * 3] * s * kIcoTT, scale[i * 3 + 1] * s * kIcoTT, scale[i * 3 + 2] * s * kIcoTT};
    const float emin = fminf(e[0], fminf(e[1], e[2])), emax = fmaxf(e[0], fmaxf(e[1], e[2]));
    if (!(2.0f * emax > tau)) return g;
    float Rg[9];
    mat3_cast(quat[i * 4], quat[i * 4 + 1], quat[i * 4 + 2], quat[i * 4 + 3], Rg);
    float len = fmaxf(tau, 2.0f * emin); // cells about as long as the proxy is thick: their boxes come out compact
    uint32_t p[3];
    for (int it = 0; it < 16; it++) {
        for (int r = 0; r < 3; r++) p[r] = (uint32_t)fminf(fmaxf(ceilf(2.0f * e[r] / len), 1.0f), 32.0f); // (5 bits per axis: piece_desc)
        if (p[0] * p[1] * p[2] <= kMaxPieces) break;
        len *= 1.5f;
    }
    if (p[0] * p[1] * p[2] > kMaxPieces || p[0] * p[1] * p[2] <= 1u) return g;
    // worth it only when the cells' boxes hold much less than the proxy's box does (an axis-aligned needle gains nothing)
    float v1 = (float)(p[0] * p[1] * p[2]);
    for (int k = 0; k < 3; k++) {
        float h = 0.0f; // half-size of a cell's box along world axis k (column c of R = Rg[c*3 + k])
        for (int c = 0; c < 3; c++) h += fabsf(Rg[c * 3 + k]) * (e[c] / (float)p[c]);
        v1 *= fminf(2.0f * h, (k == 0) ? hi.x - lo.x : (k == 1) ? hi.y - lo.y : hi.z - lo.z);
    }
    const float v0 = (hi.x - lo.x) * (hi.y - lo.y) * (hi.z - lo.z);
    if (!(v1 < volf * v0)) return g;
    g.p[0] = p[0]; g.p[1] = p[1]; g.p[2] = p[2];
    return g;
}

// 64-bit sum of a 32-bit array (the piece total: the exclusive scan beside it runs in 32 bits and could wrap)
__global__ void k_sum_u32(const uint32_t* __restrict__ v, uint64_t n, unsigned long long* __restrict__ out)
{
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) acc += v[i];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63u) == 0u && acc) atomicAdd(out, acc);
}

__global__ void k_piece_counts(const float* __restrict__ scale, const float* __restrict__ quat, const float* __restrict__ s_arr,
                               const float4* __restrict__ lo, const float4* __restrict__ hi, uint32_t n, float tau, float volf,
                               uint32_t* __restrict__ counts)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const PieceGrid g = piece_grid(scale, quat, s_arr[i], i, tau, lo[i], hi[i], volf);
    counts[i] = g.p[0] * g.p[1] * g.p[2];
}

// one thread per piece: its owner by binary search in the offsets, its cell, its box
__global__ void k_piece_boxes(const float* __restrict__ pos, const float* __restrict__ scale, const float* __restrict__ quat,
                              const float* __restrict__ s_arr, const float4* __restrict__ lo, const float4* __restrict__ hi,
                              const uint32_t* __restrict__ offs, uint32_t n, uint32_t n_pieces, float tau, float volf,
                              float4* __restrict__ plo, float4* __restrict__ phi, uint32_t* __restrict__ owner,
                              uint32_t* __restrict__ desc)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_pieces) return;
    desc[j] = 0u;
    uint32_t a = 0, b = n; // largest i with offs[i] <= j
    while (b - a > 1u) {
        const uint32_t m = (a + b) >> 1;
        if (offs[m] <= j) a = m; else b = m;
    }
    const uint32_t i = a;
    owner[j] = i;
    const float4 l = lo[i], h = hi[i];
    const PieceGrid g = piece_grid(scale, quat, s_arr[i], i, tau, l, h, volf);
    if (g.p[0] * g.p[1] * g.p[2] <= 1u) { plo[j] = l; phi[j] = h; return; }
    uint32_t q = j - offs[i];
    const uint32_t k0 = q % g.p[0]; q /= g.p[0];
    const uint32_t k1 = q % g.p[1], k2 = q / g.p[1];
    const uint32_t kk[3] = {k0, k1, k2};
    desc[j] = piece_desc(kk, g.p);
    const float s = s_arr[i];
    float Rg[9];
    mat3_cast(quat[i * 4], quat[i * 4 + 1], quat[i * 4 + 2], quat[i * 4 + 3], Rg);
    float mid[3], half[3];
    for (int r = 0; r < 3; r++) {
        const float e = scale[i * 3 + r] * s * kIcoTT, w = 2.0f * e / (float)g.p[r];
        mid[r] = -e + ((float)kk[r] + 0.5f) * w;
        half[r] = 0.5f * w * (1.0f + 1e-5f) + 1e-6f * e; // the cells overlap by a hair: no point of the proxy falls between two
    }
    float bl[3], bh[3];
    const float L[3] = {l.x, l.y, l.z}, H[3] = {h.x, h.y, h.z};
    for (int k = 0; k < 3; k++) {
        float c = pos[i * 3 + k], hw = 0.0f;
        for (int r = 0; r < 3; r++) { c += Rg[r * 3 + k] * mid[r]; hw += fabsf(Rg[r * 3 + k]) * half[r]; }
        const float m = 2e-5f * (1.0f + fabsf(c) + hw); // rounding of the nine products above, and then some
        bl[k] = fmaxf(c - hw - m, L[k]);
        bh[k] = fminf(c + hw + m, H[k]);
        if (!(bl[k] <= bh[k])) { bl[k] = L[k]; bh[k] = H[k]; } // (cannot happen: the cell meets the proxy's box)
    }
    plo[j] = make_float4(bl[0], bl[1], bl[2], 0.0f);
    phi[j] = make_float4(bh[0], bh[1], bh[2], INFINITY); // (a cell has no bounding sphere worth testing: see k_proxy_boxes)
}

// sum over the hittable proxies of log(box diagonal), per workgroup (fixed order; the host adds the partials in double):
// exp(mean) is the typical proxy size the split length is a multiple of
__global__ void k_log_diag_partial(const float4* __restrict__ lo, const float4* __restrict__ hi, uint32_t n, float* __restrict__ part,
                                   uint32_t* __restrict__ cnt)
{
    float sum = 0.0f;
    uint32_t c = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 l = lo[i], h = hi[i];
        if (l.x <= h.x) {
            sum += logf(fmaxf(sqrtf((h.x - l.x) * (h.x - l.x) + (h.y - l.y) * (h.y - l.y) + (h.z - l.z) * (h.z - l.z)), 1e-30f));
            c++;
        }
    }
    for (int off = 32; off > 0; off >>= 1) { sum += __shfl_xor(sum, off); c += (uint32_t)__shfl_xor((int)c, off); }
    __shared__ float ssum[4];
    __shared__ uint32_t scnt[4];
    if ((threadIdx.x & 63) == 0) { ssum[threadIdx.x >> 6] = sum; scnt[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = ssum[0]; uint32_t k = scnt[0];
        for (uint32_t w = 1; w < (blockDim.x + 63u) / 64u; w++) { t += ssum[w]; k += scnt[w]; }
        part[blockIdx.x] = t; cnt[blockIdx.x] = k;
    }
}

// Proxy record in Morton order, 64 B = 4 x float4:
//   (mu.x mu.y mu.z s) (A00 A01 A02 opacity) (A10 A11 A12 id-bits) (A20 A21 A22 cell-bits)
// A = diag(1/scale) * R^T exactly as computeResponse forms it per hit (shaders/tracer.cuh:191-201).
// (owner / desc: piece -> particle and the piece's cell (piece_desc) when large proxies were split, else nullptr:
//  primitive = particle; the last word of the record is the cell descriptor, 0 for a whole proxy)
__global__ void k_gather_records(const float* __restrict__ pos, const float* __restrict__ scale,
                                 const float* __restrict__ quat, const float* __restrict__ opacity,
                                 const float* __restrict__ s_arr, const uint32_t* __restrict__ order,
                                 const uint32_t* __restrict__ owner, const uint32_t* __restrict__ desc, uint32_t m,
                                 float4* __restrict__ rec)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const uint32_t i = owner ? owner[order[j]] : order[j];
    const uint32_t cd = desc ? desc[order[j]] : 0u;
    float Rg[9];
    mat3_cast(quat[i * 4], quat[i * 4 + 1], quat[i * 4 + 2], quat[i * 4 + 3], Rg);
    float A[9];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const float inv = 1.0f / scale[i * 3 + r];
#pragma unroll
        for (int c = 0; c < 3; c++) A[r * 3 + c] = inv * Rg[r * 3 + c];
    }
    rec[(size_t)j * 4 + 0] = make_float4(pos[i * 3], pos[i * 3 + 1], pos[i * 3 + 2], s_arr[i]);
    rec[(size_t)j * 4 + 1] = make_float4(A[0], A[1], A[2], opacity[i]);
    rec[(size_t)j * 4 + 2] = make_float4(A[3], A[4], A[5], __uint_as_float(i));
    rec[(size_t)j * 4 + 3] = make_float4(A[6], A[7], A[8], __uint_as_float(cd));
}

// Eye records: everything in the proxy test that depends on the ray ORIGIN only.  Camera rays share one origin,
// so the streaming kernel reads these (wave-uniform, scalar loads) instead of recomputing them on every lane:
//   (o_g.x o_g.y o_g.z cc)      o_g = A (eye - mu), cc = |o_g|^2 - R^2 of the pre-test sphere
// — each by the very operation sequence the kernels use per lane (grt_device.h), so results are bit-identical.
// Rebuilt only when the eye moves or the records change: one pass over the particles (~20 us per million).
__global__ void k_eye_records(const float4* __restrict__ rec, uint32_t m, float ex, float ey, float ez,
                              float4* __restrict__ erec)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const float4 r0 = rec[(size_t)j * 4], r1 = rec[(size_t)j * 4 + 1], r2 = rec[(size_t)j * 4 + 2],
                 r3 = rec[(size_t)j * 4 + 3];
    m33 A;
    A.a[0] = r1.x; A.a[1] = r1.y; A.a[2] = r1.z;
    A.a[3] = r2.x; A.a[4] = r2.y; A.a[5] = r2.z;
    A.a[6] = r3.x; A.a[7] = r3.y; A.a[8] = r3.z;
    const f3 o_g = matvec(A, sub3(mk3(ex, ey, ez), mk3(r0.x, r0.y, r0.z)));
    erec[j] = make_float4(o_g.x, o_g.y, o_g.z, proxy_sphere_cc(o_g, r0.w));
}

// Wide eye records for the tile kernel, 64 B: the same (o_g, cc) plus the ten slab projections of o_g
// (slab_project, grt_device.h) that the exact proxy test otherwise forms on every lane from wave-uniform inputs:
//   (o_g.x o_g.y o_g.z cc) (a0 a1 a2 a3) (a4 a5 a6 a7) (a8 a9 0 0)
__global__ void k_eye_records_wide(const float4* __restrict__ rec, uint32_t m, float ex, float ey, float ez,
                                   float4* __restrict__ erec)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const float4 r0 = rec[(size_t)j * 4], r1 = rec[(size_t)j * 4 + 1], r2 = rec[(size_t)j * 4 + 2],
                 r3 = rec[(size_t)j * 4 + 3];
    m33 A;
    A.a[0] = r1.x; A.a[1] = r1.y; A.a[2] = r1.z;
    A.a[3] = r2.x; A.a[4] = r2.y; A.a[5] = r2.z;
    A.a[6] = r3.x; A.a[7] = r3.y; A.a[8] = r3.z;
    const f3 o_g = matvec(A, sub3(mk3(ex, ey, ez), mk3(r0.x, r0.y, r0.z)));
    float pa[10];
    slab_project(o_g, pa);
    float4* e = erec + (size_t)j * 4;
    e[0] = make_float4(o_g.x, o_g.y, o_g.z, proxy_sphere_cc(o_g, r0.w));
    e[1] = make_float4(pa[0], pa[1], pa[2], pa[3]);
    e[2] = make_float4(pa[4], pa[5], pa[6], pa[7]);
    e[3] = make_float4(pa[8], pa[9], 0.0f, 0.0f);
}

// degree-0 radiance max(0.5 + SH_C0 * sh[0], 0) (shaders/tracer.cuh:223,263), by original id
__global__ void k_color0(const float* __restrict__ sh, uint32_t n, float4* __restrict__ color0)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* s = sh + (size_t)i * 48;
    color0[i] = make_float4(fmaxf(0.5f + GRT_SH_C0 * s[0], 0.0f), fmaxf(0.5f + GRT_SH_C0 * s[1], 0.0f),
                            fmaxf(0.5f + GRT_SH_C0 * s[2], 0.0f), 0.0f);
}

__global__ void k_tri_boxes(const float* __restrict__ verts, const uint32_t* __restrict__ faces, uint32_t nf,
                            float4* __restrict__ lo, float4* __restrict__ hi)
{
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nf) return;
    float l[3], h[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float a = verts[faces[f * 3] * 3 + k], b = verts[faces[f * 3 + 1] * 3 + k],
                    c = verts[faces[f * 3 + 2] * 3 + k];
        l[k] = fminf(a, fminf(b, c));
        h[k] = fmaxf(a, fmaxf(b, c));
        const float e = 1e-5f * (1.0f + fmaxf(fabsf(l[k]), fabsf(h[k])));
        l[k] -= e;
        h[k] += e;
    }
    lo[f] = make_float4(l[0], l[1], l[2], 0.0f);
    hi[f] = make_float4(h[0], h[1], h[2], 0.0f);
}

__global__ void k_gather_tris(const float* __restrict__ verts, const uint32_t* __restrict__ faces,
                              const uint32_t* __restrict__ order, uint32_t m, float4* __restrict__ tri)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const uint32_t f = order[j];
    const uint32_t i0 = faces[f * 3], i1 = faces[f * 3 + 1], i2 = faces[f * 3 + 2];
    tri[(size_t)j * 3 + 0] = make_float4(verts[i0 * 3], verts[i0 * 3 + 1], verts[i0 * 3 + 2], __uint_as_float(f));
    tri[(size_t)j * 3 + 1] = make_float4(verts[i1 * 3], verts[i1 * 3 + 1], verts[i1 * 3 + 2], 0.0f);
    tri[(size_t)j * 3 + 2] = make_float4(verts[i2 * 3], verts[i2 * 3 + 1], verts[i2 * 3 + 2], 0.0f);
}

// Cold-start scheduling estimate: how many particle centres project into each 8x8 tile (scheduling unit).  Used ONLY to
// order the launch of a frame that has no previous-frame costs (first frame, camera cut, new size): dense tiles first.
// The projection inverts getRay / getFishEyeRay (shaders/tracer.cuh:115-165) for a point instead of a pixel; it never
// touches a pixel value.
// the pixel a world point projects to (the inverse of getRay / getFishEyeRay, shaders/tracer.cuh:115-165); false: behind the camera,
// outside the fisheye circle or outside the image
__device__ __forceinline__ bool project_point(const grt_params& p, f3 x, float& fx, float& fy)
{
    const f3 v = sub3(x, mk3(p.eye[0], p.eye[1], p.eye[2]));
    const f3 U = mk3(p.U[0], p.U[1], p.U[2]), V = mk3(p.V[0], p.V[1], p.V[2]), W = mk3(p.W[0], p.W[1], p.W[2]);
    // components of v in the (-U, -V, W) basis the ray generators use (the three are mutually orthogonal)
    const float su = -dot3(v, U) / fmaxf(dot3(U, U), 1e-30f), sv = -dot3(v, V) / fmaxf(dot3(V, V), 1e-30f),
                sw = dot3(v, W) / fmaxf(dot3(W, W), 1e-30f);
    float dx, dy;
    if (!p.mode_fisheye) {
        if (!(sw > 1e-6f)) return false; // behind the camera
        dx = su / sw; dy = sv / sw;
    } else {
        const float len = sqrtf(su * su + sv * sv + sw * sw);
        if (!(len > 0.0f)) return false;
        const float ct = fminf(fmaxf(sw / len, -1.0f), 1.0f);
        const float r = sqrtf(2.0f) * sqrtf(fmaxf(0.5f * (1.0f - ct), 0.0f)); // sqrt(2) sin(theta / 2)
        const float rho = sqrtf(su * su + sv * sv);
        if (!(r <= 1.0f) || !(rho > 0.0f)) return false;
        dx = r * su / rho; dy = r * sv / rho;
    }
    fx = (dx + 1.0f) * 0.5f * (float)p.width; fy = (dy + 1.0f) * 0.5f * (float)p.height;
    return fx >= 0.0f && fy >= 0.0f && fx < (float)p.width && fy < (float)p.height;
}

__global__ void k_estimate_costs(const float* __restrict__ pos, uint32_t n, uint32_t stride, const RenderArgs a,
                                 uint32_t* __restrict__ cost)
{
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) * stride; // a sample of the particles is enough for an ORDER
    if (i >= n) return;
    float fx, fy;
    if (!project_point(a.p, mk3(pos[i * 3], pos[i * 3 + 1], pos[i * 3 + 2]), fx, fy)) return;
    const uint32_t px = (uint32_t)fx, py = (uint32_t)fy;
    uint32_t blk, lx, ly;
    if (a.mode == 0) {
        if (px < a.x0 || py < a.y0 || px >= a.x1 || py >= a.y1) return;
        lx = px - a.x0; ly = py - a.y0;
        blk = (ly / 16u) * a.nbx + lx / 16u;
    } else {
        const uint32_t tile = (py / a.tile_h) * a.tiles_x + px / a.tile_w;
        if (tile < a.first_tile || (tile - a.first_tile) % a.tile_stride) return;
        const uint32_t j = (tile - a.first_tile) / a.tile_stride;
        if (j >= a.n_tiles) return;
        lx = px % a.tile_w; ly = py % a.tile_h;
        blk = j * (a.nbx * a.nby) + (ly / 16u) * a.nbx + lx / 16u;
    }
    atomicAdd(&cost[blk * 4u + (((ly % 16u) / 8u) << 1) + ((lx % 16u) / 8u)], 1u);
}

// FNV-1a over the face indices of the meshes in the order given (grt_update_meshes checks the topology with it)
static uint64_t faces_hash(uint64_t h, const uint32_t* f, size_t n)
{
    if (h == 0) h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= f[i]; h *= 1099511628211ull; }
    return h;
}

// The tile kernel's give-up reasons travel in the per-tile cost words (see the watchdog in grt_render_tile.hip): a cost
// above the step watchdog = the watchdog fired, high bits = stack guard / two passes without progress.  One pass over
// the costs right behind the frame ORs them into the context's sticky error word.
// behind every frame (do_launch): error word and overflow demand to their pinned host words, overflow counter reset
// (d_ovf_next[0] = chunks this frame asked for, [1] = the largest demand since the host last took one: the pinned word is written
//  only when the host is not waiting for an earlier value — h_ovf_used == nullptr otherwise — so a peak between two reads is kept)
__global__ void k_frame_tail(const uint32_t* __restrict__ d_err, uint32_t* __restrict__ h_err, uint32_t* __restrict__ d_ovf_next,
                             uint32_t* __restrict__ h_ovf_used, const uint32_t* __restrict__ d_qpcount, uint32_t* __restrict__ h_qpcount)
{
    *h_err = *d_err;
    if (d_qpcount) *h_qpcount = d_qpcount[0]; // how many four-way parts the quad kernel's list holds (a launch that knows it is none skips that kernel)
    if (d_ovf_next) {
        const uint32_t peak = max(d_ovf_next[1], d_ovf_next[0]);
        d_ovf_next[0] = 0u;
        d_ovf_next[1] = h_ovf_used ? 0u : peak;
        if (h_ovf_used) *h_ovf_used = peak;
    }
    __threadfence_system();
}

__global__ void k_check_costs(const uint32_t* __restrict__ cost, uint32_t n, uint32_t max_iters, uint32_t* __restrict__ err_word)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = cost[i];
    uint32_t e = 0;
    if (c & kCostStackBit) e |= kErrStack;
    if (c & kCostStallBit) e |= kErrStall;
    if ((c & kCostStepsMask) > max_iters) e |= kErrWatchdog; // (bits 27-28: the tile ran as parts, grt_internal.h)
    if (e) atomicOr(err_word, e);
}

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
extern "C" {

int grt_create(grt_ctx** out, int device)
{
    if (!out) return GRT_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_err = "grt_create: no HIP device available (" + std::string(e != hipSuccess ? hipGetErrorString(e) : "0 devices") +
                       "); libgrt_hip has no CPU fallback";
        return GRT_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) {
        g_create_err = "grt_create: device index out of range";
        return GRT_ERR_INVALID;
    }
    if ((e = hipSetDevice(device)) != hipSuccess) {
        g_create_err = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return GRT_ERR_HIP;
    }
    grt_ctx* c = new grt_ctx();
    c->device = device;
    if ((e = hipStreamCreate(&c->stream)) != hipSuccess || (e = hipEventCreate(&c->ev0)) != hipSuccess ||
        (e = hipEventCreate(&c->ev1)) != hipSuccess ||
        (e = hipMalloc(&c->d_counters, kNumCounters * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipStreamCreate(&c->aux_stream)) != hipSuccess || (e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming)) != hipSuccess ||
        (e = hipMalloc(&c->d_n_heavy, sizeof(uint32_t))) != hipSuccess ||
        (e = hipMalloc(&c->d_err, sizeof(uint32_t))) != hipSuccess || (e = hipMemset(c->d_err, 0, sizeof(uint32_t))) != hipSuccess ||
        (e = hipHostMalloc(&c->h_ovf_used, sizeof(uint32_t), hipHostMallocDefault)) != hipSuccess ||
        (e = hipHostMalloc(&c->h_err, 2 * sizeof(uint32_t), hipHostMallocDefault)) != hipSuccess || // [0] error word, [1] parts in the quad list
        (e = hipEventCreateWithFlags(&c->ev_ovf, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming)) != hipSuccess) {
        g_create_err = std::string("grt_create: ") + hipGetErrorString(e);
        free_slot_state(c);
        delete c;
        return GRT_ERR_HIP;
    }
    *c->h_ovf_used = 0;
    *c->h_err = 0;
    *out = c;
    return GRT_OK;
}

// A second frame slot on the SAME scene: its own eye records, scheduling feedback, overflow pool, queues, counters and
// events, the parent's Gaussians / BVHs / meshes.  What D frames in flight need (bench.py, a double-buffering viewer)
// without D scene replicas.  Scene calls (upload, build, meshes) go to the parent; the parent must outlive its use by
// the view's renders (grt_destroy of a parent with live views is deferred until the last of them is destroyed).
int grt_create_view(grt_ctx* parent, grt_ctx** out)
{
    if (!out) return GRT_ERR_INVALID;
    *out = nullptr;
    if (!parent || parent->parent) {
        g_create_err = "grt_create_view: the parent must be a context made by grt_create";
        return GRT_ERR_INVALID;
    }
    grt_ctx* v = nullptr;
    const int rc = grt_create(&v, parent->device);
    if (rc != GRT_OK) return rc;
    v->parent = parent;
    {
        std::lock_guard<std::mutex> lk(parent->views_mu);
        parent->n_views++;
        parent->views.push_back(v);
    }
    *out = v;
    return GRT_OK;
}

static void free_gaussians(grt_ctx* c)
{
    (void)hipFree(c->d_pos); (void)hipFree(c->d_scale); (void)hipFree(c->d_quat); (void)hipFree(c->d_opacity);
    (void)hipFree(c->d_sh); (void)hipFree(c->d_color0);
    c->d_pos = c->d_scale = c->d_quat = c->d_opacity = c->d_sh = nullptr;
    c->d_color0 = nullptr;
    c->n = 0;
    c->built = false;
}

static void free_meshes(grt_ctx* c)
{
    (void)hipFree(c->d_tri); (void)hipFree(c->d_faces); (void)hipFree(c->d_vnormals);
    c->d_tri = nullptr; c->d_faces = nullptr; c->d_vnormals = nullptr;
    c->n_faces = c->n_verts = 0;
    free_bvh(&c->mbvh);
}

// everything a frame slot owns (a view has nothing else)
static void reap_old_pools(grt_ctx* c, bool force, hipStream_t s);
static void free_slot_state(grt_ctx* c)
{
    (void)hipFree(c->d_erec); (void)hipFree(c->d_erec_wide);
    (void)hipFree(c->d_counters);
    (void)hipFree(c->d_cost); (void)hipFree(c->d_order); (void)hipFree(c->d_cost_dil); (void)hipFree(c->d_ord_scratch); (void)hipFree(c->d_qparts); (void)hipFree(c->d_qpcount);
    (void)hipFree(c->d_prec); (void)hipFree(c->d_queue); (void)hipFree(c->d_qcount); (void)hipFree(c->d_heavy); (void)hipFree(c->d_fqueue);
    (void)hipFree(c->d_bverdict); (void)hipFree(c->d_qunit); (void)hipFree(c->d_qskip); (void)hipFree(c->d_heavy_a);
    reap_old_pools(c, true, nullptr);
    (void)hipFree(c->d_ovf); (void)hipFree(c->d_ovf_next);
    (void)hipFree(c->d_err);
    if (c->h_ovf_used) (void)hipHostFree(c->h_ovf_used);
    if (c->h_err) (void)hipHostFree(c->h_err);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_ovf) (void)hipEventDestroy(c->ev_ovf);
    if (c->ev_tail) (void)hipEventDestroy(c->ev_tail);
    (void)hipFree(c->d_n_heavy);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
}

static void destroy_now(grt_ctx* c)
{
    if (!c->parent) {
        free_gaussians(c);
        free_meshes(c);
        free_bvh(&c->gbvh);
        (void)hipFree(c->d_rec);
    }
    free_slot_state(c);
    delete c;
}

void grt_destroy(grt_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize(); // renders of this slot (and, for a scene, of its views) may be in flight on any stream
    if (c->parent) {
        grt_ctx* p = c->parent;
        bool last;
        {   // (a sibling's launch may be looking at this view's frame-end event: off the list first, under the lock, then destroyed)
            std::lock_guard<std::mutex> lk(p->views_mu);
            p->views.erase(std::remove(p->views.begin(), p->views.end(), c), p->views.end());
            last = --p->n_views == 0;
        }
        destroy_now(c);
        if (last && p->zombie) destroy_now(p);
        return;
    }
    if (c->n_views > 0) { c->zombie = true; return; } // its views still render this scene
    destroy_now(c);
}

const char* grt_last_error(const grt_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

int grt_set_option(grt_ctx* c, int option, int value)
{
    if (!c) return GRT_ERR_INVALID;
    c->order_ready = false; // prepared with the old options
    if (option == GRT_OPT_COUNTERS) c->opt_counters = value ? 1 : 0;
    else if (option == GRT_OPT_KERNEL) {
        if (value < 0 || value > GRT_KERNEL_MAX) { c->err = "GRT_OPT_KERNEL must be 0.." + std::to_string(GRT_KERNEL_MAX); return GRT_ERR_INVALID; }
        c->opt_kernel = value;
        c->cost_valid = false; // scheduling units differ between kernels
    }
    else if (option == GRT_OPT_FEEDBACK) { c->opt_feedback = value ? 1 : 0; c->opt_heavy_split = (value & 4) ? 1 : ((value & 2) ? 0 : 2); c->cost_valid = false;
                                             c->bv_epoch = ~0ull; /* (what the frames before taught — costs, and the bundle verdicts of mesh frames — is forgotten) */ }
    else if (option == GRT_OPT_HEAVY_THRESHOLD_X2) { c->opt_heavy_thr_x2 = std::max(2, value); }
    else if (option == GRT_OPT_HEAVY_CAP_DIV) { c->opt_heavy_cap_div = std::max(1, value); }
    else if (option == GRT_OPT_SWIZZLE) {
        if (value < 0) { c->err = "GRT_OPT_SWIZZLE must be >= 0"; return GRT_ERR_INVALID; }
        c->opt_swizzle = value;
    }
    else if (option == GRT_OPT_TILE_READY_MIN) { c->opt_tile_ready = std::min(64, std::max(1, value)); }
    else if (option == GRT_OPT_TILE_BAND) { c->opt_tile_band = std::max(0, value); }
    else if (option == GRT_OPT_TILE_LOOKAHEAD) { c->opt_tile_look = std::max(0, value); }
    else if (option == GRT_OPT_COLD_ESTIMATE) { c->opt_cold_estimate = std::min(2, std::max(0, value)); c->cost_valid = false; }
    else if (option == GRT_OPT_COLD_PARTS_PCT) { c->opt_cold_parts_pct = std::max(0, value); c->cost_valid = false; }
    else if (option == GRT_OPT_BUNDLE_ROUNDS) {
        if (value < 0 || value > kMaxBundleRounds) { c->err = "GRT_OPT_BUNDLE_ROUNDS must be 0.." + std::to_string(kMaxBundleRounds); return GRT_ERR_INVALID; }
        c->opt_bundle_rounds = value;
    }
    else if (option == GRT_OPT_BUNDLE_BUDGET) { c->opt_bundle_budget = std::max(1, value); c->bv_epoch = ~0ull; /* (a verdict is a statement about THIS budget) */ }
    else if (option == GRT_OPT_LANE_BUDGET) { c->opt_lane_budget = std::max(1, value); }
    else if (option == GRT_OPT_MESH_PRIMARY_WAVE) { c->opt_mesh_primary_wave = value < 0 ? 0 : (value > 2 ? 2 : value); }
    else if (option == GRT_OPT_BUNDLE_PREDICT) { c->opt_bundle_predict = value ? 1 : 0; c->bv_epoch = ~0ull; /* (verdicts start afresh) */ }
    else if (option == GRT_OPT_SINGLE_LOOKAHEAD) { c->opt_single_look = std::max(0, value); }
    else if (option == GRT_OPT_SINGLE_BAND) { c->opt_single_band = std::max(0, value); }
    else if (option == GRT_OPT_SIZE_CLASSES) { NOT_A_VIEW(c, "GRT_OPT_SIZE_CLASSES"); c->opt_size_classes = value ? 1 : 0; }
    else if (option == GRT_OPT_BVH_ROTATIONS) { NOT_A_VIEW(c, "GRT_OPT_BVH_ROTATIONS"); c->opt_bvh_rotations = value < 0 ? -1 : (value > 8 ? 8 : value); }
    else if (option == GRT_OPT_SPLIT_VOL_PCT) { NOT_A_VIEW(c, "GRT_OPT_SPLIT_VOL_PCT"); c->opt_split_vol_pct = std::max(1, value); }
    else if (option == GRT_OPT_SPLIT) { NOT_A_VIEW(c, "GRT_OPT_SPLIT"); c->opt_split = value < 0 ? -1 : std::min(1024, value); }
    else if (option == GRT_OPT_TILE_BAND_ABS) { c->opt_band_abs = std::max(0, value); }
    else if (option == GRT_OPT_OVF_CHUNKS) { c->opt_ovf_chunks = value; c->ovf_demand = 0; c->ovf_hist_n = 0; c->ovf_short = false; c->ovf_sized = false; }
    else if (option == GRT_OPT_OVF_ENTRIES) {
        if (value < 0 || value > (int)kTileOvfEntries) { c->err = "GRT_OPT_OVF_ENTRIES must be 0.." + std::to_string(kTileOvfEntries); return GRT_ERR_INVALID; }
        c->opt_ovf_entries = value;
    }
    else if (option == GRT_OPT_MAX_ITERS) { // (the steps travel in 27 bits of the tile's cost word: a limit the word cannot exceed could never be reported)
        if (value < 0 || (uint32_t)value > kCostStepsMask - 1u) { c->err = "GRT_OPT_MAX_ITERS: 0 (default) .. 2^27 - 2"; return GRT_ERR_INVALID; }
        c->opt_max_iters = value;
    }
    else if (option == GRT_OPT_COST_RADIUS) { c->opt_cost_radius = std::min(8, std::max(0, value)); }
    else if (option == GRT_OPT_TILE_PARTS2_PCT) { c->opt_tile_parts2_pct = std::min(100, std::max(0, value)); c->cost_valid = false; c->order_ready = false; }
    else if (option == GRT_OPT_TILE_PARTS4_PCT) { c->opt_tile_parts4_pct = std::min(100, std::max(0, value)); c->cost_valid = false; c->order_ready = false; }
    else if (option == GRT_OPT_STATIC_SHARP) { c->opt_static_sharp = value != 0; c->cost_valid = false; c->order_ready = false; }
    else if (option == GRT_OPT_ORDER_MULTI_MIN) { c->opt_order_multi_min = std::max(1, value); }
    else if (option == GRT_OPT_MESH_PARTS) { c->opt_mesh_parts = value != 0; c->cost_valid = false; c->order_ready = false; }
    else if (option == GRT_OPT_OVF_CLASSES) { c->opt_ovf_classes = value ? 1 : 0; c->cost_valid = false; c->order_ready = false; }
    else if (option == GRT_OPT_QUAD_PARTS) { c->opt_quad_parts = std::max(0, value); c->cost_valid = false; c->order_ready = false; } // (2: whatever the launch's size; > 2: and that many parts at most — testing)
    else if (option == GRT_OPT_TILE_PARTS_LOAD_PCT) { c->opt_tile_parts_load_pct = std::min(100000, std::max(0, value)); c->cost_valid = false; c->order_ready = false; }
    else if (option == GRT_OPT_TILE_PRIO_DIV) { c->opt_tile_prio = std::max(0, value); }
    else if (option == GRT_OPT_TILE_RESERVE) { c->opt_tile_reserve = std::min(63, std::max(-1, value)); }
    else if (option == GRT_OPT_LEAF_MAX) {
        if (value < 1 || value > (int)kLeafMaxPrims) { c->err = "GRT_OPT_LEAF_MAX must be 1..8"; return GRT_ERR_INVALID; }
        NOT_A_VIEW(c, "GRT_OPT_LEAF_MAX");
        c->opt_leaf_max = value; // takes effect at the next grt_build_bvh / grt_set_meshes
    }
    else { c->err = "grt_set_option: unknown option"; return GRT_ERR_INVALID; }
    return GRT_OK;
}

// ------------------------------------------------------------------------------------------------
// scene
// ------------------------------------------------------------------------------------------------
int grt_upload_gaussians(grt_ctx* c, const grt_gaussians* g, uint64_t n)
{
    if (c) c->scene_epoch++;
    if (!c || (n && (!g || !g->pos || !g->scale || !g->quat || !g->opacity || !g->sh))) {
        if (c) c->err = "grt_upload_gaussians: null argument";
        return GRT_ERR_INVALID;
    }
    if (n >= (1ull << 26)) { c->err = "grt_upload_gaussians: more than 2^26-1 particles (hit keys carry a 26-bit id)"; return GRT_ERR_LIMIT; }
    NOT_A_VIEW(c, "grt_upload_gaussians");
    CHK(c, hipSetDevice(c->device));
    CHK(c, hipDeviceSynchronize()); // frames of this scene may be in flight on any stream (caller's, views')
    free_gaussians(c);
    c->h_opacity.assign(g ? g->opacity : nullptr, g ? g->opacity + n : nullptr);
    if (n == 0) return GRT_OK;
    CHK(c, hipMalloc(&c->d_pos, n * 3 * sizeof(float)));
    CHK(c, hipMalloc(&c->d_scale, n * 3 * sizeof(float)));
    CHK(c, hipMalloc(&c->d_quat, n * 4 * sizeof(float)));
    CHK(c, hipMalloc(&c->d_opacity, n * sizeof(float)));
    CHK(c, hipMalloc(&c->d_sh, n * 48 * sizeof(float)));
    CHK(c, hipMalloc(&c->d_color0, n * sizeof(float4)));
    CHK(c, hipMemcpyAsync(c->d_pos, g->pos, n * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    CHK(c, hipMemcpyAsync(c->d_scale, g->scale, n * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    CHK(c, hipMemcpyAsync(c->d_quat, g->quat, n * 4 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    CHK(c, hipMemcpyAsync(c->d_opacity, g->opacity, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    CHK(c, hipMemcpyAsync(c->d_sh, g->sh, n * 48 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_color0, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->d_sh, (uint32_t)n,
                       c->d_color0);
    CHK(c, hipGetLastError());
    CHK(c, hipStreamSynchronize(c->stream)); // host arrays are only borrowed for the call
    c->n = n;
    return GRT_OK;
}

int grt_build_bvh(grt_ctx* c, float alpha_min)
{
    if (c) c->scene_epoch++;
    if (!c) return GRT_ERR_INVALID;
    if (!(alpha_min > 0.0f)) { c->err = "grt_build_bvh: alpha_min must be > 0"; return GRT_ERR_INVALID; }
    NOT_A_VIEW(c, "grt_build_bvh");
    CHK(c, hipSetDevice(c->device));
    CHK(c, hipDeviceSynchronize());
    c->built = false;
    c->alpha_min = alpha_min;
    const uint32_t n = (uint32_t)c->n;
    c->gbvh.n_prims = 0;
    c->gbvh.root_ref = kNoRoot;
    c->gbvh.height = 0;
    c->has_pieces = false;
    if (n == 0) { c->built = true; return GRT_OK; }
    // proxy half-width s = sqrtf(2 logf(opacity/alpha_min)) on the HOST, as the reference does
    // (src/GaussianTracer.cpp:306) — keeps the libm-dependent value identical to the host libm's.
    std::vector<float> s(n);
    for (uint32_t i = 0; i < n; i++) s[i] = sqrtf(2.0f * logf(c->h_opacity[i] / alpha_min));
    float* d_s = nullptr;
    float4 *d_lo = nullptr, *d_hi = nullptr;
    int rc = GRT_OK;
    hipError_t e;
    if ((e = hipMalloc(&d_s, n * sizeof(float))) != hipSuccess || (e = hipMalloc(&d_lo, n * sizeof(float4))) != hipSuccess ||
        (e = hipMalloc(&d_hi, n * sizeof(float4))) != hipSuccess) {
        c->err = std::string("grt_build_bvh: hipMalloc: ") + hipGetErrorString(e);
        rc = GRT_ERR_HIP;
    }
    float4 *d_plo = nullptr, *d_phi = nullptr; // piece boxes (when large proxies are split)
    uint32_t *d_cnt = nullptr, *d_offs = nullptr, *d_owner = nullptr, *d_desc = nullptr;
    void* d_scan_tmp = nullptr;
    uint32_t n_pieces = n;
    c->n_hittable = 0;
    for (uint32_t i = 0; i < n; i++) c->n_hittable += (s[i] > 0.0f) ? 1u : 0u;
    if (rc == GRT_OK) {
        (void)hipMemcpyAsync(d_s, s.data(), n * sizeof(float), hipMemcpyHostToDevice, c->stream);
        (void)hipEventRecord(c->ev0, c->stream);
        hipLaunchKernelGGL(k_proxy_boxes, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->d_pos, c->d_scale, c->d_quat,
                           d_s, n, d_lo, d_hi);
    }
    if (rc == GRT_OK && c->opt_split != 0 && n > 1) {
        // ---- spatial splits (see k_piece_boxes): piece length = opt_split/4 x the geometric-mean proxy diagonal ----
        float* d_part = nullptr;
        uint32_t* d_pcnt = nullptr;
        const int grid = (int)std::min<uint32_t>((n + 255) / 256, 256u);
        std::vector<float> h_part(grid);
        std::vector<uint32_t> h_pcnt(grid);
        if ((e = hipMalloc(&d_part, grid * sizeof(float))) != hipSuccess || (e = hipMalloc(&d_pcnt, grid * sizeof(uint32_t))) != hipSuccess ||
            (e = hipMalloc(&d_cnt, (size_t)n * sizeof(uint32_t))) != hipSuccess || (e = hipMalloc(&d_offs, (size_t)n * sizeof(uint32_t))) != hipSuccess) {
            c->err = std::string("grt_build_bvh: hipMalloc(split): ") + hipGetErrorString(e);
            rc = GRT_ERR_HIP;
        }
        float tau = 0.0f;
        if (rc == GRT_OK) {
            hipLaunchKernelGGL(k_log_diag_partial, dim3(grid), dim3(256), 0, c->stream, d_lo, d_hi, n, d_part, d_pcnt);
            (void)hipMemcpyAsync(h_part.data(), d_part, grid * sizeof(float), hipMemcpyDeviceToHost, c->stream);
            (void)hipMemcpyAsync(h_pcnt.data(), d_pcnt, grid * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
            if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) { c->err = std::string("grt_build_bvh: ") + hipGetErrorString(e); rc = GRT_ERR_HIP; }
        }
        if (rc == GRT_OK) {
            double sum = 0.0; uint64_t cnt = 0;
            for (int b = 0; b < grid; b++) { sum += h_part[b]; cnt += h_pcnt[b]; }
            if (cnt) { c->gm_diag = (float)std::exp(sum / (double)cnt); tau = 0.25f * (float)(c->opt_split < 0 ? 8 : c->opt_split) * c->gm_diag; }
        }
        // pieces at a given piece length: per-proxy counts (d_cnt), their exclusive scan (d_offs) and the total, summed in 64 bits (the scan runs in
        // 32 bits: 512 pieces per particle times 2^26 particles could wrap it; a scene whose pieces would not fit the leaf index keeps whole proxies)
        const float volf = 0.01f * (float)c->opt_split_vol_pct;
        auto count_pieces = [&](float tau_, uint64_t& total_) -> int {
            hipLaunchKernelGGL(k_piece_counts, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->d_scale, c->d_quat, d_s, d_lo, d_hi, n, tau_, volf, d_cnt);
            if (device_exclusive_scan_u32(d_cnt, d_offs, n, c->stream, &c->err) != GRT_OK) return GRT_ERR_HIP;
            unsigned long long* d_tot = nullptr;
            unsigned long long h_tot = 0;
            hipError_t e_ = hipMalloc(&d_tot, sizeof(*d_tot));
            if (e_ == hipSuccess) e_ = hipMemsetAsync(d_tot, 0, sizeof(*d_tot), c->stream);
            if (e_ == hipSuccess) {
                hipLaunchKernelGGL(k_sum_u32, dim3(std::min<uint32_t>((uint32_t)((n + 255) / 256), 1024u)), dim3(256), 0, c->stream, d_cnt, n, d_tot);
                if ((e_ = hipMemcpyAsync(&h_tot, d_tot, sizeof(h_tot), hipMemcpyDeviceToHost, c->stream)) == hipSuccess) e_ = hipStreamSynchronize(c->stream);
            }
            (void)hipFree(d_tot);
            if (e_ != hipSuccess) { c->err = std::string("grt_build_bvh: piece total: ") + hipGetErrorString(e_); return GRT_ERR_HIP; }
            total_ = h_tot;
            return GRT_OK;
        };
        if (rc == GRT_OK && tau > 0.0f) {
            uint64_t total = 0;
            rc = count_pieces(tau, total);
            // GRT_OPT_SPLIT < 0 (default): the piece length follows the scene.  Measured on the 1 M scene with per-axis log-scale noise sigma,
            // every proxy longer than the piece length cut (GRT_OPT_SPLIT_VOL_PCT = 400), kernel ms by length in quarters of the typical
            // diagonal (profiles/r06_experiments_log.md 9): sigma 0.7: 2.29 (6) 2.28 (8) 2.57 (12); 0.85: 2.46 (6) 2.50 (8) 2.63 (10); 1.0: 2.97 (6)
            // 3.00 (8) 3.45 (12); 1.2: 4.28 (6) 4.20 (8) 4.42 (10); 1.4: 8.31 (6) 7.67 (8) 7.60 (10) 8.07 (12); 1.6: 9.12 (8) 8.75 (10) 8.62 (12) 9.06
            // (16); 2.0: 70 (8) 59 (12) 53 (16).  Mildly anisotropic proxies want SHORT pieces, scene-sized needles long ones (every piece
            // re-tests its particle).  The primitives per proxy that cutting at 8 gives tell the scenes apart (1.10 / 1.16 / 1.23 / 1.37 /
            // 1.56 / 1.82 / 2.62 for the sigmas above): under 1.25 -> 6, under 1.5 -> 8, under 1.7 -> 10, under 2.2 -> 12, else 16; and a scene
            // that would gain under 2 % of primitives keeps whole proxies and the kernels without the piece logic (the benchmark scenes
            // C1-C5: cut finer they only lose, 1.78 -> 1.86 ms at 3 % of pieces).
            if (rc == GRT_OK && c->opt_split < 0) {
                const double r8 = (double)total / (double)n;
                const int q = r8 < 1.02 ? 8 : (r8 < 1.25 ? 6 : (r8 < 1.5 ? 8 : (r8 < 1.7 ? 10 : (r8 < 2.2 ? 12 : 16))));
                if (q != 8) {
                    tau = 0.25f * (float)q * c->gm_diag;
                    rc = count_pieces(tau, total);
                }
                c->split_used = q;
            } else {
                c->split_used = c->opt_split;
            }
            // (a scene where splitting adds less than 2 % of primitives has no population of needles and sheets to speak of:
            //  it keeps whole proxies — size classes deal with the odd large one — and the kernels without the piece logic)
            if (rc == GRT_OK && total > (uint64_t)n + n / 50u && total <= (uint64_t)kLeafIndexMask) {
                n_pieces = (uint32_t)total;
                if ((e = hipMalloc(&d_plo, (size_t)n_pieces * sizeof(float4))) != hipSuccess || (e = hipMalloc(&d_phi, (size_t)n_pieces * sizeof(float4))) != hipSuccess ||
                    (e = hipMalloc(&d_owner, (size_t)n_pieces * sizeof(uint32_t))) != hipSuccess ||
                    (e = hipMalloc(&d_desc, (size_t)n_pieces * sizeof(uint32_t))) != hipSuccess) {
                    c->err = std::string("grt_build_bvh: hipMalloc(pieces): ") + hipGetErrorString(e);
                    rc = GRT_ERR_HIP;
                } else {
                    hipLaunchKernelGGL(k_piece_boxes, dim3((n_pieces + 255) / 256), dim3(256), 0, c->stream, c->d_pos, c->d_scale, c->d_quat, d_s,
                                       d_lo, d_hi, d_offs, n, n_pieces, tau, volf, d_plo, d_phi, d_owner, d_desc);
                }
            }
        }
        (void)hipFree(d_part); (void)hipFree(d_pcnt);
    }
    if (rc == GRT_OK)
        rc = build_lbvh(d_owner ? d_plo : d_lo, d_owner ? d_phi : d_hi, d_owner ? n_pieces : n, (uint32_t)c->opt_leaf_max, true, false,
                        c->opt_size_classes, &c->gbvh, c->stream, &c->err, d_owner != nullptr, c->opt_bvh_rotations);
    if (rc == GRT_OK && c->gbvh.n_prims) {
        const uint32_t m = c->gbvh.n_prims;
        if (c->cap_rec < m) {
            (void)hipFree(c->d_rec);
            c->d_rec = nullptr;
            c->cap_rec = 0;
            if ((e = hipMalloc(&c->d_rec, (size_t)m * 4 * sizeof(float4) + 256)) != hipSuccess) {
                c->err = std::string("grt_build_bvh: hipMalloc(rec): ") + hipGetErrorString(e);
                rc = GRT_ERR_HIP;
            } else c->cap_rec = m;
        }
        if (rc == GRT_OK)
            hipLaunchKernelGGL(k_gather_records, dim3((m + 255) / 256), dim3(256), 0, c->stream, c->d_pos, c->d_scale,
                               c->d_quat, c->d_opacity, d_s, c->gbvh.order, d_owner, d_owner ? d_desc : nullptr, m, c->d_rec);
    }
    if (rc == GRT_OK) {
        (void)hipEventRecord(c->ev1, c->stream);
        e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) { c->err = std::string("grt_build_bvh: ") + hipGetErrorString(e); rc = GRT_ERR_HIP; }
        else (void)hipEventElapsedTime(&c->build_ms, c->ev0, c->ev1);
    }
    (void)hipFree(d_s); (void)hipFree(d_lo); (void)hipFree(d_hi);
    const bool have_pieces = d_owner != nullptr;
    (void)hipFree(d_plo); (void)hipFree(d_phi); (void)hipFree(d_cnt); (void)hipFree(d_offs); (void)hipFree(d_owner); (void)hipFree(d_desc); (void)hipFree(d_scan_tmp);
    c->have_timing = false;
    c->cost_valid = false;
    c->erec_valid = false;
    if (rc == GRT_OK) { c->built = true; c->built_leaf_max = c->opt_leaf_max; c->has_pieces = have_pieces; }
    return rc;
}

int grt_set_meshes(grt_ctx* c, const grt_mesh* meshes, uint32_t n_meshes)
{
    if (c) c->scene_epoch++;
    if (!c || (n_meshes && !meshes)) return GRT_ERR_INVALID;
    NOT_A_VIEW(c, "grt_set_meshes");
    CHK(c, hipSetDevice(c->device));
    CHK(c, hipDeviceSynchronize());
    free_meshes(c);
    c->mesh_nv.clear(); c->mesh_nf.clear(); c->faces_hash = 0;
    // Flatten every primitive into one world-space soup; face indices are offset per mesh in the order
    // given (reference: one instance per primitive, instanceId = order of creation,
    // src/GaussianTracer.cpp:592-593; lowest (mesh, face) wins exact-t ties).
    std::vector<float> v, nrm;
    std::vector<uint32_t> f;
    for (uint32_t k = 0; k < n_meshes; k++) {
        const grt_mesh& m = meshes[k];
        if ((m.nv && (!m.verts || !m.normals)) || (m.nf && !m.faces)) { c->err = "grt_set_meshes: null array"; return GRT_ERR_INVALID; }
        const uint32_t base = (uint32_t)(v.size() / 3);
        c->mesh_nv.push_back(m.nv); c->mesh_nf.push_back(m.nf);
        c->faces_hash = faces_hash(c->faces_hash, m.faces, (size_t)m.nf * 3);
        v.insert(v.end(), m.verts, m.verts + (size_t)m.nv * 3);
        nrm.insert(nrm.end(), m.normals, m.normals + (size_t)m.nv * 3);
        for (size_t i = 0; i < (size_t)m.nf * 3; i++) {
            if (m.faces[i] >= m.nv) { c->err = "grt_set_meshes: face index out of range"; return GRT_ERR_INVALID; }
            f.push_back(m.faces[i] + base);
        }
    }
    const uint32_t nf = (uint32_t)(f.size() / 3), nv = (uint32_t)(v.size() / 3);
    if (nf == 0) return GRT_OK;
    float* d_verts = nullptr;
    float4 *d_lo = nullptr, *d_hi = nullptr;
    int rc = GRT_OK;
    hipError_t e = hipSuccess;
    // every failure falls through to the common clean-up below (temporaries freed, mesh state reset)
    if ((e = hipMalloc(&d_verts, (size_t)nv * 3 * sizeof(float))) != hipSuccess ||
        (e = hipMalloc(&c->d_vnormals, (size_t)nv * 3 * sizeof(float))) != hipSuccess ||
        (e = hipMalloc(&c->d_faces, (size_t)nf * 3 * sizeof(uint32_t))) != hipSuccess ||
        (e = hipMalloc(&d_lo, (size_t)nf * sizeof(float4))) != hipSuccess ||
        (e = hipMalloc(&d_hi, (size_t)nf * sizeof(float4))) != hipSuccess ||
        (e = hipMalloc(&c->d_tri, (size_t)nf * 3 * sizeof(float4))) != hipSuccess ||
        (e = hipMemcpyAsync(d_verts, v.data(), (size_t)nv * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream)) != hipSuccess ||
        (e = hipMemcpyAsync(c->d_vnormals, nrm.data(), (size_t)nv * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream)) != hipSuccess ||
        (e = hipMemcpyAsync(c->d_faces, f.data(), (size_t)nf * 3 * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream)) != hipSuccess) {
        c->err = std::string("grt_set_meshes: ") + hipGetErrorString(e);
        rc = GRT_ERR_HIP;
    }
    if (rc == GRT_OK) {
        (void)hipEventRecord(c->ev0, c->stream);
        hipLaunchKernelGGL(k_tri_boxes, dim3((nf + 255) / 256), dim3(256), 0, c->stream, d_verts, c->d_faces, nf, d_lo, d_hi);
        rc = build_lbvh(d_lo, d_hi, nf, (uint32_t)c->opt_leaf_max, false, true, 0, &c->mbvh, c->stream, &c->err);
    }
    if (rc == GRT_OK) {
        hipLaunchKernelGGL(k_gather_tris, dim3((nf + 255) / 256), dim3(256), 0, c->stream, d_verts, c->d_faces,
                           c->mbvh.order, nf, c->d_tri);
        (void)hipEventRecord(c->ev1, c->stream);
        e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) { c->err = std::string("grt_set_meshes: ") + hipGetErrorString(e); rc = GRT_ERR_HIP; }
        else (void)hipEventElapsedTime(&c->mesh_update_ms, c->ev0, c->ev1);
    } else {
        (void)hipStreamSynchronize(c->stream); // host vectors are borrowed by the async copies
    }
    (void)hipFree(d_verts); (void)hipFree(d_lo); (void)hipFree(d_hi);
    if (rc != GRT_OK) { free_meshes(c); return rc; }
    c->have_timing = false;
    c->n_faces = nf;
    c->n_verts = nv;
    return GRT_OK;
}

// Same meshes, moved (a gizmo drag: reference updateInstanceTransforms, src/GaussianTracer.cpp:711-736, rebuilds GAS and
// IAS every time and leaks the old ones): new vertex positions / normals for the SAME topology; the mesh LBVH keeps its
// hierarchy and only re-fits its boxes.  Fails with GRT_ERR_INVALID when the counts differ from the last grt_set_meshes.
int grt_update_meshes(grt_ctx* c, const grt_mesh* meshes, uint32_t n_meshes)
{
    if (c) c->scene_epoch++;
    if (!c || (n_meshes && !meshes)) return GRT_ERR_INVALID;
    NOT_A_VIEW(c, "grt_update_meshes");
    CHK(c, hipSetDevice(c->device));
    // the topology must be the one grt_set_meshes built the tree for: per mesh, not just in total (two meshes that swap
    // sizes keep the sums), and the same indices (only positions / normals may move)
    bool same = n_meshes == c->mesh_nv.size() && c->n_faces != 0;
    for (uint32_t k = 0; same && k < n_meshes; k++) same = meshes[k].nv == c->mesh_nv[k] && meshes[k].nf == c->mesh_nf[k];
    if (!same) {
        c->err = "grt_update_meshes: mesh count or per-mesh vertex / face counts differ from the last grt_set_meshes (call that instead)";
        return GRT_ERR_INVALID;
    }
    uint64_t nv = 0, nf = 0, fh = 0;
    for (uint32_t k = 0; k < n_meshes; k++) {
        const grt_mesh& m = meshes[k];
        if ((m.nv && (!m.verts || !m.normals)) || (m.nf && !m.faces)) { c->err = "grt_update_meshes: null array"; return GRT_ERR_INVALID; }
        fh = faces_hash(fh, m.faces, (size_t)m.nf * 3);
        nv += m.nv; nf += m.nf;
    }
    if (fh != c->faces_hash) {
        c->err = "grt_update_meshes: face indices differ from the last grt_set_meshes (call that instead)";
        return GRT_ERR_INVALID;
    }
    std::vector<float> v, nrm;
    for (uint32_t k = 0; k < n_meshes; k++) {
        const grt_mesh& m = meshes[k];
        v.insert(v.end(), m.verts, m.verts + (size_t)m.nv * 3);
        nrm.insert(nrm.end(), m.normals, m.normals + (size_t)m.nv * 3);
    }
    // the node boxes, triangles and normals are overwritten in place: no frame may still be reading them, on whatever
    // stream it was launched (renders are asynchronous on the caller's stream; views have streams of their own)
    CHK(c, hipDeviceSynchronize());
    float* d_verts = nullptr;
    float4 *d_lo = nullptr, *d_hi = nullptr;
    int rc = GRT_OK;
    hipError_t e = hipSuccess;
    if ((e = hipMalloc(&d_verts, (size_t)nv * 3 * sizeof(float))) != hipSuccess ||
        (e = hipMalloc(&d_lo, (size_t)nf * sizeof(float4))) != hipSuccess ||
        (e = hipMalloc(&d_hi, (size_t)nf * sizeof(float4))) != hipSuccess ||
        (e = hipMemcpyAsync(d_verts, v.data(), (size_t)nv * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream)) != hipSuccess ||
        (e = hipMemcpyAsync(c->d_vnormals, nrm.data(), (size_t)nv * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream)) != hipSuccess) {
        c->err = std::string("grt_update_meshes: ") + hipGetErrorString(e);
        rc = GRT_ERR_HIP;
    }
    if (rc == GRT_OK) {
        (void)hipEventRecord(c->ev0, c->stream);
        hipLaunchKernelGGL(k_tri_boxes, dim3(((uint32_t)nf + 255) / 256), dim3(256), 0, c->stream, d_verts, c->d_faces, (uint32_t)nf, d_lo, d_hi);
        rc = refit_lbvh(d_lo, d_hi, (uint32_t)nf, &c->mbvh, c->stream, &c->err);
    }
    if (rc == GRT_OK) {
        hipLaunchKernelGGL(k_gather_tris, dim3(((uint32_t)nf + 255) / 256), dim3(256), 0, c->stream, d_verts, c->d_faces,
                           c->mbvh.order, (uint32_t)nf, c->d_tri);
        (void)hipEventRecord(c->ev1, c->stream);
        e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) { c->err = std::string("grt_update_meshes: ") + hipGetErrorString(e); rc = GRT_ERR_HIP; }
        else (void)hipEventElapsedTime(&c->mesh_update_ms, c->ev0, c->ev1);
    } else {
        (void)hipStreamSynchronize(c->stream);
    }
    (void)hipFree(d_verts); (void)hipFree(d_lo); (void)hipFree(d_hi);
    c->have_timing = false;
    return rc;
}

int grt_get_bvh_info(const grt_ctx* c, grt_bvh_info* o)
{
    if (!c || !o) return GRT_ERR_INVALID;
    const grt_ctx* sc = scene_of(c);
    memset(o, 0, sizeof(*o));
    o->n_particles = sc->n;
    o->n_proxies = sc->n_hittable;
    o->n_primitives = sc->gbvh.n_prims;
    o->n_nodes = sc->gbvh.n_prims ? sc->gbvh.n_prims - 1 : 0;
    o->height = sc->gbvh.height;
    o->mesh_faces = sc->n_faces;
    o->mesh_height = sc->mbvh.height;
    o->build_ms = sc->build_ms;
    o->mesh_update_ms = sc->mesh_update_ms;
    for (int k = 0; k < 3; k++) { o->scene_lo[k] = sc->gbvh.lo[k]; o->scene_hi[k] = sc->gbvh.hi[k]; }
    return GRT_OK;
}

// (testing) The depth of the Gaussian LBVH as a traversal meets it, WALKED on the host over a copy of the binary node records — independent
// of the level bookkeeping of the build, whose root level grt_get_bvh_info reports as `height` and the kernels size their stacks by.
int grt_debug_bvh_depth(grt_ctx* c, uint32_t* out_depth)
{
    if (!c || !out_depth) return GRT_ERR_INVALID;
    grt_ctx* sc = scene_of(c);
    *out_depth = 0;
    const uint32_t m = sc->gbvh.n_prims;
    if (m == 0 || (sc->gbvh.root_ref & kLeafBit) || !sc->gbvh.nodes) return GRT_OK;
    (void)hipSetDevice(sc->device);
    std::vector<float4> h((size_t)(m - 1) * 4);
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(h.data(), sc->gbvh.nodes, h.size() * sizeof(float4), hipMemcpyDeviceToHost) != hipSuccess) {
        c->err = "grt_debug_bvh_depth: copy of the node records failed";
        return GRT_ERR_HIP;
    }
    std::vector<std::pair<uint32_t, uint32_t>> stack; // (node, depth counted in internal nodes)
    stack.emplace_back(0u, 1u);
    uint64_t visited = 0;
    while (!stack.empty()) {
        const auto [i, d] = stack.back();
        stack.pop_back();
        if (i >= m - 1 || ++visited > (uint64_t)m) { c->err = "grt_debug_bvh_depth: the node records are not a tree"; return GRT_ERR_LIMIT; }
        *out_depth = std::max(*out_depth, d);
        uint32_t ch[2];
        memcpy(&ch[0], &h[(size_t)i * 4 + 3].x, 4);
        memcpy(&ch[1], &h[(size_t)i * 4 + 3].y, 4);
        for (int k = 0; k < 2; k++)
            if (!(ch[k] & kLeafBit)) stack.emplace_back(ch[k], d + 1u);
    }
    return GRT_OK;
}

// Device memory held: by the scene this context renders (shared by a context and its views) and by this frame slot.
int grt_get_memory_info(const grt_ctx* c, grt_memory_info* o)
{
    if (!c || !o) return GRT_ERR_INVALID;
    const grt_ctx* sc = scene_of(c);
    memset(o, 0, sizeof(*o));
    const uint64_t n = sc->n, m = sc->gbvh.n_prims, nf = sc->n_faces, nv = sc->n_verts;
    uint64_t b = n * (3 + 3 + 4 + 1 + 48) * 4 + n * 16;                     // raw attributes + color0
    b += sc->cap_rec * 64;                                                    // proxy records
    b += sc->gbvh.cap_nodes * 64 + sc->gbvh.cap_order * 4;                    // binary nodes, order
    if (sc->gbvh.wnodes && m > 1) b += (m - 1) * 128;                         // 4-wide view
    if (sc->gbvh.qnodes && m > 1) b += (m - 1) * 32ull * kTileWide;           // 8-wide per-child view
    if (sc->gbvh.pbox) b += m * 32;
    b += nf * (48 + 12) + nv * 12 + sc->mbvh.cap_nodes * 64 + sc->mbvh.cap_order * 4 + (sc->mbvh.wnodes && nf > 1 ? (nf - 1) * 128 : 0) +
         (sc->mbvh.level && nf > 1 ? (nf - 1) * 4 : 0);
    o->scene_bytes = b;
    uint64_t v = (uint64_t)c->cap_erec * 16 + (uint64_t)c->cap_erec_wide * 64;
    v += (uint64_t)c->ovf_chunks * kTileOvfChunkBytes;
    v += (uint64_t)c->cost_cap * 12;
    v += (uint64_t)c->wf_cap * (48 + 128 + 4 + 64);
    o->slot_bytes = v;
    o->overflow_pool_bytes = (uint64_t)c->ovf_chunks * kTileOvfChunkBytes;
    o->overflow_chunks = c->ovf_chunks;
    o->overflow_demand = c->ovf_demand;
    return GRT_OK;
}

// ------------------------------------------------------------------------------------------------
// render
// ------------------------------------------------------------------------------------------------
static int fill_common(grt_ctx* c, const grt_params* p, RenderArgs* a)
{
    if (!c || !p) return GRT_ERR_INVALID;
    const grt_ctx* sc = scene_of(c);
    if (!sc->built) { c->err = "render: grt_build_bvh has not been called after the last upload"; return GRT_ERR_INVALID; }
    if (p->sh_degree_max > 3) { c->err = "render: sh_degree_max must be 0..3"; return GRT_ERR_INVALID; }
    if (p->type < 0 || p->type > 2) { c->err = "render: type must be MIRROR/NORMAL/GLASS"; return GRT_ERR_INVALID; }
    if (!(p->t_min > 0.0f)) { c->err = "render: t_min must be > 0"; return GRT_ERR_INVALID; }
    memset(a, 0, sizeof(*a));
    a->p = *p;
    a->rec = sc->d_rec;
    a->nodes = sc->gbvh.nodes;
    a->wnodes = sc->gbvh.wnodes;
    a->qnodes = sc->gbvh.qnodes;
    a->pbox = sc->gbvh.pbox;
    a->root_ref = sc->gbvh.root_ref;
    a->n_prox = sc->gbvh.n_prims;
    a->has_pieces = sc->has_pieces ? 1u : 0u; // set by the build that made pieces (not inferred from counts: build_lbvh drops NaN boxes)
    a->color0 = sc->d_color0;
    a->sh = sc->d_sh;
    a->mnodes = sc->mbvh.nodes;
    a->tri = sc->d_tri;
    a->mroot = sc->n_faces ? sc->mbvh.root_ref : kNoRoot;
    a->n_faces = sc->n_faces;
    a->faces = sc->d_faces;
    a->vnormals = sc->d_vnormals;
    a->counters = c->opt_counters ? c->d_counters : nullptr;
    a->swizzle_chunk = (uint32_t)c->opt_swizzle;
    a->err_word = c->d_err;
    a->max_iters = c->opt_max_iters > 0 ? (uint32_t)c->opt_max_iters : kTileMaxItersDefault;
    return GRT_OK;
}

// Frame-to-frame scheduling feedback: every render records each 16x16 block's cost (the largest number of
// traversal iterations among its waves); the next render with the same frame geometry launches the blocks
// heaviest-first, so the long-running tiles (10x the mean on the benchmark scenes) no longer form the tail of the
// launch.  Pure scheduling: pixels do not depend on the order.  A viewer's consecutive frames are nearly identical,
// which is what makes last frame's cost a good predictor; the first frame (or any change of size / mode) runs in
// the default XCD-chunked order.
// launch order of the units from the costs the last frame left in d_cost (dilated for full-frame launches)
// launch entries beyond one per tile (the parts of split tiles): a quarter of the tiles — and, for a launch that does not fill the
// machine, whatever fills it (a 256^2 frame is 1024 tiles on 4096 wave slots, its tiles' costs lie close together, and with room for
// 106 split tiles the 107th, whole, bounded the frame)
static uint32_t parts_extra_cap(uint32_t n_units)
{
    const uint32_t base = n_units / 4u + 64u;
    return n_units < kTileResidentWaves ? std::max(base, std::min(3u * n_units, kTileResidentWaves - n_units)) : base;
}
// four-way parts on the quad kernel: camera rays without meshes or pieces (that kernel has no mesh stage and no piece bookkeeping; such
// frames keep part waves of the camera-ray kernel)
// — and launches that the parts of their heaviest tiles BOUND: up to three times the machine's resident waves (measured,
// profiles/r05_experiments_log.md 4: a 256^2 frame, 1024 tiles, 0.46 -> 0.27 ms; a rank's share of a 1080p frame, 4050 / 8100 tiles,
// 0.77 -> 0.55 / 0.75 -> 0.61 ms; a 720p frame of 14 400 tiles is bound by its total work and loses 5 % to the second kernel).
// GRT_OPT_QUAD_PARTS = 2 forces it on whatever the size (tests).
// is a frame of another slot of this scene still running?  (its frame-end event, recorded on its stream behind its last kernels)
static bool sibling_frames_in_flight(const grt_ctx* c)
{
    grt_ctx* sc = const_cast<grt_ctx*>(scene_of(c));
    std::lock_guard<std::mutex> lk(sc->views_mu); // (views come and go on other threads: grt_create_view / grt_destroy take the same lock)
    if (sc->n_views == 0) return false;
    auto busy = [&](const grt_ctx* s) { return s != c && s->have_timing && s->ev1 && hipEventQuery(s->ev1) == hipErrorNotReady; };
    if (busy(sc)) return true;
    for (const grt_ctx* v : sc->views) if (busy(v)) return true;
    return false;
}
static bool quad_parts_ok(const grt_ctx* c, uint32_t n_units)
{
    const grt_ctx* sc = scene_of(c);
    if (!c->opt_quad_parts || sc->n_faces || sc->has_pieces) return false;
    // (while frames of the scene's OTHER slots are in flight the machine is shared, the frames are bound by their total work and the quad
    //  kernel — a latency tool — only takes capacity: a rank of 8 with eight frames in flight 0.26 -> 0.33 ms per frame with it)
    return c->opt_quad_parts >= 2 || (n_units <= 3u * kTileResidentWaves && !sibling_frames_in_flight(c));
}
// the four-way threshold such launches use: parts on the quad kernel cost a third of what part waves of the camera-ray kernel cost, so
// more tiles are worth splitting the fewer tiles there are per resident wave — pct4 at two tiles per wave, half of it at one and below
static uint32_t quad_pct4(const grt_ctx* c, uint32_t n_units)
{
    const uint32_t p = (uint32_t)c->opt_tile_parts4_pct;
    return std::min(p, std::max(p / 2u, (uint32_t)((uint64_t)p * n_units / (2u * kTileResidentWaves))));
}

static int order_from_costs(grt_ctx* c, const RenderArgs& a, uint32_t n_units, hipStream_t s, bool* used_split, bool zero_costs = false)
{
    uint32_t* d_zero = zero_costs ? c->d_cost : nullptr; // (the ordering kernel zeroes the consumed costs itself: one packet less)
    const bool split = c->opt_heavy_split == 1 || (c->opt_heavy_split == 2 && a.n_blocks <= 3072u);
    const uint32_t* cost_src = c->d_cost;
    // (the dilation is for a camera that MOVES: a heavy tile of the last frame is a slightly different tile of the next.  When this
    //  frame ran with an order that was made for exactly this view — the camera stood still for two frames — the next order is made
    //  from the tiles' own costs: a dilated map orders a standing view worse, C2 0.86 -> 0.80 ms, C3 -1.3 %; if the camera then moves,
    //  one frame runs with an undilated order)
    //  (camera-ray frames without meshes: on C4 the undilated order costs 2 % — the stages behind the primary one take the tiles'
    //   continuation rays in the order the primary waves finish)
    const bool dilate = c->opt_cost_radius > 0 && !(c->launch_order_matched && c->opt_static_sharp && !scene_of(c)->n_faces);
    if (dilate && a.mode == 0 && n_units == a.n_blocks * 4u && !split) {
        int rcd = dilate_unit_costs(c->d_cost, c->d_cost_dil, a.nbx, a.nby, c->opt_cost_radius, s, &c->err);
        if (rcd != GRT_OK) return rcd;
        cost_src = c->d_cost_dil;
    }
    c->order_launch = 0;
    if (c->parts_ok && (c->opt_tile_parts2_pct > 0 || c->opt_tile_parts4_pct > 0) && n_units == a.n_blocks * 4u) {
        // tile kernel, camera rays, no meshes: the heaviest tiles of this frame run as 2 / 4 waves in the next one
        *used_split = false;
        const uint32_t cap = parts_extra_cap(n_units);
        const bool quad = quad_parts_ok(c, n_units) && c->d_qparts;
        int rcp = order_units_with_parts(cost_src, c->d_cost, c->d_order, n_units, cap, (uint32_t)c->opt_tile_parts2_pct,
                                         quad ? quad_pct4(c, n_units) : (uint32_t)c->opt_tile_parts4_pct, (uint32_t)c->opt_tile_parts_load_pct, kTileResidentWaves, d_zero, c->d_ord_scratch,
                                         (uint32_t)c->opt_order_multi_min, c->opt_ovf_classes ? 1u : 0u, s, &c->err);
        c->order_classes = rcp == GRT_OK && c->opt_ovf_classes != 0;
        if (rcp == GRT_OK) {
            c->order_launch = n_units + cap;
            c->qparts_valid = false;
            if (quad) { // the four-way parts as a list for the quad kernel (one more small kernel behind the ordering)
                rcp = quad_part_list(c->d_order, c->order_launch, c->d_qparts, c->d_qpcount, (uint32_t)c->opt_quad_parts > 2u ? (uint32_t)c->opt_quad_parts : kQuadListCap, s, &c->err);
                c->qparts_valid = rcp == GRT_OK;
                c->qlist_epoch++;
            }
        }
        return rcp;
    }
    *used_split = split;
    c->order_classes = false; // (entries of this order are bare unit numbers)
    return order_units_by_cost(cost_src, c->d_order, n_units, std::max(1u, n_units / (uint32_t)c->opt_heavy_cap_div),
                               (uint32_t)c->opt_heavy_thr_x2, split ? c->d_n_heavy : nullptr, d_zero, s, &c->err);
}

static int prepare_feedback(grt_ctx* c, RenderArgs& a, hipStream_t s, uint32_t n_units, bool need_cost)
{
    a.order = nullptr;
    a.n_launch = 0;
    a.cost = nullptr;
    a.n_heavy = nullptr;
    a.heavy_role = 0;
    a.n_units = n_units;
    if ((!c->opt_feedback && !need_cost) || n_units == 0) return GRT_OK;
    const uint64_t sig[6] = {a.mode | ((uint64_t)n_units << 8), a.n_blocks, ((uint64_t)a.p.width << 32) | a.p.height,
                             ((uint64_t)a.x0 << 48) ^ ((uint64_t)a.y0 << 32) ^ ((uint64_t)a.x1 << 16) ^ a.y1,
                             ((uint64_t)a.first_tile << 32) | a.tile_stride, ((uint64_t)a.tile_w << 32) | a.tile_h};
    if (c->cost_cap < n_units) {
        (void)hipFree(c->d_cost); (void)hipFree(c->d_order); (void)hipFree(c->d_cost_dil); (void)hipFree(c->d_qparts);
        c->d_cost = c->d_order = c->d_cost_dil = c->d_qparts = nullptr;
        c->qparts_valid = false;
        c->cost_cap = 0;
        c->cost_valid = false;
        CHK(c, hipMalloc(&c->d_cost, sizeof(uint32_t) * n_units));
        CHK(c, hipMalloc(&c->d_order, sizeof(uint32_t) * ((size_t)n_units + parts_extra_cap(n_units) + 4u))); // (+ 3 diagnostic words)
        // (every entry starts as padding: an entry the ordering kernels ever failed to write would make its wave exit instead of
        //  indexing costs, queues and pixels with whatever the allocation held — the likely cause of round 4's one unexplained abort,
        //  profiles/r05_experiments_log.md 1)
        // (on the FRAME'S stream, as everything below: a null-stream hipMemset is not ordered against a non-blocking stream and
        //  may land after the kernels of this very frame have written the array)
        CHK(c, hipMemsetAsync(c->d_order, 0xFF, sizeof(uint32_t) * ((size_t)n_units + parts_extra_cap(n_units) + 4u), s));
        if (!c->d_ord_scratch) { // counts and cursors of the several-workgroup ordering (grt_bvh.hip: k_ord_a); zero once, phase C keeps it so
            CHK(c, hipMalloc(&c->d_ord_scratch, order_scratch_bytes()));
            CHK(c, hipMemsetAsync(c->d_ord_scratch, 0, order_scratch_bytes(), s));
        }
        CHK(c, hipMalloc(&c->d_cost_dil, sizeof(uint32_t) * n_units));
        CHK(c, hipMalloc(&c->d_qparts, sizeof(uint32_t) * kQuadListCap));
        if (!c->d_qpcount) {
            CHK(c, hipMalloc(&c->d_qpcount, 2 * sizeof(uint32_t)));
            CHK(c, hipMemsetAsync(c->d_qpcount, 0, 2 * sizeof(uint32_t), s));
        }
        c->cost_cap = n_units;
    }
    if (!c->opt_feedback) { // no scheduling feedback: the cost words are only collected for k_check_costs (tile kernel)
        CHK(c, hipMemsetAsync(c->d_cost, 0, sizeof(uint32_t) * n_units, s));
        a.cost = c->d_cost;
        c->cost_valid = false;
        c->order_ready = false;
        c->cost_zeroed = false;
        return GRT_OK;
    }
    const bool same = c->cost_valid && memcmp(sig, c->cost_sig, sizeof(sig)) == 0;
    const grt_ctx* sc = scene_of(c);
    // was the order this frame is launched with made for THIS frame (same scene, camera, options)?  Remembered for the order that
    // will be made from this frame's costs: costs measured under another view's part waves — its heavy tiles split, this view's not —
    // make an order that is one feedback step short of the fixed point (a 256^2 frame after a camera move: 0.61 instead of 0.53 ms,
    // for as long as the view stands still), so such an order is used but not KEPT: the next identical frame collects once more.
    const bool order_is_for_this_frame = c->order_valid && c->order_epoch == sc->scene_epoch && memcmp(&c->order_params, &a.p, sizeof(grt_params)) == 0;
    c->launch_order_matched = same && order_is_for_this_frame; // (a frame without usable costs runs in the cold order: no match)
    if (same && c->order_ready && c->order_settled && order_is_for_this_frame) {
        // the very frame the order was made from: a tile's cost does not depend on the launch order (nor, once settled, on which
        // tiles run as parts), so this frame would measure the same costs and make the same order again — keep it, collect nothing
        a.order = c->d_order;
        a.n_launch = c->order_launch;
        if (c->order_split) a.n_heavy = c->d_n_heavy;
        return GRT_OK;
    }
    if (same) {
        // (normally already there: do_launch orders the units for the next frame right behind this frame's kernels,
        //  where it fills the gap between two frames instead of delaying the next one)
        if (!c->order_ready && !c->cost_zeroed) { // the costs of the last frame are still in d_cost
            int rc = order_from_costs(c, a, n_units, s, &c->order_split);
            if (rc != GRT_OK) return rc;
            c->order_valid = true;
        }
        // (an option was changed after the costs were consumed and zeroed: the order made from them is still the best
        //  there is — this frame collects costs again)
        if (c->order_valid) {
            a.order = c->d_order;
            a.n_launch = c->order_launch;
            if (c->order_split) a.n_heavy = c->d_n_heavy;
        }
    } else if (c->opt_cold_estimate && n_units == a.n_blocks * 4u && sc->n && (a.mode == 0 || a.mode == 1)) {
        // no costs of a previous frame with this geometry: order the tiles by the number of particle centres that
        // project into them (dense tiles first), so that the first frame's long tiles do not start last
        CHK(c, hipMemsetAsync(c->d_cost_dil, 0, sizeof(uint32_t) * n_units, s));
        const uint32_t stride = sc->n > 400000 ? 4u : 1u; // original (unsorted) order: every 4th particle is a fair sample
        const uint32_t ns = ((uint32_t)sc->n + stride - 1) / stride;
        hipLaunchKernelGGL(k_estimate_costs, dim3((ns + 255) / 256), dim3(256), 0, s, sc->d_pos, (uint32_t)sc->n, stride, a, c->d_cost_dil);
        const uint32_t* src = c->d_cost_dil;
        if (a.mode == 0) { // proxies are a few tiles wide: a tile next to a dense one is heavy too
            int rcd = dilate_unit_costs(c->d_cost_dil, c->d_cost, a.nbx, a.nby, 1, s, &c->err);
            if (rcd != GRT_OK) return rcd;
            src = c->d_cost;
        }
        c->order_launch = 0;
        int rc;
        if (c->opt_cold_estimate >= 2 && c->parts_ok && a.mode == 0 && c->opt_tile_parts4_pct > 0 && !sc->n_faces) {
            // (not on mesh frames: a split tile queues four thin bundles of continuation rays — C4's cold frame 3.30 -> 3.90 ms)
            // (GRT_OPT_COLD_ESTIMATE = 2, the default since round 4: the estimate also decides which tiles of the cold frame run as part
            //  waves — above GRT_OPT_COLD_PARTS_PCT % of the heaviest estimate: C1's cold frame 0.637 -> 0.553 ms, C2 1.048 -> 1.012, C3 -1.3 %)
            const uint32_t cap = parts_extra_cap(n_units);
            rc = order_units_with_parts(src, c->d_cost_dil, c->d_order, n_units, cap, 0u, (uint32_t)c->opt_cold_parts_pct,
                                        (uint32_t)c->opt_tile_parts_load_pct, kTileResidentWaves, nullptr, c->d_ord_scratch,
                                        (uint32_t)c->opt_order_multi_min, c->opt_ovf_classes ? 2u /* (particle counts, not cost words) */ : 0u, s, &c->err);
            c->order_classes = rc == GRT_OK && c->opt_ovf_classes != 0; // (every whole tile "not known": it starts in one chunk)
            if (rc == GRT_OK) {
                c->order_launch = n_units + cap; a.n_launch = c->order_launch;
                c->qparts_valid = false;
                if (quad_parts_ok(c, n_units) && c->d_qparts) {
                    rc = quad_part_list(c->d_order, c->order_launch, c->d_qparts, c->d_qpcount, (uint32_t)c->opt_quad_parts > 2u ? (uint32_t)c->opt_quad_parts : kQuadListCap, s, &c->err);
                    c->qparts_valid = rc == GRT_OK;
                    c->qlist_epoch++;
                }
            }
        } else {
            c->order_classes = false;
            rc = order_units_by_cost(src, c->d_order, n_units, 1u, (uint32_t)c->opt_heavy_thr_x2, nullptr, nullptr, s, &c->err);
        }
        if (rc != GRT_OK) return rc;
        a.order = c->d_order;
        c->order_valid = true;
        c->order_split = false;
    } else {
        c->order_valid = false; // nothing in d_order is meant for this launch geometry
    }
    if (!(same && c->cost_zeroed)) CHK(c, hipMemsetAsync(c->d_cost, 0, sizeof(uint32_t) * n_units, s));
    a.cost = c->d_cost;
    memcpy(c->cost_sig, sig, sizeof(sig));
    c->cost_valid = true;
    c->order_ready = false;
    c->cost_zeroed = false;
    return GRT_OK;
}

// The tile kernel's pool of window-overflow bags, in chunks of kTileOvfChunkBytes = 32 KiB (32 entries x 64 lanes): a tile that
// overflows takes one to three in a row, by how deep its bags got in the frame before (grt_render_tile.hip kSub).  A tile that
// finds the pool empty falls back to another pass (never wrong; a pool for a quarter of the tiles ran dry on the default 1 M scene and
// cost that frame 12 %, and on the needle scene C3a a pool of 3/8 of the tiles made the first frames 2.3 x slower).  Round 2 held a full
// bag for EVERY tile of the launch for good (3.1 GB at 1080p, 12.4 GB at 4K, per frame slot); rounds 3-4 sized the pool from the
// largest demand ever seen, in whole 96-entry bags (1.3 GB at 1080p).  Now the pool follows the DEMAND both ways: every frame's chunk
// counter is read back behind it (pinned word, no sync; the device keeps the peak between two reads) and the last eight readings are
// kept.  The pool is 1.25 x their MEDIAN + 64 (one cold frame — a camera cut: no size classes, every tile asks for more — does not move
// it); it GROWS when the median comes within 10 % of it and SHRINKS when it is more than 1.2 x what the LARGEST of the eight would ask
// for (so a spike only delays a shrink, and a demand that wanders does not re-make the pool every frame).  A re-size does not wait for
// the device: the new pool is allocated beside the old one, which is freed once the frames that may use it have drained (hipFree
// synchronises: it is called when nothing of this scene is in flight, or at the latest when a third pool would pile up).  The first
// frame of a launch geometry still gets three chunks per tile (or what a sibling frame slot of the same scene has learnt).  An
// allocation that fails is retried at half the size down to nothing and not asked for again: rendering never fails for want of an
// optimisation buffer.
static void overflow_demand_stats(const grt_ctx* c, uint32_t* median, uint32_t* largest)
{
    *median = *largest = 0u;
    if (!c->ovf_hist_n) return;
    uint32_t v[8];
    const uint32_t n = std::min(c->ovf_hist_n, 8u);
    for (uint32_t i = 0; i < n; i++) v[i] = c->ovf_hist[i];
    std::sort(v, v + n);
    *median = v[n / 2u]; // (the upper median of an even count)
    *largest = v[n - 1u];
}

// old pools whose last users may still run: freed when their event has passed and no frame of the scene is in flight (force: now)
static void reap_old_pools(grt_ctx* c, bool force, hipStream_t s = nullptr)
{
    if (c->ovf_old.empty()) return;
    // (hipFree waits for the whole device: not while frames of this slot are queued on its stream — a loop that queues its frames
    //  without waiting would drain, and the head of such a loop would pay for the free — nor while a sibling slot has frames in flight; a
    //  third pool piling up is freed regardless)
    (void)s;
    if (!force && c->ovf_old.size() < 3u && (sibling_frames_in_flight(c) || c->ovf_idle_run < 2u)) return;
    for (size_t i = 0; i < c->ovf_old.size();) {
        auto& o = c->ovf_old[i];
        if (force || c->ovf_old.size() >= 3u || hipEventQuery(o.second) == hipSuccess) {
            if (hipEventQuery(o.second) != hipSuccess) (void)hipEventSynchronize(o.second);
            (void)hipFree(o.first);
            (void)hipEventDestroy(o.second);
            c->ovf_old.erase(c->ovf_old.begin() + (long)i);
        } else {
            i++;
        }
    }
}

static int size_overflow_pool(grt_ctx* c, uint32_t n_tiles, hipStream_t s)
{
    grt_ctx* sc = scene_of(c);
    // (does the application wait for its frames?  Two frames in a row that find their stream idle: the first frame behind a synchronisation
    //  point may be the head of a loop that does not wait)
    c->ovf_idle_run = (hipStreamQuery(s) == hipSuccess) ? std::min(c->ovf_idle_run + 1u, 8u) : 0u;
    (void)hipGetLastError();
    reap_old_pools(c, false, s);
    if (c->ovf_units != n_tiles) { c->ovf_hist_n = 0; c->ovf_demand = 0; c->ovf_short = false; c->ovf_sized = false; c->ovf_units = n_tiles; c->ovf_stale = c->ovf_pending; } // another launch geometry: start over
    if (c->ovf_pending && hipEventQuery(c->ev_ovf) == hipSuccess) {
        if (!c->ovf_stale) { // (a reading asked for under the geometry before says nothing about this one)
            if (*c->h_ovf_used != 0u) { // (0: nothing but frames that do not count since the last reading)
                c->ovf_hist[c->ovf_hist_n % 8u] = *c->h_ovf_used;
                c->ovf_hist_n++;
                overflow_demand_stats(c, &c->ovf_demand, &c->ovf_demand_max);
                sc->ovf_hint_units = c->ovf_units;
                sc->ovf_hint = c->ovf_demand;
            }
        }
        c->ovf_stale = false;
        c->ovf_pending = false;
    }
    const uint32_t cap = (uint32_t)((16ull << 30) / kTileOvfChunkBytes);
    const uint32_t most = (uint32_t)std::min<uint64_t>(cap, std::max<uint64_t>((uint64_t)n_tiles * kTileOvfChunksPerTile, 1u));
    uint32_t want;
    bool resize;
    if (c->opt_ovf_chunks != 0) {
        want = c->opt_ovf_chunks > 0 ? (uint32_t)c->opt_ovf_chunks : 0u;
        resize = c->ovf_chunks != want;
    } else {
        const bool own = c->ovf_demand != 0u;
        const uint32_t d = own ? c->ovf_demand : ((sc->ovf_hint_units == n_tiles) ? sc->ovf_hint : 0u);
        const uint32_t dmax = own ? c->ovf_demand_max : d;
        if (d == 0u) { // nothing known yet
            want = most;
            resize = c->ovf_chunks < want && !c->ovf_short;
        } else {
                want = std::min(most, d + d / 4u + 64u);
            const uint32_t keep = std::min(most, dmax + dmax / 4u + 64u); // what the largest recent demand would ask for
            // WHEN: a re-size costs milliseconds of host time (hipMalloc of half a gigabyte: ~5 ms), during which a loop that queues its
            // frames without waiting runs dry.  So the pool is re-made only at a frame the application waited for (nothing queued on
            // the frame's stream) — except the first sizing from a known demand (the pool still holds three chunks for every tile: it
            // happens in the first frames of a view) and a pool that ran DRY (the last frame asked for more than there is)
            const bool idle = c->ovf_idle_run >= 2u;
            const uint32_t last = c->ovf_hist_n ? c->ovf_hist[(c->ovf_hist_n - 1u) % 8u] : d;
            const bool first = !c->ovf_sized;
            const bool grow = c->ovf_chunks < std::min(most, d + d / 10u) && !c->ovf_short && (idle || first || last > c->ovf_chunks);
            const bool shrink = c->ovf_chunks > keep + keep / 5u && (idle || first);
            resize = grow || shrink;
            if (shrink && !grow) want = keep;
            if (resize) c->ovf_sized = true;
        }
    }
    if (resize) {
        if (c->d_ovf) { // frames in flight may still use it: it goes when they have drained
            hipEvent_t ev = nullptr;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess && hipEventRecord(ev, s) == hipSuccess) {
                c->ovf_old.emplace_back(c->d_ovf, ev);
            } else {
                if (ev) (void)hipEventDestroy(ev);
                (void)hipGetLastError();
                (void)hipDeviceSynchronize();
                (void)hipFree(c->d_ovf);
            }
        }
        c->d_ovf = nullptr;
        c->ovf_chunks = 0;
        for (uint32_t n = want; n >= 1u; n /= 2u) {
            if (hipMalloc(&c->d_ovf, (size_t)n * kTileOvfChunkBytes) == hipSuccess) { c->ovf_chunks = n; break; }
            c->ovf_short = true; // (the memory is not there: growing is not tried again for this launch geometry)
            (void)hipGetLastError(); // out of memory is not an error of the frame: a smaller pool, or none
            c->d_ovf = nullptr;
            reap_old_pools(c, true, nullptr); // (what waits to be freed may be what is missing)
            if (c->opt_ovf_chunks > 0) break;
        }
    }
    return GRT_OK;
}

static int do_launch(grt_ctx* c, RenderArgs& a, void* stream)
{
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    grt_ctx* sc = scene_of(c);
    CHK(c, hipSetDevice(c->device));
    if (c->seen_epoch != sc->scene_epoch) { // the scene changed under this slot (a view learns of it here)
        c->erec_valid = false;
        c->cost_valid = false;
        c->order_ready = false;
        c->seen_epoch = sc->scene_epoch;
    }
    // the zeroing / ordering queued behind the last frame ran on THAT frame's stream: a launch on another stream waits
    // for it (else it could start before its counters are reset — two tiles taking the same overflow chunk)
    if (c->tail_pending && c->tail_stream != s) CHK(c, hipStreamWaitEvent(s, c->ev_tail, 0));
    // (has the work behind the last frame finished?  then the pinned words it wrote are current: the error word, and how many parts
    //  the quad kernel's list holds)
    const bool tail_done = !c->tail_pending || hipEventQuery(c->ev_tail) == hipSuccess;
    if (tail_done && c->tail_qepoch != ~0ull) { c->qknown_epoch = c->tail_qepoch; c->qknown_count = c->h_err[1]; } // (the list that tail saw, and its length)
    c->tail_pending = false;
    {
        // the streaming kernel runs one 8x8 tile (one wave) per workgroup and is scheduled per tile; the other
        // kernels per 16x16 block (same test as launch_render)
        const uint32_t h = std::max(std::max(sc->gbvh.height, sc->n_faces ? sc->mbvh.height : 0u), 1u);
        const bool stream_kernel = uses_stream_kernel(c->opt_kernel, a.mode, h);
        // (mesh frames too, since round 4: a part wave of the primary stage queues its own chunk of continuation rays, <= 16 of them)
        c->parts_ok = a.mode != 2 && (!sc->n_faces || c->opt_mesh_parts) && uses_tile_kernel(c->opt_kernel, a.mode, h, sc->built_leaf_max, sc->gbvh.n_prims);
        int rcf = prepare_feedback(c, a, s, stream_kernel ? a.n_blocks * 4u : a.n_blocks,
                                   uses_tile_kernel(c->opt_kernel, a.mode, h, sc->built_leaf_max, sc->gbvh.n_prims));
        if (rcf != GRT_OK) return rcf;
    }
    if (c->opt_counters) CHK(c, hipMemsetAsync(c->d_counters, 0, kNumCounters * sizeof(unsigned long long), s));
    const uint32_t depth = std::max(std::max(sc->gbvh.height, sc->n_faces ? sc->mbvh.height : 0u), 1u);
    a.prec = nullptr; a.queue = nullptr; a.qcount = nullptr;
    a.queue_in = nullptr; a.qcount_in = nullptr; a.queue_alt = nullptr;
    a.heavy = nullptr; a.hcount = nullptr; a.hnext = nullptr; a.fqueue = nullptr; a.fcount = nullptr;
    a.bundle_rounds = (uint32_t)c->opt_bundle_rounds;
    // (rays that went through a glass body are defocused but each of them light: their bundles are worth twice the work
    //  before they are given up — 1 M scene + glass sphere 8.2 -> 5.9 ms; a mirror's limb bundles are not: 4.0 -> 4.5)
    a.bundle_budget = (uint32_t)c->opt_bundle_budget * (a.p.type == GRT_GLASS ? 2u : 1u);
    a.lane_budget = (uint32_t)c->opt_lane_budget;
    a.single_own_mesh = 0;
    a.mstack_depth = sc->mbvh.height + 2u;
    // (2 = fused into the tile kernel's primary stage, on its depth-first stack: only when the mesh tree's walk fits it)
    a.mesh_primary_wave = (uint32_t)c->opt_mesh_primary_wave;
    if (a.mesh_primary_wave == 2u && a.mstack_depth > kTileStack) a.mesh_primary_wave = 1u;
    a.single_look = (float)c->opt_single_look / 1024.0f;
    a.single_band = (float)c->opt_single_band / 1024.0f;
    if (sc->n_faces && a.mode != 2) { // mesh frame: buffers of the wavefront pipeline (one record per launched thread)
        // (entries of the continuation queues: one 64-entry chunk per wave of the primary stage — per 8x8 tile, and per PART of a
        //  heavy tile when the launch order splits some)
        const size_t need = ((size_t)a.n_blocks * 4 + parts_extra_cap(a.n_blocks * 4u)) * 64;
        if (c->wf_cap < need) {
            (void)hipFree(c->d_prec); (void)hipFree(c->d_queue); (void)hipFree(c->d_heavy); (void)hipFree(c->d_fqueue);
            (void)hipFree(c->d_qunit); (void)hipFree(c->d_qskip); (void)hipFree(c->d_heavy_a); // (sized by the queues: re-made with the verdicts below)
            c->d_qunit = c->d_qskip = c->d_heavy_a = nullptr;
            c->d_prec = c->d_queue = c->d_fqueue = nullptr;
            c->d_heavy = nullptr;
            c->wf_cap = 0;
            CHK(c, hipMalloc(&c->d_prec, need * 3 * sizeof(float4)));
            CHK(c, hipMalloc(&c->d_queue, 2 * need * 4 * sizeof(float4))); // two queues: the stages ping-pong
            CHK(c, hipMalloc(&c->d_heavy, need * sizeof(uint32_t)));
            CHK(c, hipMalloc(&c->d_fqueue, need * 4 * sizeof(float4)));
            c->wf_cap = need;
        }
        if (!c->d_qcount) CHK(c, hipMalloc(&c->d_qcount, sizeof(uint32_t) * kWfCounters));
        a.prec = c->d_prec; a.queue = c->d_queue; a.qcount = c->d_qcount;
        a.queue_alt = c->d_queue + c->wf_cap * 4;
        a.heavy = c->d_heavy; a.fqueue = c->d_fqueue;
    }
    a.bverdict = nullptr; a.qunit = nullptr; a.qunit_out = nullptr; a.qunit_cap = 0; a.qskip = nullptr; a.heavy_a = nullptr; a.hcount_a = nullptr; a.bverdict_epoch = 0;
    const bool tile_kernel = uses_tile_kernel(c->opt_kernel, a.mode, depth, sc->built_leaf_max, sc->gbvh.n_prims);
    if (sc->n_faces && a.mode != 2 && tile_kernel && c->opt_bundle_predict && c->opt_bundle_rounds > 0 && c->wf_cap) {
        // bundle verdicts (RenderArgs::bverdict): one word per 8x8 tile of the launch; they belong to a launch geometry and a scene —
        // anything else starts from "every tile is a bundle" — and are used up under a view that changes
        const uint32_t nu = a.n_blocks * 4u;
        const uint64_t sig[6] = {a.mode | ((uint64_t)nu << 8), a.n_blocks, ((uint64_t)a.p.width << 32) | a.p.height,
                                 ((uint64_t)a.x0 << 48) ^ ((uint64_t)a.y0 << 32) ^ ((uint64_t)a.x1 << 16) ^ a.y1,
                                 ((uint64_t)a.first_tile << 32) | a.tile_stride, ((uint64_t)a.tile_w << 32) | a.tile_h};
        const size_t chunks = c->wf_cap / 64;
        bool fresh = false;
        if (c->bv_cap < nu || !c->d_qunit) {
            (void)hipFree(c->d_bverdict); (void)hipFree(c->d_qunit); (void)hipFree(c->d_qskip); (void)hipFree(c->d_heavy_a);
            c->d_bverdict = c->d_qunit = c->d_qskip = c->d_heavy_a = nullptr;
            c->bv_cap = 0;
            CHK(c, hipMalloc(&c->d_bverdict, sizeof(uint32_t) * (size_t)nu * kMaxBundleRounds)); // (a set of verdicts per bundle round)
            CHK(c, hipMalloc(&c->d_qunit, sizeof(uint32_t) * 2 * chunks));                       // (tile numbers of the chunks of either queue)
            CHK(c, hipMalloc(&c->d_qskip, sizeof(uint32_t) * chunks));
            CHK(c, hipMalloc(&c->d_heavy_a, sizeof(uint32_t) * c->wf_cap));
            c->bv_cap = nu;
            fresh = true;
        }
        if (fresh || memcmp(sig, c->bv_sig, sizeof(sig)) != 0 || c->bv_epoch != sc->scene_epoch) {
            CHK(c, hipMemsetAsync(c->d_bverdict, 0, sizeof(uint32_t) * (size_t)nu * kMaxBundleRounds, s)); // (on the frame's stream: ordered against its kernels)
            memcpy(c->bv_sig, sig, sizeof(sig));
            c->bv_epoch = sc->scene_epoch;
            c->bv_params_valid = false;
        }
        a.bverdict = c->d_bverdict; a.qunit = c->d_qunit; a.qunit_cap = (uint32_t)chunks; a.qskip = c->d_qskip; a.heavy_a = c->d_heavy_a;
        if (!(c->bv_params_valid && memcmp(&c->bv_params, &a.p, sizeof(grt_params)) == 0)) c->bv_view = (c->bv_view + 1u) ? c->bv_view + 1u : 1u; // another view (never 0: a cleared word is no verdict)
        a.bverdict_epoch = c->bv_view;
        c->bv_params = a.p;
        c->bv_params_valid = true;
    }
    // allocations first (they may synchronise): eye records of this slot, overflow pool
    const uint32_t m = sc->gbvh.n_prims;
    const bool want_erec = a.mode != 2 && m && c->opt_kernel != 1 && c->opt_kernel != 2;
    if (want_erec) {
        if (tile_kernel && c->cap_erec_wide < m) {
            (void)hipFree(c->d_erec_wide);
            c->d_erec_wide = nullptr;
            c->cap_erec_wide = 0;
            CHK(c, hipMalloc(&c->d_erec_wide, (size_t)m * 4 * sizeof(float4) + 256));
            c->cap_erec_wide = m;
            c->erec_valid = false;
        }
        if (!tile_kernel && c->cap_erec < m) {
            (void)hipFree(c->d_erec);
            c->d_erec = nullptr;
            c->cap_erec = 0;
            CHK(c, hipMalloc(&c->d_erec, (size_t)m * sizeof(float4) + 256));
            c->cap_erec = m;
            c->erec_valid = false;
        }
    }
    a.ovf_pool = nullptr; a.ovf_next = nullptr; a.ovf_chunks = 0;
    a.ovf_entries = c->opt_ovf_entries > 0 ? (uint32_t)c->opt_ovf_entries : kTileOvfEntries;
    // (an order whose entries carry size classes: a whole tile without one — no cost word yet — starts in one chunk and moves when it
    //  outgrows it; an order of bare unit numbers, or none: a full bag, as ever)
    //  (a scene whose tiles are nearly all deep — the needle scene: 2.8 chunks asked for per tile of the launch — gains nothing from the
    //   small start and pays a move per tile on every cold frame, 16.5 -> 18.5 ms: there a tile without a class starts with a full bag too)
    a.ovf_cls0 = (a.order && c->order_classes && !(c->ovf_units == a.n_blocks * 4u && (uint64_t)c->ovf_demand * 2u > (uint64_t)a.n_blocks * 4u * 3u)) ? 1u : 3u;
    if (tile_kernel) {
        int rco = size_overflow_pool(c, a.n_blocks * 4u, s);
        if (rco != GRT_OK) return rco;
        if (!c->d_ovf_next) { // [0] next free chunk, [1] running peak of the demand (k_frame_tail)
            CHK(c, hipMalloc(&c->d_ovf_next, 2 * sizeof(uint32_t)));
            CHK(c, hipMemsetAsync(c->d_ovf_next, 0, 2 * sizeof(uint32_t), s));
            c->ovf_zeroed = true;
        }
    }
    CHK(c, hipEventRecord(c->ev0, s));
    a.erec = nullptr;
    if (want_erec) {
        // wave-per-tile kernels on camera rays: refresh the eye records when the eye moved (part of the timed frame)
        const bool wide = tile_kernel;
        if (!c->erec_valid || c->erec_is_wide != wide || memcmp(c->erec_eye, a.p.eye, sizeof(c->erec_eye)) != 0) {
            if (wide)
                hipLaunchKernelGGL(k_eye_records_wide, dim3((m + 255) / 256), dim3(256), 0, s, sc->d_rec, m, a.p.eye[0],
                                   a.p.eye[1], a.p.eye[2], c->d_erec_wide);
            else
                hipLaunchKernelGGL(k_eye_records, dim3((m + 255) / 256), dim3(256), 0, s, sc->d_rec, m, a.p.eye[0], a.p.eye[1],
                                   a.p.eye[2], c->d_erec);
            memcpy(c->erec_eye, a.p.eye, sizeof(c->erec_eye));
            c->erec_valid = true;
            c->erec_is_wide = wide;
        }
        a.erec = wide ? c->d_erec_wide : c->d_erec;
    }
    LaunchAux aux;
    aux.aux = c->aux_stream; aux.fork = c->ev_fork; aux.join = c->ev_join;
    aux.heavy_cap = a.n_heavy ? std::max(1u, a.n_units / (uint32_t)c->opt_heavy_cap_div) : 0u; // in scheduling units
    aux.force_big = c->opt_kernel == 4;
    if (tile_kernel) { a.n_heavy = nullptr; aux.heavy_cap = 0; } // no big-window split on the tile kernel
    a.tile_ready_min = (uint32_t)c->opt_tile_ready;
    a.tile_band = (float)c->opt_tile_band / 1024.0f;
    a.tile_look = (float)c->opt_tile_look / 1024.0f;
    // (auto: 16 — C2 -4.5 %, C1 -2 % against 24, C3 / C5 / the dense-core camera unchanged — but 24 for trees with pieces, whose frontier
    //  is crowded with far ranges: the needle scene C3a 14.5 ms at 24, 15.1 at 20, 16.4 at 16, 24.1 at 14)
    a.tile_reserve = c->opt_tile_reserve >= 0 ? (uint32_t)c->opt_tile_reserve : (sc->has_pieces ? 24u : 16u);
    a.tile_band_abs = (float)c->opt_band_abs / 64.0f * sc->gm_diag;
    a.tile_prio_div = (uint32_t)c->opt_tile_prio;
    a.quad_parts = (quad_parts_ok(c, a.n_units) && tile_kernel && c->parts_ok && a.mode != 2 && a.order && a.n_launch && c->qparts_valid) ? 1u : 0u;
    a.qparts = c->d_qparts; a.qpart_count = c->d_qpcount;
    // the list's length when the host knows it (a frame tail that ran behind the list's making has copied it to the pinned word; frames of
    // a standing view keep their order and their list, so frames queued without a host synchronisation between them know it too): a launch
    // without parts skips the quad kernel and its fork / join, one with parts launches exactly that many waves; else kQuadListCap waves,
    // the idle ones exit
    a.quad_known = (a.quad_parts && c->qknown_epoch == c->qlist_epoch) ? (c->qknown_count + 1u) : 0u;
    if (a.quad_known == 1u) a.quad_parts = 0u; // (known to be empty; the order then holds no code-3 entry either)
    if (tile_kernel) {
        if (!c->ovf_zeroed) CHK(c, hipMemsetAsync(c->d_ovf_next, 0, sizeof(uint32_t), s));
        c->ovf_zeroed = false;
        a.ovf_pool = c->d_ovf; a.ovf_next = c->d_ovf_next; a.ovf_chunks = c->ovf_chunks;
    }
    {
        static const bool dbg = getenv("GRT_DEBUG_LAUNCH") != nullptr; // one line per frame on stderr: what the launch is made of
        if (dbg) {
            uint32_t d3[3] = {0, 0, 0}; // entries in use, four-way threshold, heaviest tile (as the ordering kernel left them)
            if (a.order && a.n_launch) { (void)hipDeviceSynchronize(); (void)hipMemcpy(d3, c->d_order + a.n_launch, sizeof(d3), hipMemcpyDeviceToHost); }
            fprintf(stderr, "grt launch: ctx %p mode %u units %u order %s entries %u cost %s parts_ok %d cost_valid %d order_ready %d used %u thr4 %u max %u\n",
                    (void*)c, a.mode, a.n_units, a.order ? "yes" : "no", a.n_launch, a.cost ? "collect" : "-", (int)c->parts_ok, (int)c->cost_valid,
                    (int)c->order_ready, d3[0], d3[1], d3[2]);
            static const bool dbg2 = getenv("GRT_DEBUG_LAUNCH")[0] == '2'; // ... and the launch order's parts
            if (dbg2 && a.order && a.n_launch) {
                std::vector<uint32_t> ord(a.n_launch);
                (void)hipMemcpy(ord.data(), c->d_order, sizeof(uint32_t) * a.n_launch, hipMemcpyDeviceToHost);
                uint32_t n4 = 0, n2 = 0;
                std::string head, tiles4;
                for (uint32_t i = 0; i < d3[0] && i < a.n_launch; i++) {
                    const uint32_t e = ord[i], code = e >> 30, part = (e >> 28) & 3u, unit = e & kOrderUnitMask;
                    if (part == 0u) { n4 += code == 2u; n2 += code == 1u; }
                    if (part == 0u && head.size() < 300) head += " " + std::to_string(unit) + (code ? "/" + std::to_string(code) : "");
                }
                fprintf(stderr, "grt order: four-way %u two-way %u first:%s\n", n4, n2, head.c_str());
            }
        }
    }
    int rc = launch_render(a, c->opt_counters != 0, c->opt_kernel, depth, tile_kernel, s, &aux, &c->err);
    CHK(c, hipEventRecord(c->ev1, s));
    {
        static const bool dbg_wf = getenv("GRT_DEBUG_LAUNCH") != nullptr; // mesh frames: what went through the wavefront stages (grt_internal.h: kWfCounters)
        if (dbg_wf && rc == GRT_OK && a.qcount) {
            uint32_t q[kWfCounters];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(q, a.qcount, sizeof(q), hipMemcpyDeviceToHost);
            std::string t = "grt wavefront: chunks after primary / bundle rounds:";
            for (int k = 0; k <= kMaxBundleRounds; k++) t += " " + std::to_string(q[k]);
            t += " | heavy rays per round:";
            for (int k = 0; k < kMaxBundleRounds; k++) t += " " + std::to_string(q[kMaxBundleRounds + 1 + k]);
            t += " | retry queue: " + std::to_string(q[2 * kMaxBundleRounds + 1]) + " | early lists per round (tiles known not to be bundles):";
            for (int k = 0; k < kMaxBundleRounds; k++) t += " " + std::to_string(q[3 * kMaxBundleRounds + 3 + 2 * k]);
            fprintf(stderr, "%s\n", t.c_str());
        }
    }
    c->have_timing = (rc == GRT_OK);
    // ---- behind the frame, outside its timing (grt_last_kernel_ms brackets ev0..ev1; the feedback kernels below are
    //      ~60 us per frame under a moving camera and are what `frame ms - kernel ms` of bench.py's orbit leg shows) ----
    bool tail = false;
    if (rc == GRT_OK && tile_kernel && a.cost) { // the tiles' give-up reasons -> sticky error word (a frame that repeats the
        // last one exactly collects no costs and is not checked again: it is the same computation)
        hipLaunchKernelGGL(k_check_costs, dim3((a.n_units + 255u) / 256u), dim3(256), 0, s, c->d_cost, a.n_units, a.max_iters, c->d_err);
        tail = true;
    }
    static const bool dbg_costs = getenv("GRT_DEBUG_LAUNCH") != nullptr && getenv("GRT_DEBUG_LAUNCH")[0] == '2';
    if (dbg_costs && rc == GRT_OK && a.cost && c->cost_valid) {
        // diagnostics: the heaviest tiles of the frame just rendered (steps / part code) before the costs are consumed
        (void)hipStreamSynchronize(s);
        std::vector<uint32_t> h(a.n_units);
        (void)hipMemcpy(h.data(), c->d_cost, sizeof(uint32_t) * a.n_units, hipMemcpyDeviceToHost);
        std::vector<uint32_t> idx(a.n_units);
        for (uint32_t i = 0; i < a.n_units; i++) idx[i] = i;
        std::sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return (h[x] & kCostStepsMask) > (h[y] & kCostStepsMask); });
        std::string t;
        for (uint32_t k = 0; k < 24 && k < a.n_units; k++)
            t += " " + std::to_string(idx[k]) + ":" + std::to_string(h[idx[k]] & kCostStepsMask) + "/" + std::to_string((h[idx[k]] >> kCostPartShift) & 3u);
        fprintf(stderr, "grt costs (unit:steps/code, by raw steps):%s\n", t.c_str());
    }
    if (rc == GRT_OK && a.cost && c->cost_valid) { // the next frame's launch order
        if (order_from_costs(c, a, a.n_units, s, &c->order_split, true) == GRT_OK) {
            c->order_ready = true;
            c->order_valid = true;
            // (settled: this frame itself ran with an order made for it — or no tile is ever split, and costs do not depend on the order)
            c->order_settled = c->launch_order_matched || c->order_launch == 0;
            c->order_params = a.p;
            c->order_epoch = sc->scene_epoch;
            // ... and the zeroing the next frame needs before its first wave: done by the ordering kernel itself (costs consumed)
            c->cost_zeroed = true;
            tail = true;
        }
    }
    // ONE single-thread kernel behind the frame (it was two copies and a memset, each a packet of its own on the queue): the sticky
    // error word as the frame and k_check_costs left it -> pinned host word (grt_sync reads it behind ev_tail: no blocking
    // null-stream copy, which waited for every other frame slot's stream too); the overflow chunks this frame asked for -> pinned
    // host word (sizes the pool of the frames to come), and their counter reset for the next frame
    {
        uint32_t* ovf = (rc == GRT_OK) ? a.ovf_next : nullptr;
        hipLaunchKernelGGL(k_frame_tail, dim3(1), dim3(1), 0, s, c->d_err, c->h_err, ovf, c->ovf_pending ? (uint32_t*)nullptr : c->h_ovf_used, c->d_qpcount, c->h_err + 1);
        c->tail_qepoch = c->qlist_epoch;
        if (hipGetLastError() == hipSuccess) {
            tail = true;
            if (ovf) {
                c->ovf_zeroed = true;
                if (!c->ovf_pending && hipEventRecord(c->ev_ovf, s) == hipSuccess) c->ovf_pending = true;
            }
        }
    }
    if (tail && hipEventRecord(c->ev_tail, s) == hipSuccess) { c->tail_pending = true; c->tail_stream = s; }
    return rc;
}

int grt_render(grt_ctx* c, const grt_params* p, uint8_t* d_rgb8, float* d_rgbf, uint32_t x0, uint32_t y0, uint32_t x1,
               uint32_t y1, void* stream)
{
    RenderArgs a;
    int rc = fill_common(c, p, &a);
    if (rc != GRT_OK) return rc;
    if (x1 > p->width || y1 > p->height || x0 > x1 || y0 > y1) { c->err = "grt_render: window outside the frame"; return GRT_ERR_INVALID; }
    if (!d_rgb8 && !d_rgbf) { c->err = "grt_render: no output buffer"; return GRT_ERR_INVALID; }
    a.out8 = d_rgb8; a.outf = d_rgbf;
    a.mode = 0;
    a.x0 = x0; a.y0 = y0; a.x1 = x1; a.y1 = y1;
    a.nbx = (x1 - x0 + 15) / 16;
    a.nby = (y1 - y0 + 15) / 16;
    a.n_blocks = a.nbx * a.nby;
    return do_launch(c, a, stream);
}

int grt_render_tiles(grt_ctx* c, const grt_params* p, uint8_t* d_rgb8, float* d_rgbf, uint32_t tile_w, uint32_t tile_h,
                     uint32_t first_tile, uint32_t tile_stride, uint32_t n_tiles, void* stream)
{
    RenderArgs a;
    int rc = fill_common(c, p, &a);
    if (rc != GRT_OK) return rc;
    if (!tile_w || !tile_h || (tile_w % 16) || (tile_h % 16)) { c->err = "grt_render_tiles: tile size must be a multiple of 16"; return GRT_ERR_INVALID; }
    if (!d_rgb8 && !d_rgbf) { c->err = "grt_render_tiles: no output buffer"; return GRT_ERR_INVALID; }
    const uint32_t tiles_x = (p->width + tile_w - 1) / tile_w, tiles_y = (p->height + tile_h - 1) / tile_h;
    if (n_tiles && (!tile_stride || (uint64_t)first_tile + (uint64_t)(n_tiles - 1) * tile_stride >= (uint64_t)tiles_x * tiles_y)) {
        c->err = "grt_render_tiles: tile range outside the frame's tile grid";
        return GRT_ERR_INVALID;
    }
    a.out8 = d_rgb8; a.outf = d_rgbf;
    a.mode = 1;
    a.tile_w = tile_w; a.tile_h = tile_h; a.first_tile = first_tile; a.tile_stride = tile_stride; a.n_tiles = n_tiles;
    a.tiles_x = tiles_x;
    a.nbx = tile_w / 16; a.nby = tile_h / 16;
    a.n_blocks = n_tiles * a.nbx * a.nby;
    return do_launch(c, a, stream);
}

// rank 0 of an N-rank frame: the gathered compact buffers back into screen order ("rank 0 un-permutes tiles
// with a trivial copy kernel").  Tile t of the row-major tile grid was rendered by rank t % world as its tile t / world.
__global__ void k_assemble_tiles(const uint8_t* __restrict__ g, uint32_t world, uint32_t max_cnt, uint32_t tile_w, uint32_t tile_h,
                                 uint32_t tiles_x, uint32_t width, uint32_t height, uint8_t* __restrict__ out)
{
    const uint32_t px = blockIdx.x * blockDim.x + threadIdx.x, py = blockIdx.y;
    if (px >= width || py >= height) return;
    const uint32_t t = (py / tile_h) * tiles_x + px / tile_w;
    const uint32_t r = t % world, j = t / world;
    const size_t src = ((((size_t)r * max_cnt + j) * tile_h + py % tile_h) * tile_w + px % tile_w) * 3;
    const size_t dst = ((size_t)py * width + px) * 3;
    out[dst] = g[src]; out[dst + 1] = g[src + 1]; out[dst + 2] = g[src + 2];
}

int grt_assemble_tiles(grt_ctx* c, const uint8_t* d_gathered, uint32_t world, uint32_t max_cnt, uint32_t tile_w, uint32_t tile_h,
                       uint32_t width, uint32_t height, uint8_t* d_rgb8, void* stream)
{
    if (!c) return GRT_ERR_INVALID;
    if (!d_gathered || !d_rgb8 || !world || !tile_w || !tile_h || !width || !height) { c->err = "grt_assemble_tiles: bad arguments"; return GRT_ERR_INVALID; }
    const uint32_t tiles_x = (width + tile_w - 1) / tile_w, tiles_y = (height + tile_h - 1) / tile_h;
    if ((uint64_t)max_cnt * world < (uint64_t)tiles_x * tiles_y) { c->err = "grt_assemble_tiles: world x max_cnt tiles do not cover the frame"; return GRT_ERR_INVALID; }
    CHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    hipLaunchKernelGGL(k_assemble_tiles, dim3((width + 255) / 256, height), dim3(256), 0, s, d_gathered, world, max_cnt, tile_w, tile_h,
                       tiles_x, width, height, d_rgb8);
    CHK(c, hipGetLastError());
    return GRT_OK;
}

int grt_render_rays(grt_ctx* c, const grt_params* p, const float* d_rays, uint64_t n, float* d_rgbf, void* stream)
{
    RenderArgs a;
    int rc = fill_common(c, p, &a);
    if (rc != GRT_OK) return rc;
    if (n && (!d_rays || !d_rgbf)) { c->err = "grt_render_rays: null buffer"; return GRT_ERR_INVALID; }
    if (n > 0xFFFFFFFFull * 64) { c->err = "grt_render_rays: too many rays"; return GRT_ERR_LIMIT; }
    a.outf = d_rgbf;
    a.mode = 2;
    a.rays = d_rays; a.n_rays = n;
    a.n_blocks = (uint32_t)((n + 255) / 256);
    return do_launch(c, a, stream);
}

// The sticky device error word (RenderArgs::err_word): a wave that had to give up on a ray — watchdog, depth-first stack
// guard, two passes without progress — ORs its reason in, in EVERY kernel variant (counters on or off).  Read (and
// cleared) at the synchronising entry points; the reference turns traversal trouble into exceptions the same way
// (OptiX exception flags, src/GaussianTracer.cpp:114-119; src/Exception.h:31-80).
static int check_device_error(grt_ctx* c)
{
    // the last frame may have gone to ANY stream (the caller's, a view's): what ran behind it — k_check_costs, which
    // turns the tile kernel's give-up reasons into the error word, and the copy of the word to h_err — is finished once
    // ev_tail is.  (Waiting for c->stream and ev1 alone returned GRT_OK for a frame on a non-blocking side stream whose
    // tiles had given up: ADVICE r03.)
    if (c->tail_pending) CHK(c, hipEventSynchronize(c->ev_tail));
    const uint32_t w = *(volatile uint32_t*)c->h_err;
    if (!w) return GRT_OK;
    hipStream_t s = c->tail_stream ? c->tail_stream : c->stream;
    CHK(c, hipMemsetAsync(c->d_err, 0, sizeof(w), s)); // stream-ordered before the next frame's kernels (do_launch waits on ev_tail)
    *c->h_err = 0;
    if (hipEventRecord(c->ev_tail, s) == hipSuccess) { c->tail_pending = true; c->tail_stream = s; }
    c->err = std::string("render: a wave gave up on live rays (pixels are missing hits):") +
             ((w & kErrWatchdog) ? " step watchdog expired;" : "") + ((w & kErrStack) ? " depth-first overflow stack full;" : "") +
             ((w & kErrStall) ? " two passes without progress;" : "");
    return GRT_ERR_LIMIT;
}

int grt_sync(grt_ctx* c)
{
    if (!c) return GRT_ERR_INVALID;
    CHK(c, hipSetDevice(c->device));
    CHK(c, hipStreamSynchronize(c->stream));
    // (renders go to the caller's stream: the last frame is finished once the event recorded behind its kernels is)
    if (c->have_timing) CHK(c, hipEventSynchronize(c->ev1));
    CHK(c, hipGetLastError());
    return check_device_error(c);
}

int grt_get_counters(grt_ctx* c, grt_counters* out)
{
    if (!c || !out) return GRT_ERR_INVALID;
    CHK(c, hipSetDevice(c->device));
    CHK(c, hipDeviceSynchronize());
    unsigned long long h[kNumCounters];
    CHK(c, hipMemcpy(h, c->d_counters, sizeof(h), hipMemcpyDeviceToHost));
    out->rays = h[0]; out->segments = h[1]; out->hit_evals = h[2]; out->rounds = h[3];
    out->node_visits = h[4]; out->proxy_tests = h[5]; out->rec_fetches = h[6]; out->stall_exits = h[7];
    return check_device_error(c);
}

int grt_last_kernel_ms(grt_ctx* c, float* ms)
{
    if (!c || !ms) return GRT_ERR_INVALID;
    if (!c->have_timing) { c->err = "grt_last_kernel_ms: no render has been launched"; return GRT_ERR_INVALID; }
    CHK(c, hipSetDevice(c->device));
    CHK(c, hipEventSynchronize(c->ev1));
    CHK(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return GRT_OK;
}

} // extern "C"
