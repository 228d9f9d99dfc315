// grt_bvh.hip — LBVH builder for gfx950 (replaces the closed OptiX accel build behind
// GaussianTracer::createGaussianParticlesBVH / createGAS / buildAccelationStructure,
// src/GaussianTracer.cpp:297-473 of the reference).
//
// Pipeline (all on the device, one stream):
//   1. k_scene_bounds   centroid bounds + count of valid primitives (wave reduce -> ordered-int atomics)
//   2. k_morton         63-bit Morton key of the box centroid (21 bits / axis); invalid => ~0 (sorts last)
//   3. stable LSD radix sort of (key, primitive index), hand-written (radix_sort_pairs_64_32 below)
//   4. k_leaf_boxes     boxes gathered into sorted order
//   5. k_hierarchy      Karras 2012 internal nodes from the sorted keys (ties broken by index)
//   6. k_refit_pass     bottom-up boxes, one launch per tree level: a node is finished in pass p only
//                       from children finished in passes < p, so every hand-off crosses a kernel
//                       boundary (no in-launch inter-workgroup visibility protocol needed);
//                       the number of passes IS the tree height, which sizes the traversal stack.
// The BVH only culls: boxes are inflated by 1e-5*(1+|coordinate|) so that the exact slab test in
// grt_render.hip, not the box, decides every hit.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string.h>

#include <hip/hip_runtime.h>

#include "grt_internal.h"

namespace grt {

#define HIPCHK(x)                                                                                     \
    do {                                                                                              \
        hipError_t e_ = (x);                                                                          \
        if (e_ != hipSuccess) {                                                                       \
            if (err) *err = std::string(#x) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"; \
            goto fail;                                                                                \
        }                                                                                             \
    } while (0)

__device__ __forceinline__ uint32_t f2ord(float f)
{
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ __forceinline__ float ord2f(uint32_t o)
{
    uint32_t b = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(b);
#else
    float f;
    memcpy(&f, &b, 4);
    return f;
#endif
}

// bounds[0..2] = min centroid (ordered uint), [3..5] = max, [6] = valid count, [7] = sum of the box diagonals (float),
// [8 + b] = workgroup b's part of that sum.  The sum decides the size classes (k_morton) and with them the Morton order
// and the tree, so it is reduced in a FIXED order (lanes by shuffles, waves and workgroups sequentially): the same
// scene gives the same BVH on every run and every rank (a float atomicAdd across waves did not).
__global__ void k_scene_bounds(const float4* __restrict__ lo, const float4* __restrict__ hi, uint32_t n,
                               uint32_t* __restrict__ bounds)
{
    uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0, 0, 0}, cnt = 0;
    float dsum = 0.0f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 l = lo[i], h = hi[i];
        if (l.x <= h.x) {
            const float c[3] = {0.5f * (l.x + h.x), 0.5f * (l.y + h.y), 0.5f * (l.z + h.z)};
            for (int k = 0; k < 3; k++) {
                const uint32_t o = f2ord(c[k]);
                mn[k] = min(mn[k], o);
                mx[k] = max(mx[k], o);
            }
            cnt++;
            dsum += sqrtf((h.x - l.x) * (h.x - l.x) + (h.y - l.y) * (h.y - l.y) + (h.z - l.z) * (h.z - l.z));
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        for (int k = 0; k < 3; k++) {
            mn[k] = min(mn[k], (uint32_t)__shfl_xor((int)mn[k], off));
            mx[k] = max(mx[k], (uint32_t)__shfl_xor((int)mx[k], off));
        }
        cnt += (uint32_t)__shfl_xor((int)cnt, off);
        dsum += __shfl_xor(dsum, off);
    }
    // one set of atomics per WORKGROUP (the waves meet in LDS first): 7 x 256 atomics instead of 7 x 8192 on one line
    __shared__ uint32_t red[4][7];
    __shared__ float dred[4];
    const uint32_t wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        for (int k = 0; k < 3; k++) { red[wv][k] = mn[k]; red[wv][3 + k] = mx[k]; }
        red[wv][6] = cnt;
        dred[wv] = dsum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t nw = (blockDim.x + 63u) >> 6;
        float ds = dred[0];
        for (uint32_t w = 1; w < nw; w++) {
            for (int k = 0; k < 3; k++) { red[0][k] = min(red[0][k], red[w][k]); red[0][3 + k] = max(red[0][3 + k], red[w][3 + k]); }
            red[0][6] += red[w][6];
            ds += dred[w];
        }
        bounds[8 + blockIdx.x] = __float_as_uint(ds);
        for (int k = 0; k < 3; k++) {
            atomicMin(&bounds[k], red[0][k]);
            atomicMax(&bounds[3 + k], red[0][3 + k]);
        }
        atomicAdd(&bounds[6], red[0][6]);
    }
}

__global__ void k_scene_bounds_finish(uint32_t* __restrict__ bounds, uint32_t n_parts)
{
    float s = 0.0f;
    for (uint32_t b = 0; b < n_parts; b++) s += __uint_as_float(bounds[8 + b]);
    bounds[7] = __float_as_uint(s);
}

__device__ __forceinline__ uint64_t expand21(uint32_t v)
{
    uint64_t x = v & 0x1FFFFFu;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ void k_morton(const float4* __restrict__ lo, const float4* __restrict__ hi, uint32_t n,
                         const uint32_t* __restrict__ bounds, uint64_t* __restrict__ keys,
                         uint32_t* __restrict__ vals, int size_classes)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 l = lo[i], h = hi[i];
    uint64_t key = ~0ull;
    if (l.x <= h.x) {
        const float c[3] = {0.5f * (l.x + h.x), 0.5f * (l.y + h.y), 0.5f * (l.z + h.z)};
        uint32_t q[3];
        for (int k = 0; k < 3; k++) {
            const float a = ord2f(bounds[k]), b = ord2f(bounds[3 + k]);
            const float ext = b - a;
            float u = ext > 0.0f ? (c[k] - a) / ext : 0.0f;
            u = fminf(fmaxf(u, 0.0f), 1.0f);
            q[k] = min((uint32_t)(u * 2097152.0f), 2097151u);
        }
        key = (expand21(q[0]) << 2) | (expand21(q[1]) << 1) | expand21(q[2]);
        // Size classes.  A primitive much larger than the average (a needle / pancake Gaussian, a background splat) sits
        // among small neighbours in Morton order and inflates every ancestor's box.  The two key bits ABOVE the 61-bit
        // code (the code gives up its two lowest bits) hold log_6 of its diagonal over the mean diagonal, clamped to
        // 0..3: Karras splits on those bits first, so each class gets a subtree of its own under the root and the
        // normal-sized majority keeps tight boxes (LBVH-quality measure of SURVEY §7.5; a pure re-ordering of the
        // primitives: hits never depend on it).
        const float mean_d = __uint_as_float(bounds[7]) / fmaxf((float)bounds[6], 1.0f);
        const float d = sqrtf((h.x - l.x) * (h.x - l.x) + (h.y - l.y) * (h.y - l.y) + (h.z - l.z) * (h.z - l.z));
        const uint32_t cls = size_classes ? ((d > 216.0f * mean_d) ? 3u : (d > 36.0f * mean_d) ? 2u : (d > 6.0f * mean_d) ? 1u : 0u) : 0u;
        key = (key >> 2) | ((uint64_t)cls << 61);
    }
    keys[i] = key;
    vals[i] = i;
}

__global__ void k_leaf_boxes(const float4* __restrict__ lo, const float4* __restrict__ hi,
                             const uint32_t* __restrict__ order, uint32_t m, float4* __restrict__ lb_lo,
                             float4* __restrict__ lb_hi)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const uint32_t i = order[j];
    lb_lo[j] = lo[i];
    lb_hi[j] = hi[i];
}

// ---- hand-written stable LSD radix sort of (64-bit Morton key, 32-bit primitive index) pairs -------------------------
// Eight passes of 8 bits.  Per pass: k_rs_hist (one 256-bin histogram per 2048-key tile, in LDS), k_rs_scan (ONE
// workgroup: exclusive scan of the bin-major [256][tiles] table, i.e. every tile's first output slot per digit),
// k_rs_scatter (the tile again, 256 keys at a time IN ORDER: a key's rank among the tile's equal digits = what earlier
// rounds of the tile counted + what lower waves of this round counted + its rank inside the wave, from eight ballots —
// gfx950 has no match-any — so the sort is stable and Karras' index tie-break keeps meaning the input order).
// 24 launches, ~0.3 ms for 1 M keys; the build is one-off.  k_rs_check proves the result (sorted, stable) on the device
// after every build: a wrong tree would otherwise only show as missing hits.
constexpr uint32_t kRsTile = 2048u; // keys per workgroup and pass (8 rounds of 256)

__global__ __launch_bounds__(256) void k_rs_hist(const uint64_t* __restrict__ keys, uint32_t n, uint32_t shift, uint32_t n_tiles,
                                                 uint32_t* __restrict__ hist)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t base = blockIdx.x * kRsTile;
    for (uint32_t r = 0; r < kRsTile / 256u; r++) {
        const uint32_t i = base + r * 256u + threadIdx.x;
        if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * n_tiles + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of `cnt` uint32 values in place, one workgroup of 1024 (the table of a pass: 256 x tiles entries)
__global__ __launch_bounds__(1024) void k_rs_scan(uint32_t* __restrict__ v, uint32_t cnt)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0u;
    __syncthreads();
    for (uint32_t base = 0; base < cnt; base += 1024u) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t x = i < cnt ? v[i] : 0u;
        uint32_t incl = x;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = (uint32_t)__shfl_up((int)incl, off);
            if (lane >= (uint32_t)off) incl += y;
        }
        if (lane == 63u) wsum[wv] = incl;
        __syncthreads();
        uint32_t before = carry;
        for (uint32_t w = 0; w < wv; w++) before += wsum[w];
        if (i < cnt) v[i] = before + incl - x;
        __syncthreads();
        if (threadIdx.x == 1023u) carry = before + incl;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_rs_scatter(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t n,
                                                    uint32_t shift, uint32_t n_tiles, const uint32_t* __restrict__ offs,
                                                    uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out)
{
    __shared__ uint32_t next[256];     // next free output slot of every digit for this tile
    __shared__ uint32_t wcnt[4][256];  // keys of every digit in each wave of the current round
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    next[threadIdx.x] = offs[threadIdx.x * n_tiles + blockIdx.x];
    const uint32_t base = blockIdx.x * kRsTile;
    for (uint32_t r = 0; r < kRsTile / 256u; r++) {
        for (uint32_t w = 0; w < 4u; w++) wcnt[w][threadIdx.x] = 0u;
        __syncthreads();
        const uint32_t i = base + r * 256u + threadIdx.x;
        const bool ok = i < n;
        const uint64_t key = ok ? keys[i] : 0ull;
        const uint32_t d = (uint32_t)(key >> shift) & 255u;
        // lanes of this wave that hold the same digit (and a key at all)
        uint64_t same = __ballot(ok);
        for (uint32_t b = 0; b < 8u; b++) {
            const uint64_t m = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        if (ok && rank == 0u) wcnt[wv][d] = (uint32_t)__popcll(same);
        __syncthreads();
        if (ok) {
            uint32_t pos = next[d] + rank;
            for (uint32_t w = 0; w < wv; w++) pos += wcnt[w][d];
            keys_out[pos] = key;
            vals_out[pos] = vals[i];
        }
        __syncthreads();
        next[threadIdx.x] += wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
        __syncthreads();
    }
}

// ---- exclusive scan of n uint32 values (the piece offsets of grt_build_bvh's spatial splits): per 1024-value block its
// sum, one-workgroup scan of the block sums (k_rs_scan), then every block scans itself on top of its offset ----
__global__ __launch_bounds__(1024) void k_scan_block_sums(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ bsum)
{
    __shared__ uint32_t wsum[16];
    const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
    uint32_t x = i < n ? in[i] : 0u;
    for (int off = 32; off > 0; off >>= 1) x += (uint32_t)__shfl_xor((int)x, off);
    if ((threadIdx.x & 63u) == 0u) wsum[threadIdx.x >> 6] = x;
    __syncthreads();
    if (threadIdx.x == 0u) {
        uint32_t t = 0;
        for (int w = 0; w < 16; w++) t += wsum[w];
        bsum[blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(1024) void k_scan_blocks(const uint32_t* __restrict__ in, uint32_t n, const uint32_t* __restrict__ boffs,
                                                      uint32_t* __restrict__ out)
{
    __shared__ uint32_t wsum[16];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
    const uint32_t x = i < n ? in[i] : 0u;
    uint32_t incl = x;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)incl, off);
        if (lane >= (uint32_t)off) incl += y;
    }
    if (lane == 63u) wsum[wv] = incl;
    __syncthreads();
    uint32_t before = boffs[blockIdx.x];
    for (uint32_t w = 0; w < wv; w++) before += wsum[w];
    if (i < n) out[i] = before + incl - x;
}
int device_exclusive_scan_u32(const uint32_t* d_in, uint32_t* d_out, uint32_t n, hipStream_t stream, std::string* err)
{
    if (n == 0) return GRT_OK;
    const uint32_t nb = (n + 1023u) / 1024u;
    uint32_t* d_bsum = nullptr;
    hipError_t e = hipMalloc(&d_bsum, sizeof(uint32_t) * nb);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_scan_block_sums, dim3(nb), dim3(1024), 0, stream, d_in, n, d_bsum);
        hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(1024), 0, stream, d_bsum, nb);
        hipLaunchKernelGGL(k_scan_blocks, dim3(nb), dim3(1024), 0, stream, d_in, n, d_bsum, d_out);
        e = hipStreamSynchronize(stream);
        if (e == hipSuccess) e = hipGetLastError();
    }
    (void)hipFree(d_bsum);
    if (e != hipSuccess) {
        if (err) *err = std::string("device_exclusive_scan_u32: ") + hipGetErrorString(e);
        return GRT_ERR_HIP;
    }
    return GRT_OK;
}

// sorted ascending, and stable (equal keys keep the order of their values = input positions): else *bad is set
__global__ void k_rs_check(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t n, uint32_t* __restrict__ bad)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1u >= n) return;
    const uint64_t a = keys[i], b = keys[i + 1u];
    if (a > b || (a == b && vals[i] >= vals[i + 1u])) atomicOr(bad, 1u);
}

// keys / vals are clobbered (ping-pong); the sorted pairs end in keys_out / vals_out.  tmp: 256 x tiles uint32.
static hipError_t radix_sort_pairs_64_32(uint64_t* keys, uint64_t* keys_out, uint32_t* vals, uint32_t* vals_out, uint32_t n,
                                         uint32_t* tmp, hipStream_t stream)
{
    const uint32_t n_tiles = (n + kRsTile - 1u) / kRsTile;
    uint64_t* k[2] = {keys, keys_out};
    uint32_t* v[2] = {vals, vals_out};
    int in = 0;
    for (uint32_t pass = 0; pass < 8u; pass++) { // (an even number of passes: the result lands in keys_out after a final swap below)
        hipLaunchKernelGGL(k_rs_hist, dim3(n_tiles), dim3(256), 0, stream, k[in], n, pass * 8u, n_tiles, tmp);
        hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(1024), 0, stream, tmp, 256u * n_tiles);
        hipLaunchKernelGGL(k_rs_scatter, dim3(n_tiles), dim3(256), 0, stream, k[in], v[in], n, pass * 8u, n_tiles, tmp, k[in ^ 1], v[in ^ 1]);
        in ^= 1;
    }
    // eight passes: the sorted data is back in keys / vals — the caller wants it in keys_out / vals_out
    hipError_t e = hipMemcpyAsync(keys_out, k[in], sizeof(uint64_t) * n, hipMemcpyDeviceToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(vals_out, v[in], sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, stream);
    if (e == hipSuccess) e = hipGetLastError();
    return e;
}

__device__ __forceinline__ int delta(const uint64_t* __restrict__ keys, int m, int i, int j)
{
    if (j < 0 || j >= m) return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((uint32_t)i ^ (uint32_t)j);
    return __clzll((long long)(a ^ b));
}

// Karras, "Maximizing parallelism in the construction of BVHs, octrees, and k-d trees" (HPG 2012)
__global__ void k_hierarchy(const uint64_t* __restrict__ keys, int m, float4* __restrict__ nodes,
                            uint2* __restrict__ range)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m - 1) return;
    const int d = (delta(keys, m, i, i + 1) - delta(keys, m, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, m, i, i - d);
    int lmax = 2;
    while (delta(keys, m, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, m, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, m, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (delta(keys, m, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + min(d, 0);
    const uint32_t c0 = (min(i, j) == gamma) ? ((uint32_t)gamma | kLeafBit) : (uint32_t)gamma;
    const uint32_t c1 = (max(i, j) == gamma + 1) ? ((uint32_t)(gamma + 1) | kLeafBit) : (uint32_t)(gamma + 1);
    nodes[(size_t)i * 4 + 3] = make_float4(__uint_as_float(c0), __uint_as_float(c1), 0.0f, 0.0f);
    range[i] = make_uint2((uint32_t)min(i, j), (uint32_t)max(i, j));
}

// Collapse bottom subtrees: a child that is an internal node covering <= leaf_max sorted primitives becomes a
// leaf RANGE reference (the nodes below it are simply never reached).  Every Karras node covers a contiguous
// range of the sorted order, so no data moves.
__global__ void k_collapse(float4* __restrict__ nodes, const uint2* __restrict__ range, int m, uint32_t leaf_max)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m - 1) return;
    float4 q3 = nodes[(size_t)i * 4 + 3];
    uint32_t c[2] = {__float_as_uint(q3.x), __float_as_uint(q3.y)};
#pragma unroll
    for (int k = 0; k < 2; k++) {
        if (!(c[k] & kLeafBit)) {
            const uint2 r = range[c[k]];
            const uint32_t n = r.y - r.x + 1u;
            if (n <= leaf_max) c[k] = kLeafBit | ((n - 1u) << 28) | r.x;
        }
    }
    nodes[(size_t)i * 4 + 3] = make_float4(__uint_as_float(c[0]), __uint_as_float(c[1]), 0.0f, 0.0f);
}

__device__ __forceinline__ bool child_box(uint32_t c, uint32_t pass, const float4* __restrict__ nodes,
                                          const uint32_t* __restrict__ level, const float4* __restrict__ lb_lo,
                                          const float4* __restrict__ lb_hi, float lo[3], float hi[3])
{
    if (c & kLeafBit) {
        const float4 l = lb_lo[c & ~kLeafBit], h = lb_hi[c & ~kLeafBit];
        lo[0] = l.x; lo[1] = l.y; lo[2] = l.z;
        hi[0] = h.x; hi[1] = h.y; hi[2] = h.z;
        return true;
    }
    const uint32_t lv = level[c];
    if (lv == 0 || lv >= pass) return false; // not finished before this launch
    const float4 q0 = nodes[(size_t)c * 4], q1 = nodes[(size_t)c * 4 + 1], q2 = nodes[(size_t)c * 4 + 2];
    lo[0] = fminf(q0.x, q1.z); lo[1] = fminf(q0.y, q1.w); lo[2] = fminf(q0.z, q2.x);
    hi[0] = fmaxf(q0.w, q2.y); hi[1] = fmaxf(q1.x, q2.z); hi[2] = fmaxf(q1.y, q2.w);
    return true;
}

__global__ void k_refit_pass(float4* __restrict__ nodes, uint32_t* __restrict__ level, int m, uint32_t pass,
                             const float4* __restrict__ lb_lo, const float4* __restrict__ lb_hi)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m - 1) return;
    if (level[i] != 0) return;
    const float4 q3 = nodes[(size_t)i * 4 + 3];
    const uint32_t c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y);
    float l0[3], h0[3], l1[3], h1[3];
    if (!child_box(c0, pass, nodes, level, lb_lo, lb_hi, l0, h0)) return;
    if (!child_box(c1, pass, nodes, level, lb_lo, lb_hi, l1, h1)) return;
    nodes[(size_t)i * 4 + 0] = make_float4(l0[0], l0[1], l0[2], h0[0]);
    nodes[(size_t)i * 4 + 1] = make_float4(h0[1], h0[2], l1[0], l1[1]);
    nodes[(size_t)i * 4 + 2] = make_float4(l1[2], h1[0], h1[1], h1[2]);
    level[i] = pass;
}

// ---- tree rotations: the quality pass behind the Karras build (reference: OPTIX_BUILD_FLAG_PREFER_FAST_TRACE, src/GaussianTracer.cpp:360) ----
// An LBVH takes its topology from the Morton order alone; where proxies of very different sizes overlap (a trained scene, the needle /
// sheet scene C3a) a node often pairs a large box with a small one although a grandchild would make the tighter pair.  One bottom-up
// sweep of ROTATIONS (Kensler 2008) repairs the worst of that: for node N = (S, R) with R = (RL, RR) internal, swapping the sibling S
// with RL or RR changes nothing but R's own box; the swap that shrinks R's surface area most is applied (four candidates per node: either
// child may play R).  N's box is the union of the same three boxes as before, so nothing above N changes; every node record holds its
// two children's boxes, so a swap rewrites two records, N's and R's, and needs no parent pointers.  Nodes are taken in the order the
// refit finished them (a pass per level, a kernel per pass: R was finished before N), after the bottom subtrees have been collapsed into
// leaf ranges — a range stays a contiguous run of the sorted primitives, whatever happens above it.  Culling structure only: hits never
// depend on it.
__device__ __forceinline__ float box_area6(const float* b)
{
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return dx * dy + dy * dz + dz * dx;
}
__device__ __forceinline__ float union_area6(const float* a, const float* b)
{
    const float dx = fmaxf(a[3], b[3]) - fminf(a[0], b[0]), dy = fmaxf(a[4], b[4]) - fminf(a[1], b[1]), dz = fmaxf(a[5], b[5]) - fminf(a[2], b[2]);
    return dx * dy + dy * dz + dz * dx;
}
__device__ __forceinline__ void node_children(const float4* __restrict__ nodes, uint32_t i, float b0[6], float b1[6], uint32_t& c0, uint32_t& c1)
{
    const float4 q0 = nodes[(size_t)i * 4], q1 = nodes[(size_t)i * 4 + 1], q2 = nodes[(size_t)i * 4 + 2], q3 = nodes[(size_t)i * 4 + 3];
    b0[0] = q0.x; b0[1] = q0.y; b0[2] = q0.z; b0[3] = q0.w; b0[4] = q1.x; b0[5] = q1.y;
    b1[0] = q1.z; b1[1] = q1.w; b1[2] = q2.x; b1[3] = q2.y; b1[4] = q2.z; b1[5] = q2.w;
    c0 = __float_as_uint(q3.x); c1 = __float_as_uint(q3.y);
}
__device__ __forceinline__ void node_store(float4* __restrict__ nodes, uint32_t i, const float b0[6], const float b1[6], uint32_t c0, uint32_t c1)
{
    nodes[(size_t)i * 4 + 0] = make_float4(b0[0], b0[1], b0[2], b0[3]);
    nodes[(size_t)i * 4 + 1] = make_float4(b0[4], b0[5], b1[0], b1[1]);
    nodes[(size_t)i * 4 + 2] = make_float4(b1[2], b1[3], b1[4], b1[5]);
    nodes[(size_t)i * 4 + 3] = make_float4(__uint_as_float(c0), __uint_as_float(c1), 0.0f, 0.0f);
}
__global__ void k_rotate_pass(float4* __restrict__ nodes, const uint32_t* __restrict__ level, int m, uint32_t pass, uint32_t* __restrict__ n_done)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m - 1 || level[i] != pass) return;
    float bc[2][6];
    uint32_t cc[2];
    node_children(nodes, (uint32_t)i, bc[0], bc[1], cc[0], cc[1]);
    float best = 0.0f; // largest reduction of R's area found
    int bk = -1, bg = 0;
    float g[2][2][6];
    uint32_t gc[2][2];
#pragma unroll
    for (int k = 0; k < 2; k++) { // child k plays R, child 1 - k the sibling S
        if (cc[k] & kLeafBit) continue;
        node_children(nodes, cc[k], g[k][0], g[k][1], gc[k][0], gc[k][1]);
        const float a_old = box_area6(bc[k]);
#pragma unroll
        for (int t = 0; t < 2; t++) { // S <-> grandchild t: R' = (S, grandchild 1 - t)
            const float gain = a_old - union_area6(bc[1 - k], g[k][1 - t]);
            if (gain > best) { best = gain; bk = k; bg = t; }
        }
    }
    if (bk < 0 || !(best > 1e-6f * box_area6(bc[bk]))) return;
    const int k = bk, t = bg;
    // R' = (S, G[1-t]) keeps R's index; N = (G[t], R') in R's place
    float nb[6];
#pragma unroll
    for (int q = 0; q < 3; q++) { nb[q] = fminf(bc[1 - k][q], g[k][1 - t][q]); nb[q + 3] = fmaxf(bc[1 - k][q + 3], g[k][1 - t][q + 3]); }
    node_store(nodes, cc[k], bc[1 - k], g[k][1 - t], cc[1 - k], gc[k][1 - t]);
    if (k == 0) node_store(nodes, (uint32_t)i, nb, g[k][t], cc[k], gc[k][t]);
    else        node_store(nodes, (uint32_t)i, g[k][t], nb, gc[k][t], cc[k]);
    if (n_done) atomicAdd(n_done, 1u);
}

// Levels of a tree whose topology a rotation sweep has changed: level = 1 + the larger of the children's (a leaf reference counts 0), found
// the way the refit found them — a launch per level, a node is numbered in pass p only from children numbered in passes < p, so every
// hand-off crosses a kernel boundary.  The root's level is the tree's HEIGHT, which sizes the traversal stacks (the sweep pushes
// a sibling one level down, so the height the refit measured before it is no bound any more), and levels numbered for the topology as it
// stands are what makes a further sweep sound (two nodes of one pass are then never parent and child).
__global__ void k_level_pass(const float4* __restrict__ nodes, uint32_t* __restrict__ level, int m, uint32_t pass)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m - 1 || level[i] != 0) return;
    const float4 q3 = nodes[(size_t)i * 4 + 3];
    const uint32_t c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y);
    if (!(c0 & kLeafBit)) { const uint32_t lv = level[c0]; if (lv == 0 || lv >= pass) return; }
    if (!(c1 & kLeafBit)) { const uint32_t lv = level[c1]; if (lv == 0 || lv >= pass) return; }
    level[i] = pass;
}

// ---- launch order of the scheduling units: heaviest first, by cost CLASS ----
// A class keeps the leading 3 bits of the cost (124 classes); inside a class the units stay in (approximately) screen
// order, so neighbours that share BVH nodes are launched together and — through the XCD-aware rank map of the kernels —
// on one XCD / L2.  One workgroup does it all (histogram, scan, scatter in chunks of 1024 units in order): ~10 us for
// 32 k units, instead of a general radix/merge sort's eight small launches in front of every frame.  The same kernel
// counts the heavy units (cost above thr_x2/2 x the median, at class granularity) for the split launch.
__device__ __forceinline__ uint32_t cost_class(uint32_t c)
{
    if (c < 8u) return c;
    const uint32_t e = 31u - (uint32_t)__clz((int)c);
    return (e - 1u) * 4u + ((c >> (e - 2u)) & 3u);
}
__device__ __forceinline__ uint32_t cost_class_floor(uint32_t k)
{
    if (k < 8u) return k;
    if (k >= 124u) return 0xFFFFFFFFu; // classes above the largest 32-bit cost: never populated
    const uint32_t e = k / 4u + 1u, m = k & 3u;
    return (4u | m) << (e - 2u);
}

constexpr uint32_t kOrdBatch = 8u; // units per thread whose costs an ordering pass loads before it uses them
__global__ __launch_bounds__(1024) void k_cost_order(const uint32_t* cost, uint32_t n,
                                                     uint32_t* __restrict__ order, uint32_t cap, uint32_t thr_x2,
                                                     uint32_t* __restrict__ n_heavy, uint32_t* zero) // (`zero` may alias `cost`: neither is restrict)
{
    __shared__ uint32_t hist[128], cursor[128];
    const uint32_t tid = threadIdx.x;
    if (tid < 128u) hist[tid] = 0u;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += kOrdBatch * 1024u) { // (loads first, kOrdBatch per thread: see k_cost_order_parts)
        uint32_t cv[kOrdBatch];
#pragma unroll
        for (uint32_t k = 0; k < kOrdBatch; k++) { const uint32_t i = base + k * 1024u + tid; cv[k] = (i < n) ? cost[i] : 0u; }
#pragma unroll
        for (uint32_t k = 0; k < kOrdBatch; k++)
            if (base + k * 1024u + tid < n) atomicAdd(&hist[127u - cost_class(cost_eff(cv[k]))], 1u); // bucket 0 = heaviest
    }
    __syncthreads();
    if (tid == 0u) {
        uint32_t acc = 0, bmed = 127u;
        bool found = false;
        for (uint32_t b = 0; b < 128u; b++) {
            cursor[b] = acc;
            acc += hist[b];
            if (!found && acc > n / 2u) { bmed = b; found = true; }
        }
        if (n_heavy) {
            const uint32_t thr = (cost_class_floor(127u - bmed) * thr_x2) >> 1;
            uint32_t bh = 0;
            while (bh < 128u && cost_class_floor(127u - bh) > thr) bh++;
            const uint32_t heavy = bh < 128u ? cursor[bh] : n;
            *n_heavy = heavy < cap ? heavy : cap;
        }
    }
    __syncthreads();
    for (uint32_t base = 0; base < n; base += kOrdBatch * 1024u) {
        uint32_t cv[kOrdBatch];
#pragma unroll
        for (uint32_t k = 0; k < kOrdBatch; k++) { const uint32_t i = base + k * 1024u + tid; cv[k] = (i < n) ? cost[i] : 0u; }
#pragma unroll
        for (uint32_t k = 0; k < kOrdBatch; k++) { // (1024 units at a time: the units of a class stay in screen order)
            const uint32_t i = base + k * 1024u + tid;
            if (i < n) order[atomicAdd(&cursor[127u - cost_class(cost_eff(cv[k]))], 1u)] = i;
            __syncthreads();
        }
    }
    // the costs are consumed: zero them for the next frame here (one packet less behind every frame that collects costs)
    if (zero) for (uint32_t i = tid; i < n; i += 1024u) zero[i] = 0u;
}

// The chunks of the overflow pool a whole tile takes in the frame this order is made for, from the two lowest bits of its cost word
// (grt_render_tile.hip kBagKeep1 / kBagKeep2: how full the fullest bag of any of its rays got): 1, 2 or 3 (a full bag); 0 = not known —
// a tile without a cost, or costs that are not the tile kernel's words at all (the cold frame's particle counts): the launch decides
// (RenderArgs::ovf_cls0).  enabled = 0 (GRT_OPT_OVF_CLASSES off): a full bag for everyone
__device__ __forceinline__ uint32_t bag_class(uint32_t enabled, uint32_t cost_word)
{
    if (!enabled) return 3u;
    if (cost_word == 0u) return 0u;
    const uint32_t d = cost_word & 3u;
    return d == 0u ? 1u : (d == 1u ? 2u : 3u);
}

// The same order with the heaviest tiles launched as PARTS (tile kernel, camera rays without meshes).  A frame takes at least
// max(L, W / R): L the longest tile (one wave walking its frontier and compositing sweep after sweep: ~0.9 ms on the 1 M scene, half
// the frame of one GPU and ALL of the frame of a rank that owns an eighth of the tiles), W the launch's total work, R the resident
// waves.  Splitting a tile into four waves of 4 x 4 pixels costs ~3 x its work and takes its latency to ~0.8 (a proxy covers half a
// tile: a quadrant's frustum still meets most of them), so it pays for the few tiles that ARE the frame and for no others: a tile
// whose cost in the previous frame (its own, not the dilated one that orders the launch) exceeds pct4 % of the heaviest tile's AND
// pct_load % of W / R runs as four waves (pct2: as two waves of 4 x 8 pixels; off by default — half a tile takes as long as the
// whole).  With a frame that is bound by its total work (1080p on one GPU, 4K) nothing is split.  Entry = unit | part << 28 |
// code << 30; the parts of a tile are consecutive; everything past the last entry is kOrderPad.  The launch has room for extra_cap
// entries beyond one per tile: they go to the heaviest cost classes first.
__global__ __launch_bounds__(1024) void k_cost_order_parts(const uint32_t* cost, const uint32_t* raw, uint32_t n,
                                                           uint32_t* __restrict__ order, uint32_t extra_cap, uint32_t pct2, uint32_t pct4,
                                                           uint32_t pct_load, uint32_t resident_waves, uint32_t* zero, uint32_t bag_classes) // (`zero` is `raw`, and `cost` too when the map is not dilated: none of the three is restrict)
{
    __shared__ uint32_t hist[128], h2[128], h4[128], cursor[128];
    __shared__ uint8_t ok4[128], ok2[128]; // the launch has room for this cost class's four-way / two-way parts
    __shared__ uint32_t s_t2, s_t4, s_total, s_max, s_ext, s_half;
    __shared__ unsigned long long s_sum;
    const uint32_t tid = threadIdx.x;
    if (tid < 128u) { hist[tid] = 0u; h2[tid] = 0u; h4[tid] = 0u; }
    if (tid == 0u) { s_max = 0u; s_sum = 0ull; }
    __syncthreads();
    {
        uint32_t mx = 0;
        unsigned long long sm = 0;
        // (one workgroup on one CU: a plainly written pass pays one load round trip per 1024 units; every pass of this kernel loads
        //  kOrdBatch x 1024 units before it touches them)
        for (uint32_t base = 0; base < n; base += kOrdBatch * 1024u) {
            uint32_t cv[kOrdBatch], rv[kOrdBatch];
#pragma unroll
            for (uint32_t k = 0; k < kOrdBatch; k++) {
                const uint32_t i = base + k * 1024u + tid;
                cv[k] = (i < n) ? cost[i] : 0u;
                rv[k] = (i < n) ? raw[i] : 0u;
            }
#pragma unroll
            for (uint32_t k = 0; k < kOrdBatch; k++) {
                if (base + k * 1024u + tid < n) {
                    atomicAdd(&hist[127u - cost_class(cost_eff(cv[k]))], 1u); // bucket 0 = heaviest
                    const uint32_t r = cost_eff(rv[k]);
                    mx = max(mx, r);
                    sm += r;
                }
            }
        }
        for (int off = 32; off > 0; off >>= 1) { mx = max(mx, (uint32_t)__shfl_xor((int)mx, off)); sm += __shfl_xor(sm, off); }
        if ((tid & 63u) == 0u) { atomicMax(&s_max, mx); atomicAdd(&s_sum, sm); }
    }
    __syncthreads();
    if (tid == 0u) {
        const uint64_t lmax = s_max, load = s_sum / max(resident_waves, 1u); // the longest tile; the work per resident wave
        const uint64_t floor_ = load * pct_load / 100u;
        s_t4 = pct4 ? (uint32_t)min(max(lmax * pct4 / 100u, floor_), (uint64_t)0xFFFFFFFEu) : 0xFFFFFFFFu;
        s_t2 = pct2 ? (uint32_t)min(max(lmax * pct2 / 100u, floor_), (uint64_t)0xFFFFFFFEu) : 0xFFFFFFFFu;
    }
    __syncthreads();
    const uint32_t t2 = s_t2, t4 = s_t4;
    for (uint32_t base = 0; base < n; base += kOrdBatch * 1024u) {
        uint32_t cv[kOrdBatch], rv[kOrdBatch];
#pragma unroll
        for (uint32_t k = 0; k < kOrdBatch; k++) {
            const uint32_t i = base + k * 1024u + tid;
            cv[k] = (i < n) ? cost[i] : 0u;
            rv[k] = (i < n) ? raw[i] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < kOrdBatch; k++) {
            if (base + k * 1024u + tid < n) {
                const uint32_t r = cost_eff(rv[k]), b = 127u - cost_class(cost_eff(cv[k]));
                if (r > t4) atomicAdd(&h4[b], 1u);
                else if (r > t2) atomicAdd(&h2[b], 1u);
            }
        }
    }
    __syncthreads();
    // Room for extra_cap entries beyond one per tile: handed out heaviest class first — four-way parts, then two-way ones (a class that
    // no longer fits four-way is tried two-way) — so that the tiles that bound the frame keep their parts whatever the lighter ones would
    // like (deterministic: same costs, same order).  When ALL the parts fit — nearly always: the room is a quarter of the tiles — every
    // class gets what it asks for and 128 threads say so at once; only otherwise does one thread walk the classes in turn.  (That walk,
    // three loops of 128 dependent LDS round trips, was 30 of this kernel's 50 us behind every frame of a moving camera.)
    if (tid == 0u) s_ext = 0u;
    __syncthreads();
    if (tid < 128u) {
        const uint32_t want = 3u * h4[tid] + h2[tid];
        if (want) atomicAdd(&s_ext, want);
    }
    __syncthreads();
    const bool two_way_ = s_t2 != 0xFFFFFFFFu; // (two-way parts are in use at all: only then may a four-way class fall back to them)
    if (s_ext <= extra_cap) { // wave-uniform, workgroup-uniform
        if (tid < 128u) { ok4[tid] = h4[tid] != 0u ? 1 : 0; ok2[tid] = h2[tid] != 0u ? 1 : 0; }
    } else if (tid == 0u) {
        uint32_t extras = 0;
        for (uint32_t b = 0; b < 128u; b++) {
            const uint32_t e4 = 3u * h4[b];
            ok4[b] = (e4 != 0u && extras + e4 <= extra_cap) ? 1 : 0;
            if (ok4[b]) extras += e4;
        }
        for (uint32_t b = 0; b < 128u; b++) {
            const uint32_t e2 = h2[b] + ((two_way_ && !ok4[b]) ? h4[b] : 0u);
            ok2[b] = (e2 != 0u && extras + e2 <= extra_cap) ? 1 : 0;
            if (ok2[b]) extras += e2;
        }
    }
    __syncthreads();
    // cursor[b] = entries of the classes before b (exactly what the scatter below writes: a gap would leave a stale entry in the
    // launch): an exclusive scan over the 128 classes by the first two waves
    if (tid < 128u) {
        const uint32_t b = tid;
        const uint32_t mine = hist[b] + (ok4[b] ? 3u * h4[b] : 0u) + (ok2[b] ? h2[b] + ((two_way_ && !ok4[b]) ? h4[b] : 0u) : 0u);
        uint32_t inc = mine; // inclusive scan within the wave
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)inc, off);
            if ((int)(tid & 63u) >= off) inc += up;
        }
        if (tid == 63u) s_half = inc; // total of classes 0..63
        cursor[b] = inc - mine;       // (classes 64..127: + s_half below)
        if (tid == 127u) s_total = inc;
    }
    __syncthreads();
    if (tid >= 64u && tid < 128u) cursor[tid] += s_half;
    if (tid == 0u) s_total += s_half;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += kOrdBatch * 1024u) {
        uint32_t cv[kOrdBatch], rv[kOrdBatch];
#pragma unroll
        for (uint32_t k = 0; k < kOrdBatch; k++) {
            const uint32_t i = base + k * 1024u + tid;
            cv[k] = (i < n) ? cost[i] : 0u;
            rv[k] = (i < n) ? raw[i] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < kOrdBatch; k++) { // (1024 units at a time, a barrier between them: the tiles of a class stay in screen order)
            const uint32_t i = base + k * 1024u + tid;
            if (i < n) {
                const uint32_t r = cost_eff(rv[k]), b = 127u - cost_class(cost_eff(cv[k]));
                const bool two_way = t2 != 0xFFFFFFFFu;
                const uint32_t code = (r > t4) ? (ok4[b] ? 2u : ((two_way && ok2[b]) ? 1u : 0u)) : ((r > t2 && ok2[b]) ? 1u : 0u);
                const uint32_t parts = code == 2u ? 4u : (code == 1u ? 2u : 1u);
                const uint32_t pos = atomicAdd(&cursor[b], parts);
                // (a whole tile's part field carries its bags' size class: the chunks of the pool it takes, 0 = not known)
                if (code == 0u) order[pos] = i | (bag_class(bag_classes, bag_classes == 2u ? 0u : cv[k]) << 28);
                else for (uint32_t q = 0; q < parts; q++) order[pos + q] = i | (q << 28) | (code << 30);
            }
            __syncthreads();
        }
    }
    for (uint32_t i = s_total + tid; i < n + extra_cap; i += 1024u) order[i] = kOrderPad;
    if (tid == 0u) { order[n + extra_cap] = s_total; order[n + extra_cap + 1u] = s_t4; order[n + extra_cap + 2u] = s_max; } // (diagnostics: GRT_DEBUG_LAUNCH)
    if (zero) for (uint32_t i = tid; i < n; i += 1024u) zero[i] = 0u; // (costs consumed; every thread is past its last read of them: the loop above ends in a barrier)
}

// ---- the same order made by SEVERAL workgroups (launches above kOrdMultiMin units: what is left of the one-workgroup kernel grows
//      with the units, 1.7 us per 1024, 249 us for the 129 600 tiles of a 4K frame, behind every frame of a moving camera).  Workgroup g
//      owns the units [g per, (g + 1) per) — a run in screen order — and the four kernels are the four phases of k_cost_order_parts:
//      A class counts per workgroup + the heaviest tile and the total work; B the counts of the units that want to be split, with the
//      thresholds those two give; C (one workgroup) room for the parts, the classes' first entries, every workgroup's first entry in
//      every class (screen order within a class is kept: workgroup after workgroup, 1024 units after 1024 units); D the scatter, the
//      padding and the zeroing of the consumed costs.  Same entries as the one-workgroup kernel writes (same classes, same codes, same
//      room rule; inside a class both keep runs of 1024 units in order and leave the order inside a run to the atomics).
//      Scratch (uint32): S[0] heaviest, S[2..3] total (64 bit), S[4] t2, S[5] t4, S[6] entries; S[8 + b] ok4, S[136 + b] ok2;
//      S[kOrdCnt + (g * 3 + k) * 128 + b] counts (k = 0 all, 1 two-way, 2 four-way); S[kOrdCur + g * 128 + b] first entry.
//      S[0], S[2], S[3] must be zero on entry: phase C leaves them so (the context zeroes the scratch once).
constexpr uint32_t kOrdMaxGroups = 256u, kOrdCnt = 512u, kOrdCur = kOrdCnt + kOrdMaxGroups * 3u * 128u;
__host__ __device__ inline uint32_t ord_scratch_words() { return kOrdCur + kOrdMaxGroups * 128u; }
__device__ __forceinline__ void ord_thresholds(const uint32_t* S, uint32_t pct2, uint32_t pct4, uint32_t pct_load, uint32_t resident_waves,
                                               uint32_t& t2, uint32_t& t4)
{
    const uint64_t lmax = S[0], sum = (uint64_t)S[2] | ((uint64_t)S[3] << 32), load = sum / max(resident_waves, 1u);
    const uint64_t floor_ = load * pct_load / 100u;
    t4 = pct4 ? (uint32_t)min(max(lmax * pct4 / 100u, floor_), (uint64_t)0xFFFFFFFEu) : 0xFFFFFFFFu;
    t2 = pct2 ? (uint32_t)min(max(lmax * pct2 / 100u, floor_), (uint64_t)0xFFFFFFFEu) : 0xFFFFFFFFu;
}
__global__ __launch_bounds__(1024) void k_ord_a(const uint32_t* __restrict__ cost, const uint32_t* __restrict__ raw, uint32_t n, uint32_t per,
                                                uint32_t* __restrict__ S)
{
    __shared__ uint32_t hist[128];
    __shared__ uint32_t s_max;
    __shared__ unsigned long long s_sum;
    const uint32_t tid = threadIdx.x, g = blockIdx.x, lo = g * per, hi = min(n, lo + per);
    if (tid < 128u) hist[tid] = 0u;
    if (tid == 0u) { s_max = 0u; s_sum = 0ull; }
    __syncthreads();
    uint32_t mx = 0;
    unsigned long long sm = 0;
    for (uint32_t i = lo + tid; i < hi; i += 1024u) {
        atomicAdd(&hist[127u - cost_class(cost_eff(cost[i]))], 1u); // bucket 0 = heaviest
        const uint32_t r = cost_eff(raw[i]);
        mx = max(mx, r);
        sm += r;
    }
    for (int off = 32; off > 0; off >>= 1) { mx = max(mx, (uint32_t)__shfl_xor((int)mx, off)); sm += __shfl_xor(sm, off); }
    if ((tid & 63u) == 0u && mx) { atomicMax(&s_max, mx); atomicAdd(&s_sum, sm); }
    __syncthreads();
    if (tid == 0u && s_max) { atomicMax(&S[0], s_max); atomicAdd((unsigned long long*)(S + 2), s_sum); } // (one pair of global atomics per workgroup)
    if (tid < 128u) S[kOrdCnt + (g * 3u + 0u) * 128u + tid] = hist[tid];
}
__global__ __launch_bounds__(1024) void k_ord_b(const uint32_t* __restrict__ cost, const uint32_t* __restrict__ raw, uint32_t n, uint32_t per,
                                                uint32_t pct2, uint32_t pct4, uint32_t pct_load, uint32_t resident_waves, uint32_t* __restrict__ S)
{
    __shared__ uint32_t h2[128], h4[128];
    const uint32_t tid = threadIdx.x, g = blockIdx.x, lo = g * per, hi = min(n, lo + per);
    if (tid < 128u) { h2[tid] = 0u; h4[tid] = 0u; }
    uint32_t t2, t4;
    ord_thresholds(S, pct2, pct4, pct_load, resident_waves, t2, t4);
    __syncthreads();
    for (uint32_t i = lo + tid; i < hi; i += 1024u) {
        const uint32_t r = cost_eff(raw[i]);
        if (r > t4) atomicAdd(&h4[127u - cost_class(cost_eff(cost[i]))], 1u);
        else if (r > t2) atomicAdd(&h2[127u - cost_class(cost_eff(cost[i]))], 1u);
    }
    __syncthreads();
    if (tid < 128u) { S[kOrdCnt + (g * 3u + 1u) * 128u + tid] = h2[tid]; S[kOrdCnt + (g * 3u + 2u) * 128u + tid] = h4[tid]; }
}
__global__ __launch_bounds__(1024) void k_ord_c(uint32_t n, uint32_t groups, uint32_t extra_cap, uint32_t pct2, uint32_t pct4, uint32_t pct_load,
                                                uint32_t resident_waves, uint32_t* __restrict__ S, uint32_t* __restrict__ order)
{
    __shared__ uint32_t hist[128], h2[128], h4[128], first[128], runtot[8][128];
    __shared__ uint8_t ok4[128], ok2[128];
    __shared__ uint32_t s_ext, s_half, s_total;
    const uint32_t tid = threadIdx.x;
    if (tid < 128u) { hist[tid] = 0u; h2[tid] = 0u; h4[tid] = 0u; }
    if (tid == 0u) s_ext = 0u;
    uint32_t t2, t4;
    ord_thresholds(S, pct2, pct4, pct_load, resident_waves, t2, t4);
    __syncthreads();
    { // class totals over the workgroups: thread = (class, slice of the workgroups)
        const uint32_t b = tid & 127u, sl = tid >> 7;
        uint32_t a0 = 0, a1 = 0, a2 = 0;
        for (uint32_t g = sl; g < groups; g += 8u) {
            a0 += S[kOrdCnt + (g * 3u + 0u) * 128u + b]; a1 += S[kOrdCnt + (g * 3u + 1u) * 128u + b]; a2 += S[kOrdCnt + (g * 3u + 2u) * 128u + b];
        }
        if (a0) atomicAdd(&hist[b], a0);
        if (a1) atomicAdd(&h2[b], a1);
        if (a2) atomicAdd(&h4[b], a2);
    }
    __syncthreads();
    if (tid < 128u) {
        const uint32_t want = 3u * h4[tid] + h2[tid];
        if (want) atomicAdd(&s_ext, want);
    }
    __syncthreads();
    const bool two_way = t2 != 0xFFFFFFFFu;
    if (s_ext <= extra_cap) { // every class gets the parts it asks for (see k_cost_order_parts)
        if (tid < 128u) { ok4[tid] = h4[tid] != 0u ? 1 : 0; ok2[tid] = h2[tid] != 0u ? 1 : 0; }
    } else if (tid == 0u) {
        uint32_t extras = 0;
        for (uint32_t b = 0; b < 128u; b++) {
            const uint32_t e4 = 3u * h4[b];
            ok4[b] = (e4 != 0u && extras + e4 <= extra_cap) ? 1 : 0;
            if (ok4[b]) extras += e4;
        }
        for (uint32_t b = 0; b < 128u; b++) {
            const uint32_t e2 = h2[b] + ((two_way && !ok4[b]) ? h4[b] : 0u);
            ok2[b] = (e2 != 0u && extras + e2 <= extra_cap) ? 1 : 0;
            if (ok2[b]) extras += e2;
        }
    }
    __syncthreads();
    if (tid < 128u) { // first entry of every class: exclusive scan by two waves
        const uint32_t b = tid;
        const uint32_t mine = hist[b] + (ok4[b] ? 3u * h4[b] : 0u) + (ok2[b] ? h2[b] + ((two_way && !ok4[b]) ? h4[b] : 0u) : 0u);
        uint32_t inc = mine;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)inc, off);
            if ((int)(tid & 63u) >= off) inc += up;
        }
        if (tid == 63u) s_half = inc;
        first[b] = inc - mine;
        if (tid == 127u) s_total = inc;
    }
    __syncthreads();
    if (tid >= 64u && tid < 128u) first[tid] += s_half;
    if (tid == 0u) s_total += s_half;
    __syncthreads();
    { // every workgroup's first entry in every class (the entries it will write: same expression as above, its own counts): thread =
      // (class, one of eight runs of workgroups) — the runs' totals first, then each run walks its workgroups
        const uint32_t b = tid & 127u, sl = tid >> 7;
        const uint32_t m4 = ok4[b] ? 3u : 0u, m2 = ok2[b] ? 1u : 0u, m42 = (ok2[b] && two_way && !ok4[b]) ? 1u : 0u;
        const uint32_t run = (groups + 7u) / 8u, g0 = min(groups, sl * run), g1 = min(groups, g0 + run);
        uint32_t tot = 0;
        for (uint32_t g = g0; g < g1; g++) {
            const uint32_t c0 = S[kOrdCnt + (g * 3u + 0u) * 128u + b], c2 = S[kOrdCnt + (g * 3u + 1u) * 128u + b], c4 = S[kOrdCnt + (g * 3u + 2u) * 128u + b];
            tot += c0 + m4 * c4 + m2 * c2 + m42 * c4;
        }
        runtot[sl][b] = tot;
        __syncthreads();
        uint32_t cur = first[b];
        for (uint32_t k = 0; k < sl; k++) cur += runtot[k][b];
        for (uint32_t g = g0; g < g1; g++) {
            const uint32_t c0 = S[kOrdCnt + (g * 3u + 0u) * 128u + b], c2 = S[kOrdCnt + (g * 3u + 1u) * 128u + b], c4 = S[kOrdCnt + (g * 3u + 2u) * 128u + b];
            S[kOrdCur + g * 128u + b] = cur;
            cur += c0 + m4 * c4 + m2 * c2 + m42 * c4;
        }
        if (tid < 128u) { S[8u + b] = ok4[b]; S[136u + b] = ok2[b]; }
    }
    if (tid == 0u) {
        S[4] = t2; S[5] = t4; S[6] = s_total;
        order[n + extra_cap] = s_total; order[n + extra_cap + 1u] = t4; order[n + extra_cap + 2u] = S[0]; // (diagnostics: GRT_DEBUG_LAUNCH)
        S[0] = 0u; S[2] = 0u; S[3] = 0u; // consumed: the next frame's phase A starts from zero
    }
}
__global__ __launch_bounds__(1024) void k_ord_d(const uint32_t* cost, const uint32_t* raw, uint32_t n, uint32_t per,
                                                uint32_t extra_cap, const uint32_t* __restrict__ S, uint32_t* __restrict__ order,
                                                uint32_t* zero, uint32_t bag_classes) // (`zero` is `raw`, and `cost` too when the map is not dilated: none of the three is restrict)
{
    __shared__ uint32_t cursor[128];
    __shared__ uint8_t ok4[128], ok2[128];
    const uint32_t tid = threadIdx.x, g = blockIdx.x, lo = g * per, hi = min(n, lo + per);
    if (tid < 128u) { cursor[tid] = S[kOrdCur + g * 128u + tid]; ok4[tid] = (uint8_t)S[8u + tid]; ok2[tid] = (uint8_t)S[136u + tid]; }
    const uint32_t t2 = S[4], t4 = S[5], total = S[6];
    __syncthreads();
    for (uint32_t base = lo; base < hi; base += 1024u) { // (1024 units at a time, a barrier between them: the tiles of a class stay in screen order)
        const uint32_t i = base + tid;
        if (i < hi) {
            const uint32_t rw = raw[i], cw = cost[i];
            const uint32_t r = cost_eff(rw), b = 127u - cost_class(cost_eff(cw));
            const bool two_way = t2 != 0xFFFFFFFFu;
            const uint32_t code = (r > t4) ? (ok4[b] ? 2u : ((two_way && ok2[b]) ? 1u : 0u)) : ((r > t2 && ok2[b]) ? 1u : 0u);
            const uint32_t parts = code == 2u ? 4u : (code == 1u ? 2u : 1u);
            const uint32_t pos = atomicAdd(&cursor[b], parts);
            if (code == 0u) order[pos] = i | (bag_class(bag_classes, bag_classes == 2u ? 0u : cw) << 28); // (the bags' size class: k_cost_order_parts)
            else for (uint32_t q = 0; q < parts; q++) order[pos + q] = i | (q << 28) | (code << 30);
        }
        __syncthreads();
    }
    for (uint32_t i = total + g * 1024u + tid; i < n + extra_cap; i += gridDim.x * 1024u) order[i] = kOrderPad;
    if (zero) for (uint32_t i = lo + tid; i < hi; i += 1024u) zero[i] = 0u; // (this workgroup's costs: its last read of them is behind the barrier above;
                                                                            //  `cost` may be the dilated copy, `zero` is the raw array phase A-D read as `raw`)
}

uint32_t order_scratch_bytes() { return ord_scratch_words() * (uint32_t)sizeof(uint32_t); }

// The four-way parts of a launch order as a list of their own, in the order's order (heaviest first): what the quad kernel
// (grt_render_tile.hip MODE 3) takes, one wave per entry.  The first kQuadListCap of them are listed and re-coded 2 -> 3 in the order, so
// that the camera-ray kernel leaves them alone; the rest stays as it is.  One workgroup, 1024 entries per round, ranks by ballot + a scan
// of the 16 waves' counts; it runs behind the ordering kernel, i.e. behind the frame whose costs made the order — not in front of the
// frame that uses it.
__global__ __launch_bounds__(1024) void k_quad_list(uint32_t* __restrict__ order, uint32_t n, uint32_t* __restrict__ list, uint32_t* __restrict__ count, uint32_t cap)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t base;
    const uint32_t tid = threadIdx.x, wv = tid >> 6, ln = tid & 63u;
    if (tid == 0u) base = 0u;
    __syncthreads();
    for (uint32_t b = 0; b < n; b += 1024u) {
        const uint32_t i = b + tid;
        const uint32_t e = (i < n) ? order[i] : kOrderPad;
        const bool is = (e != kOrderPad) && ((e >> 30) == 2u);
        const unsigned long long m = __ballot(is);
        if (ln == 0u) wsum[wv] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t off = base;
        for (uint32_t w = 0; w < wv; w++) off += wsum[w];
        off += (uint32_t)__popcll(m & ((1ull << ln) - 1ull));
        if (is && off < cap) {
            list[off] = e;
            order[i] = e | (3u << 30);
        }
        __syncthreads();
        if (tid == 0u) { uint32_t t = 0; for (uint32_t w = 0; w < 16u; w++) t += wsum[w]; base += t; }
        __syncthreads();
        if (base >= cap) break; // (uniform: every thread reads the same word behind the barrier)
    }
    if (tid == 0u) count[0] = min(base, cap);
}

int quad_part_list(uint32_t* d_order, uint32_t n_entries, uint32_t* d_list, uint32_t* d_count, uint32_t cap, hipStream_t stream, std::string* err)
{
    hipLaunchKernelGGL(k_quad_list, dim3(1), dim3(1024), 0, stream, d_order, n_entries, d_list, d_count, std::min(cap, kQuadListCap));
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (err) *err = std::string("quad_part_list: ") + hipGetErrorString(e);
        return GRT_ERR_HIP;
    }
    return GRT_OK;
}

int order_units_with_parts(const uint32_t* d_cost_order, const uint32_t* d_cost_raw, uint32_t* d_order, uint32_t n, uint32_t extra_cap,
                           uint32_t pct2, uint32_t pct4, uint32_t pct_load, uint32_t resident_waves, uint32_t* d_zero, uint32_t* d_scratch,
                           uint32_t multi_min, uint32_t bag_classes /* 0: a full bag for every tile; 1: d_cost_raw holds the tile kernel's cost words, their two lowest bits are the bags' depth; 2: it does not (nothing known) */, hipStream_t stream, std::string* err)
{
    if (n == 0) return GRT_OK;
    const uint32_t cls = bag_classes;
    if (d_scratch && n >= multi_min) { // several workgroups, four phases (see k_ord_a)
        uint32_t per = 2048u; // units per workgroup: two runs of 1024
        if ((n + per - 1u) / per > kOrdMaxGroups) per = (((n + kOrdMaxGroups - 1u) / kOrdMaxGroups) + 1023u) & ~1023u;
        const uint32_t groups = (n + per - 1u) / per;
        hipLaunchKernelGGL(k_ord_a, dim3(groups), dim3(1024), 0, stream, d_cost_order, d_cost_raw, n, per, d_scratch);
        hipLaunchKernelGGL(k_ord_b, dim3(groups), dim3(1024), 0, stream, d_cost_order, d_cost_raw, n, per, pct2, pct4, pct_load, resident_waves, d_scratch);
        hipLaunchKernelGGL(k_ord_c, dim3(1), dim3(1024), 0, stream, n, groups, extra_cap, pct2, pct4, pct_load, resident_waves, d_scratch, d_order);
        hipLaunchKernelGGL(k_ord_d, dim3(groups), dim3(1024), 0, stream, d_cost_order, d_cost_raw, n, per, extra_cap, (const uint32_t*)d_scratch, d_order, d_zero, cls);
    } else
    hipLaunchKernelGGL(k_cost_order_parts, dim3(1), dim3(1024), 0, stream, d_cost_order, d_cost_raw, n, d_order, extra_cap, pct2, pct4, pct_load,
                       resident_waves, d_zero, cls);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (err) *err = std::string("order_units_with_parts: ") + hipGetErrorString(e);
        return GRT_ERR_HIP;
    }
    return GRT_OK;
}

// Under a MOVING camera a heavy tile of the previous frame is a slightly different tile of this one: every 8x8 tile
// takes the largest cost within `radius` tiles of itself, so the neighbourhood of last frame's heavy tiles starts
// early too.  Units are (16x16 block, quadrant): block b = by * nbx + bx, quadrant q -> tile (2 bx + (q & 1), 2 by + (q >> 1)).
__global__ void k_cost_dilate(const uint32_t* __restrict__ cost, uint32_t* __restrict__ out, uint32_t nbx, uint32_t nby, int radius)
{
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= nbx * nby * 4u) return;
    const uint32_t b = u >> 2, q = u & 3u;
    const int tx = (int)((b % nbx) * 2u + (q & 1u)), ty = (int)((b / nbx) * 2u + (q >> 1));
    const int ntx = (int)nbx * 2, nty = (int)nby * 2;
    uint32_t m = 0, deep = 0; // (deep: the two lowest bits of the tile kernel's cost words — how full the bags got, 0 / 1 / 3 — dilate too:
                              //  the size class a tile starts with under a moving camera is its neighbourhood's deepest)
    for (int dy = -radius; dy <= radius; dy++)
        for (int dx = -radius; dx <= radius; dx++) {
            const int x = tx + dx, y = ty + dy;
            if (x < 0 || y < 0 || x >= ntx || y >= nty) continue;
            const uint32_t v = cost[(((uint32_t)y >> 1) * nbx + ((uint32_t)x >> 1)) * 4u + (((uint32_t)y & 1u) << 1) + ((uint32_t)x & 1u)];
            m = max(m, cost_eff(v));
            deep = max(deep, v & 3u);
        }
    out[u] = m >= 4u ? ((m & ~3u) | deep) : m;
}

int dilate_unit_costs(const uint32_t* d_cost, uint32_t* d_out, uint32_t nbx, uint32_t nby, int radius, hipStream_t stream,
                      std::string* err)
{
    const uint32_t n = nbx * nby * 4u;
    if (n == 0) return GRT_OK;
    hipLaunchKernelGGL(k_cost_dilate, dim3((n + 255u) / 256u), dim3(256), 0, stream, d_cost, d_out, nbx, nby, radius);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (err) *err = std::string("dilate_unit_costs: ") + hipGetErrorString(e);
        return GRT_ERR_HIP;
    }
    return GRT_OK;
}

int order_units_by_cost(const uint32_t* d_cost, uint32_t* d_order, uint32_t n, uint32_t heavy_cap, uint32_t thr_x2,
                        uint32_t* d_n_heavy, uint32_t* d_zero, hipStream_t stream, std::string* err)
{
    if (n == 0) return GRT_OK;
    hipLaunchKernelGGL(k_cost_order, dim3(1), dim3(1024), 0, stream, d_cost, n, d_order, heavy_cap, thr_x2, d_n_heavy, d_zero);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (err) *err = std::string("order_units_by_cost: ") + hipGetErrorString(e);
        return GRT_ERR_HIP;
    }
    return GRT_OK;
}

// 4-wide view: wide record i = the (up to four) grandchildren of binary node i; a child that is a leaf range stays
// as it is.  Internal refs keep pointing at binary node indices, each of which has its own wide record.
__global__ void k_widen(const float4* __restrict__ nodes, int m, float4* __restrict__ wn)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m - 1) return;
    const float4 q0 = nodes[(size_t)i * 4], q1 = nodes[(size_t)i * 4 + 1], q2 = nodes[(size_t)i * 4 + 2],
                 q3 = nodes[(size_t)i * 4 + 3];
    float box[4][6];
    uint32_t ref[4];
    int ne = 0;
    const uint32_t c[2] = {__float_as_uint(q3.x), __float_as_uint(q3.y)};
    const float cb[2][6] = {{q0.x, q0.y, q0.z, q0.w, q1.x, q1.y}, {q1.z, q1.w, q2.x, q2.y, q2.z, q2.w}};
    for (int k = 0; k < 2; k++) {
        if (c[k] & kLeafBit) {
            for (int j = 0; j < 6; j++) box[ne][j] = cb[k][j];
            ref[ne++] = c[k];
        } else {
            const size_t b = (size_t)c[k] * 4;
            const float4 g0 = nodes[b], g1 = nodes[b + 1], g2 = nodes[b + 2], g3 = nodes[b + 3];
            const float gb[2][6] = {{g0.x, g0.y, g0.z, g0.w, g1.x, g1.y}, {g1.z, g1.w, g2.x, g2.y, g2.z, g2.w}};
            for (int j = 0; j < 6; j++) box[ne][j] = gb[0][j];
            ref[ne++] = __float_as_uint(g3.x);
            for (int j = 0; j < 6; j++) box[ne][j] = gb[1][j];
            ref[ne++] = __float_as_uint(g3.y);
        }
    }
    for (; ne < 4; ne++) { // unused child: ref kNoRoot (the traversal tests the ref; the slab test is symmetric in lo/hi)
        box[ne][0] = box[ne][1] = box[ne][2] = 1.0f;
        box[ne][3] = box[ne][4] = box[ne][5] = -1.0f;
        ref[ne] = kNoRoot;
    }
    float4* w = wn + (size_t)i * 8;
    // per child: (lo.x, lo.y), (hi.x, hi.y), (lo.z, hi.z) — 8-byte pairs the traversal can swap with scalar ops
#define GRT_WB(c, j) box[c][(j) == 0 ? 0 : (j) == 1 ? 1 : (j) == 2 ? 3 : (j) == 3 ? 4 : (j) == 4 ? 2 : 5]
    w[0] = make_float4(GRT_WB(0, 0), GRT_WB(0, 1), GRT_WB(0, 2), GRT_WB(0, 3));
    w[1] = make_float4(GRT_WB(0, 4), GRT_WB(0, 5), GRT_WB(1, 0), GRT_WB(1, 1));
    w[2] = make_float4(GRT_WB(1, 2), GRT_WB(1, 3), GRT_WB(1, 4), GRT_WB(1, 5));
    w[3] = make_float4(GRT_WB(2, 0), GRT_WB(2, 1), GRT_WB(2, 2), GRT_WB(2, 3));
    w[4] = make_float4(GRT_WB(2, 4), GRT_WB(2, 5), GRT_WB(3, 0), GRT_WB(3, 1));
    w[5] = make_float4(GRT_WB(3, 2), GRT_WB(3, 3), GRT_WB(3, 4), GRT_WB(3, 5));
#undef GRT_WB
    w[6] = make_float4(__uint_as_float(ref[0]), __uint_as_float(ref[1]), __uint_as_float(ref[2]), __uint_as_float(ref[3]));
    w[7] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// Per-child layout for the tile kernel (one lane tests one child box: 2 x 16-B loads), kTileWide children per node:
// record i = the descendants of binary node i up to log2(kTileWide) levels down (a leaf range stays as it is, and an
// internal descendant stops being expanded once the record is full).  Child c at [i*2W + 2c] = (lo.xyz, ref bits),
// [i*2W + 2c + 1] = (hi.xyz, 0); unused child: ref kNoRoot with an inverted box.
__global__ void k_qwiden(const float4* __restrict__ nodes, int m, float4* __restrict__ qn, int area_only)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m - 1) return;
    constexpr int W = (int)kTileWide;
    float box[W][6];
    uint32_t ref[W];
    int ne = 0;
    auto push_children = [&](uint32_t node, float (*b)[6], uint32_t* r, int& n) {
        const size_t q = (size_t)node * 4;
        const float4 q0 = nodes[q], q1 = nodes[q + 1], q2 = nodes[q + 2], q3 = nodes[q + 3];
        b[n][0] = q0.x; b[n][1] = q0.y; b[n][2] = q0.z; b[n][3] = q0.w; b[n][4] = q1.x; b[n][5] = q1.y;
        r[n++] = __float_as_uint(q3.x);
        b[n][0] = q1.z; b[n][1] = q1.w; b[n][2] = q2.x; b[n][3] = q2.y; b[n][4] = q2.z; b[n][5] = q2.w;
        r[n++] = __float_as_uint(q3.y);
    };
    push_children((uint32_t)i, box, ref, ne);
    // (round 4, log item 19): open the internal entry with the LARGEST box first, again and again, until the record is full —
    // up to W - 2 openings along any path instead of two rounds over all entries: the big boxes a tile's frustum is most likely to
    // meet are opened here, once, instead of costing the traversal a step each.  Default: the two rounds first, then what is left
    // of the record is filled that way.
    auto fill_by_area = [&]() {
    while (ne < W) {
        int best = -1;
        float ba = -1.0f;
        for (int e = 0; e < ne; e++) {
            if (ref[e] & kLeafBit) continue;
            const float dx = box[e][3] - box[e][0], dy = box[e][4] - box[e][1], dz = box[e][5] - box[e][2];
            const float ar = dx * dy + dy * dz + dz * dx;
            if (ar > ba) { ba = ar; best = e; }
        }
        if (best < 0) break;
        const uint32_t nd = ref[best];
        for (int e = best; e + 1 < ne; e++) { // close the gap, the two children go to the end
            for (int k = 0; k < 6; k++) box[e][k] = box[e + 1][k];
            ref[e] = ref[e + 1];
        }
        ne--;
        push_children(nd, box, ref, ne);
    }
    };
    // (area_only: trees with pieces — large overlapping boxes, a depth-serial traversal: the needle scene C3a -4.3 % with the largest box
    //  opened first throughout, -0.4 % with the fill; the compact scenes lose with it: C2 +14 %)
    if (area_only) fill_by_area();
    for (int round = 1; !area_only && (2 << round) <= W; round++) { // each round expands every internal entry that still fits
        float nb[W][6];
        uint32_t nr[W];
        int nn = 0;
        const int internal_left_init = [&] { int k = 0; for (int e = 0; e < ne; e++) k += (ref[e] & kLeafBit) ? 0 : 1; return k; }();
        int internal_left = internal_left_init;
        for (int e = 0; e < ne; e++) {
            const bool internal = !(ref[e] & kLeafBit);
            // expanding turns 1 entry into 2: allowed while the final count (entries so far + the rest + 1) fits
            const int rest = ne - e - 1;
            if (internal && (nn + 2 + rest) <= W) {
                push_children(ref[e], nb, nr, nn);
            } else {
                for (int k = 0; k < 6; k++) nb[nn][k] = box[e][k];
                nr[nn++] = ref[e];
            }
            (void)internal_left;
        }
        for (int e = 0; e < nn; e++) {
            for (int k = 0; k < 6; k++) box[e][k] = nb[e][k];
            ref[e] = nr[e];
        }
        ne = nn;
    }
    if (!area_only) fill_by_area();
    float4* q = qn + (size_t)i * 2 * W;
    for (int e = 0; e < W; e++) {
        if (e < ne) {
            q[2 * e] = make_float4(box[e][0], box[e][1], box[e][2], __uint_as_float(ref[e]));
            q[2 * e + 1] = make_float4(box[e][3], box[e][4], box[e][5], INFINITY); // (.w: bounding radius, none for a node)
        } else {
            q[2 * e] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(kNoRoot));
            q[2 * e + 1] = make_float4(-1.0f, -1.0f, -1.0f, 0.0f);
        }
    }
}

__global__ void k_pbox(const float4* __restrict__ lb_lo, const float4* __restrict__ lb_hi, uint32_t m,
                       float4* __restrict__ pbox)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    pbox[(size_t)j * 2] = lb_lo[j];
    pbox[(size_t)j * 2 + 1] = lb_hi[j];
}

void free_bvh(DevBvh* b)
{
    if (b->qnodes) (void)hipFree(b->qnodes);
    if (b->pbox) (void)hipFree(b->pbox);
    if (b->level) (void)hipFree(b->level);
    if (b->wnodes) (void)hipFree(b->wnodes);
    if (b->nodes) (void)hipFree(b->nodes);
    if (b->order) (void)hipFree(b->order);
    *b = DevBvh();
}

// ---- refit: same hierarchy, new boxes ----
__device__ __forceinline__ void refit_child_box(uint32_t c, const float4* __restrict__ nodes, const float4* __restrict__ lb_lo,
                                                const float4* __restrict__ lb_hi, float lo[3], float hi[3])
{
    if (c & kLeafBit) { // leaf range: union of its primitives' boxes
        const uint32_t first = leaf_first(c), cnt = leaf_count(c);
        for (int k = 0; k < 3; k++) { lo[k] = INFINITY; hi[k] = -INFINITY; }
        for (uint32_t j = 0; j < cnt; j++) {
            const float4 l = lb_lo[first + j], h = lb_hi[first + j];
            lo[0] = fminf(lo[0], l.x); lo[1] = fminf(lo[1], l.y); lo[2] = fminf(lo[2], l.z);
            hi[0] = fmaxf(hi[0], h.x); hi[1] = fmaxf(hi[1], h.y); hi[2] = fmaxf(hi[2], h.z);
        }
        return;
    }
    const float4 q0 = nodes[(size_t)c * 4], q1 = nodes[(size_t)c * 4 + 1], q2 = nodes[(size_t)c * 4 + 2];
    lo[0] = fminf(q0.x, q1.z); lo[1] = fminf(q0.y, q1.w); lo[2] = fminf(q0.z, q2.x);
    hi[0] = fmaxf(q0.w, q2.y); hi[1] = fmaxf(q1.x, q2.z); hi[2] = fmaxf(q1.y, q2.w);
}

// nodes finished in build pass `pass` get their boxes again, from children finished in earlier passes (each launch is
// one level, so every hand-off crosses a kernel boundary as in the build)
__global__ void k_refit_level(float4* __restrict__ nodes, const uint32_t* __restrict__ level, int m, uint32_t pass,
                              const float4* __restrict__ lb_lo, const float4* __restrict__ lb_hi)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m - 1 || level[i] != pass) return;
    const float4 q3 = nodes[(size_t)i * 4 + 3];
    float l0[3], h0[3], l1[3], h1[3];
    refit_child_box(__float_as_uint(q3.x), nodes, lb_lo, lb_hi, l0, h0);
    refit_child_box(__float_as_uint(q3.y), nodes, lb_lo, lb_hi, l1, h1);
    nodes[(size_t)i * 4 + 0] = make_float4(l0[0], l0[1], l0[2], h0[0]);
    nodes[(size_t)i * 4 + 1] = make_float4(h0[1], h0[2], l1[0], l1[1]);
    nodes[(size_t)i * 4 + 2] = make_float4(l1[2], h1[0], h1[1], h1[2]);
}

int refit_lbvh(const float4* d_lo, const float4* d_hi, uint32_t n_in, DevBvh* bvh, hipStream_t stream, std::string* err)
{
    float4 *d_lblo = nullptr, *d_lbhi = nullptr;
    const uint32_t m = bvh->n_prims;
    const int B = 256;
    if (m == 0 || (bvh->root_ref & kLeafBit)) return GRT_OK; // no internal nodes: the leaves are tested directly
    if (!bvh->level || !bvh->nodes || n_in < m) {
        if (err) *err = "refit_lbvh: the BVH was not built with keep_levels";
        return GRT_ERR_INVALID;
    }
    HIPCHK(hipMalloc(&d_lblo, sizeof(float4) * m));
    HIPCHK(hipMalloc(&d_lbhi, sizeof(float4) * m));
    hipLaunchKernelGGL(k_leaf_boxes, dim3((m + B - 1) / B), dim3(B), 0, stream, d_lo, d_hi, bvh->order, m, d_lblo, d_lbhi);
    for (uint32_t pass = 1; pass <= bvh->height; pass++)
        hipLaunchKernelGGL(k_refit_level, dim3((m - 1 + B - 1) / B), dim3(B), 0, stream, bvh->nodes, bvh->level, (int)m, pass,
                           d_lblo, d_lbhi);
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipGetLastError());
    (void)hipFree(d_lblo); (void)hipFree(d_lbhi);
    return GRT_OK;
fail:
    (void)hipFree(d_lblo); (void)hipFree(d_lbhi);
    return GRT_ERR_HIP;
}

int build_lbvh(const float4* d_lo, const float4* d_hi, uint32_t n_in, uint32_t leaf_max, bool want_quad, bool keep_levels,
               int size_classes, DevBvh* out, hipStream_t stream, std::string* err, bool widen_area_only, int rotation_sweeps)
{
    uint2* d_range = nullptr;
    uint32_t* d_bounds = nullptr;
    uint64_t *d_keys = nullptr, *d_keys2 = nullptr;
    uint32_t *d_vals = nullptr, *d_level = nullptr;
    float4 *d_lblo = nullptr, *d_lbhi = nullptr;
    void* d_tmp = nullptr;
    size_t tmp_bytes = 0;
    uint32_t h_bounds[8];
    uint32_t m = 0;
    const int B = 256;

    out->n_prims = 0;
    out->height = 0;
    out->root_ref = kNoRoot;
    if (n_in == 0) return GRT_OK;
    leaf_max = std::max(1u, std::min(leaf_max, kLeafMaxPrims));
    if (n_in > kLeafIndexMask) {
        if (err) *err = "build_lbvh: more than 2^28 primitives";
        return GRT_ERR_LIMIT;
    }

    HIPCHK(hipMalloc(&d_bounds, (8 + 256) * sizeof(uint32_t)));
    {
        const uint32_t init[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0, 0};
        HIPCHK(hipMemcpyAsync(d_bounds, init, sizeof(init), hipMemcpyHostToDevice, stream));
    }
    HIPCHK(hipMalloc(&d_keys, sizeof(uint64_t) * n_in));
    HIPCHK(hipMalloc(&d_keys2, sizeof(uint64_t) * n_in));
    HIPCHK(hipMalloc(&d_vals, sizeof(uint32_t) * n_in));
    if (out->cap_order < n_in) {
        if (out->order) (void)hipFree(out->order);
        out->order = nullptr;
        out->cap_order = 0;
        HIPCHK(hipMalloc(&out->order, sizeof(uint32_t) * n_in));
        out->cap_order = n_in;
    }
    {
        const int grid = (int)std::min<uint32_t>((n_in + B - 1) / B, 256u);
        hipLaunchKernelGGL(k_scene_bounds, dim3(grid), dim3(B), 0, stream, d_lo, d_hi, n_in, d_bounds);
        hipLaunchKernelGGL(k_scene_bounds_finish, dim3(1), dim3(1), 0, stream, d_bounds, (uint32_t)grid);
        hipLaunchKernelGGL(k_morton, dim3((n_in + B - 1) / B), dim3(B), 0, stream, d_lo, d_hi, n_in, d_bounds, d_keys,
                           d_vals, want_quad ? size_classes : 0);
    }
    // (d_vals2: the sort ping-pongs between two pairs of arrays; out->order receives the sorted indices)
    tmp_bytes = sizeof(uint32_t) * (256u * ((n_in + kRsTile - 1u) / kRsTile) + 1u);
    HIPCHK(hipMalloc(&d_tmp, tmp_bytes));
    {
        uint32_t* d_bad = static_cast<uint32_t*>(d_tmp) + 256u * ((n_in + kRsTile - 1u) / kRsTile);
        uint32_t h_bad = 0;
        HIPCHK(hipMemsetAsync(d_bad, 0, sizeof(uint32_t), stream));
        HIPCHK(radix_sort_pairs_64_32(d_keys, d_keys2, d_vals, out->order, n_in, static_cast<uint32_t*>(d_tmp), stream));
        hipLaunchKernelGGL(k_rs_check, dim3((n_in + B - 1) / B), dim3(B), 0, stream, d_keys2, out->order, n_in, d_bad);
        HIPCHK(hipMemcpyAsync(&h_bad, d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemcpyAsync(h_bounds, d_bounds, sizeof(h_bounds), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if (h_bad) {
            if (err) *err = "build_lbvh: internal error: the Morton sort is not sorted / not stable";
            goto fail_limit;
        }
    }
    m = h_bounds[6];
    out->n_prims = m;
    for (int k = 0; k < 3; k++) {
        out->lo[k] = m ? ord2f(h_bounds[k]) : 0.f;
        out->hi[k] = m ? ord2f(h_bounds[3 + k]) : 0.f;
    }
    if (out->qnodes) (void)hipFree(out->qnodes);
    if (out->pbox) (void)hipFree(out->pbox);
    if (out->level) (void)hipFree(out->level);
    out->qnodes = out->pbox = nullptr;
    out->level = nullptr;
    if (m) {
        HIPCHK(hipMalloc(&d_lblo, sizeof(float4) * m));
        HIPCHK(hipMalloc(&d_lbhi, sizeof(float4) * m));
        hipLaunchKernelGGL(k_leaf_boxes, dim3((m + B - 1) / B), dim3(B), 0, stream, d_lo, d_hi, out->order, m, d_lblo,
                           d_lbhi);
        if (want_quad) {
            HIPCHK(hipMalloc(&out->pbox, sizeof(float4) * 2 * (size_t)m + 256));
            hipLaunchKernelGGL(k_pbox, dim3((m + B - 1) / B), dim3(B), 0, stream, d_lblo, d_lbhi, m, out->pbox);
        }
    }
    if (m <= leaf_max) { // empty, or the whole scene is one leaf range: no node arrays (drop stale ones of an earlier build)
        if (out->wnodes) (void)hipFree(out->wnodes);
        if (out->nodes) (void)hipFree(out->nodes);
        out->wnodes = out->nodes = nullptr;
        out->cap_nodes = 0;
        if (m) out->root_ref = kLeafBit | ((m - 1u) << 28);
        HIPCHK(hipStreamSynchronize(stream));
        goto done;
    }
    if (out->cap_nodes < (size_t)(m - 1)) {
        if (out->nodes) (void)hipFree(out->nodes);
        out->nodes = nullptr;
        out->cap_nodes = 0;
        HIPCHK(hipMalloc(&out->nodes, sizeof(float4) * 4 * (size_t)(m - 1)));
        out->cap_nodes = m - 1;
    }
    HIPCHK(hipMalloc(&d_level, sizeof(uint32_t) * (m - 1)));
    HIPCHK(hipMalloc(&d_range, sizeof(uint2) * (m - 1)));
    HIPCHK(hipMemsetAsync(d_level, 0, sizeof(uint32_t) * (m - 1), stream));
    hipLaunchKernelGGL(k_hierarchy, dim3((m - 1 + B - 1) / B), dim3(B), 0, stream, d_keys2, (int)m, out->nodes, d_range);
    {
        uint32_t pass = 0, root_level = 0;
        while (root_level == 0) {
            pass++;
            if (pass > 4096) {
                if (err) *err = "build_lbvh: refit did not converge";
                goto fail_limit;
            }
            hipLaunchKernelGGL(k_refit_pass, dim3((m - 1 + B - 1) / B), dim3(B), 0, stream, out->nodes, d_level,
                               (int)m, pass, d_lblo, d_lbhi);
            // poll every few passes only (a 4-byte readback syncs the stream)
            if (pass >= 8 && (pass & 3) == 0) {
                HIPCHK(hipMemcpyAsync(&root_level, d_level, 4, hipMemcpyDeviceToHost, stream));
                HIPCHK(hipStreamSynchronize(stream));
            } else if (pass < 8 && m <= (1u << pass)) {
                HIPCHK(hipMemcpyAsync(&root_level, d_level, 4, hipMemcpyDeviceToHost, stream));
                HIPCHK(hipStreamSynchronize(stream));
            }
        }
        out->height = root_level;
        out->root_ref = 0;
    }
    if (out->wnodes) (void)hipFree(out->wnodes);
    out->wnodes = nullptr;
    HIPCHK(hipMalloc(&out->wnodes, sizeof(float4) * 8 * (size_t)(m - 1) + 256));
    if (leaf_max > 1)
        hipLaunchKernelGGL(k_collapse, dim3((m - 1 + B - 1) / B), dim3(B), 0, stream, out->nodes, d_range, (int)m, leaf_max);
    // Bottom-up sweeps of rotations, for the Gaussian BVH of a scene with pieces (widen_area_only: needles and sheets — where large
    // and small boxes overlap and a sweep pays: C3a 14.56 -> 13.77 ms; compact scenes: C3 / C2 flat, C5 -1 %, two of the three dense-core
    // cameras +3 / +13 %: left alone).  GRT_OPT_BVH_ROTATIONS (rotation_sweeps): -1 = that choice with one sweep, 0 = none, n = n sweeps on
    // any tree.  Behind EVERY sweep the levels are numbered afresh for the topology as it stands (k_level_pass) and the height is the
    // root's new level: a swap pushes the sibling one level down, so the refit's height is no bound for the rotated tree (it sizes the
    // per-lane kernel's LDS stack and decides tile_stack_fits / the wave kernel's depth test), and a sweep by stale levels lets two
    // threads of one pass rewrite one record (round 5 saw that as changed pixels and shipped a single sweep on the old levels).
    {
        const int sweeps = want_quad ? (rotation_sweeps < 0 ? (widen_area_only ? 1 : 0) : std::min(rotation_sweeps, 8)) : 0;
        for (int sw = 0; sw < sweeps; sw++) {
            for (uint32_t pass = 2; pass <= out->height; pass++)
                hipLaunchKernelGGL(k_rotate_pass, dim3((m - 1 + B - 1) / B), dim3(B), 0, stream, out->nodes, d_level, (int)m, pass, (uint32_t*)nullptr);
            HIPCHK(hipMemsetAsync(d_level, 0, sizeof(uint32_t) * (m - 1), stream));
            uint32_t pass = 0, root_level = 0;
            const uint32_t h_old = out->height;
            while (root_level == 0) {
                pass++;
                if (pass > 4096) {
                    if (err) *err = "build_lbvh: levels of the rotated tree did not converge";
                    goto fail_limit;
                }
                hipLaunchKernelGGL(k_level_pass, dim3((m - 1 + B - 1) / B), dim3(B), 0, stream, out->nodes, d_level, (int)m, pass);
                // a sweep changes the height by little: look first where the old height says the root may be done, then every 4 passes
                if (pass + 2 >= h_old && ((pass + 2 - h_old) & 3u) == 0u) {
                    HIPCHK(hipMemcpyAsync(&root_level, d_level, 4, hipMemcpyDeviceToHost, stream));
                    HIPCHK(hipStreamSynchronize(stream));
                }
            }
            out->height = root_level;
        }
    }
    hipLaunchKernelGGL(k_widen, dim3((m - 1 + B - 1) / B), dim3(B), 0, stream, out->nodes, (int)m, out->wnodes);
    if (want_quad) {
        HIPCHK(hipMalloc(&out->qnodes, sizeof(float4) * 2 * kTileWide * (size_t)(m - 1) + 256));
        hipLaunchKernelGGL(k_qwiden, dim3((m - 1 + B - 1) / B), dim3(B), 0, stream, out->nodes, (int)m, out->qnodes, widen_area_only ? 1 : 0);
    }
    HIPCHK(hipStreamSynchronize(stream));
    if (keep_levels) { out->level = d_level; d_level = nullptr; }
done:
    HIPCHK(hipGetLastError());
    (void)hipFree(d_bounds); (void)hipFree(d_keys); (void)hipFree(d_keys2); (void)hipFree(d_vals);
    (void)hipFree(d_tmp); (void)hipFree(d_lblo); (void)hipFree(d_lbhi); (void)hipFree(d_level); (void)hipFree(d_range);
    return GRT_OK;
fail_limit:
    (void)hipFree(d_bounds); (void)hipFree(d_keys); (void)hipFree(d_keys2); (void)hipFree(d_vals);
    (void)hipFree(d_tmp); (void)hipFree(d_lblo); (void)hipFree(d_lbhi); (void)hipFree(d_level); (void)hipFree(d_range);
    return GRT_ERR_LIMIT;
fail:
    (void)hipFree(d_bounds); (void)hipFree(d_keys); (void)hipFree(d_keys2); (void)hipFree(d_vals);
    (void)hipFree(d_tmp); (void)hipFree(d_lblo); (void)hipFree(d_lbhi); (void)hipFree(d_level); (void)hipFree(d_range);
    return GRT_ERR_HIP;
}

}  // namespace grt
