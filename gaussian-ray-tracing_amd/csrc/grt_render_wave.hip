// grt_render_wave.hip — wave-cooperative render kernel for coherent (camera) rays, gfx950.
//
// A wave64 owns an 8x8 pixel tile and walks the Gaussian LBVH ONCE for all 64 rays:
//   * the traversal state (current node, stack) is wave-uniform and lives in SGPRs; the stack is a
//     "wave-register stack": entry i is lane i of two VGPRs (v_writelane / v_readlane), no LDS, no scratch;
//   * node and proxy records are fetched with SCALAR loads (s_load_dwordx4 x4 through the constant
//     address space): one 64-B fetch per wave instead of 64 divergent gathers, and the record then
//     feeds every lane's VALU as SGPR operands;
//   * every lane tests the two child boxes against ITS ray and ITS current interval
//     (t_lo, k-th nearest so far); a child is entered when any lane wants it (64-bit ballot), the
//     nearer one by majority vote;
//   * at a leaf every live lane runs the exact proxy test and inserts into its own register k-buffer.
// The per-lane arithmetic (slab test, response, keys, blending) is the same code as the per-lane
// kernel (grt_device.h), so results are bit-identical to it; only the set of culled boxes differs,
// and boxes never decide a hit.  Scenes with meshes and ray-buffer input use grt_render.hip.
#include <hip/hip_runtime.h>

#include <string>

#include "grt_device.h"
#include "grt_internal.h"

namespace grt {

namespace {

constexpr int K = 7; // MaxNumHitPerTrace, shaders/tracer.cuh:11
constexpr int kBlock = 256;

struct Cnt {
    uint32_t rays = 0, segments = 0, hit_evals = 0, rounds = 0, node_visits = 0, proxy_tests = 0, fetches = 0, iters = 0;
};

struct KBuf {
    uint64_t key[K];
    float alpha[K];
};

// Branch-free insertion (pure selects): a key of ~0 (kKeyInvalid) leaves the buffer untouched.  Same result
// as the 7 compare-and-swap steps of __anyhit__anyhit (shaders/tracer.cu:124-146).  Straight-line code keeps
// the 21 buffer registers updated in place instead of being copied at every control-flow join.
__device__ __forceinline__ void kbuf_insert(KBuf& kb, uint64_t key, float alpha)
{
#pragma unroll
    for (int i = 0; i < K; i++) {
        const bool lt = key < kb.key[i];
        const uint64_t tk = kb.key[i];
        const float ta = kb.alpha[i];
        kb.key[i] = lt ? key : tk;
        kb.alpha[i] = lt ? alpha : ta;
        key = lt ? tk : key;
        alpha = lt ? ta : alpha;
    }
}

// is the event already buffered?  (a split particle is met once per piece the ray's tile crosses, with the same keys)
__device__ __forceinline__ bool kbuf_has(const KBuf& kb, uint64_t key)
{
    bool h = false;
#pragma unroll
    for (int i = 0; i < K; i++) h = h || (kb.key[i] == key);
    return h;
}

// scalar (SGPR) fetch of one float4 at a wave-uniform index: constant address space => s_load_dwordx4
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 sload4(const float4* base, uint32_t idx)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) v4f* cptr4;
    const v4f v = ((cptr4)(uintptr_t)base)[idx];
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return base[idx];
#endif
}

__device__ __forceinline__ uint32_t stack_pop(uint32_t s0, uint32_t s1, uint32_t sp)
{
    return sp < 64u ? (uint32_t)__builtin_amdgcn_readlane((int)s0, (int)sp)
                    : (uint32_t)__builtin_amdgcn_readlane((int)s1, (int)(sp - 64u));
}

template <bool COUNT>
__device__ __forceinline__ void gps_round_wave(const RenderArgs& a, f3 o, f3 d, const rayinv& ri, bool alive,
                                               uint64_t last_key, float t_hi, KBuf& kb, Cnt& c)
{
#pragma unroll
    for (int i = 0; i < K; i++) {
        kb.key[i] = kKeyInvalid;
        kb.alpha[i] = 0.0f;
    }
    const float t_lo = key_t(last_key);
    float bound = t_hi;
    uint32_t s0 = 0, s1 = 0; // wave-register stack
    const uint32_t lane_id = threadIdx.x & 63u;
    uint32_t sp = 0;
    uint32_t cur = a.root_ref;
    while (true) {
        cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur);
        c.iters++;
        if (cur & kLeafBit) {
            const uint32_t first = leaf_first(cur), cnt = leaf_count(cur);
            for (uint32_t j = 0; j < cnt; j++) {
                const uint32_t idx = (first + j) * 4u;
                const float4 r0 = sload4(a.rec, idx), r1 = sload4(a.rec, idx + 1), r2 = sload4(a.rec, idx + 2),
                             r3 = sload4(a.rec, idx + 3);
                if (COUNT) c.fetches += 4; // 64-B record, in 16-B units
                const f3 mu = mk3(r0.x, r0.y, r0.z);
                m33 A;
                A.a[0] = r1.x; A.a[1] = r1.y; A.a[2] = r1.z;
                A.a[3] = r2.x; A.a[4] = r2.y; A.a[5] = r2.z;
                A.a[6] = r3.x; A.a[7] = r3.y; A.a[8] = r3.z;
                const f3 o_g = matvec(A, sub3(o, mu));
                const f3 d_g = matvec(A, d);
                if (!__any(alive && proxy_sphere_maybe(o_g, d_g, r0.w))) continue; // no lane can touch this proxy
                if (COUNT && alive) c.proxy_tests++;
                float te, tx;
                const bool hit = proxy_slabs(o_g, d_g, r0.w, te, tx) && alive;
                const uint32_t id = __float_as_uint(r2.w);
                const uint64_t ke = mk_key(te, id, 0), kx = mk_key(tx, id, 1);
                // te/tx may be negative or NaN: the float compares gate the (unsigned) key compares
                const uint32_t cellb = __float_as_uint(r3.w); // piece of a split proxy: an event belongs to the cell its point lies in
                const bool in_e = hit && (te >= t_lo) && (te < t_hi) && (ke > last_key) && (ke < kb.key[K - 1]) && !kbuf_has(kb, ke) &&
                                  (!cellb || piece_owns(cellb, r0.w, o_g, d_g, te));
                const bool in_x = hit && (tx >= t_lo) && (tx < t_hi) && (kx > last_key) && (kx < kb.key[K - 1]) && !kbuf_has(kb, kx) &&
                                  (!cellb || piece_owns(cellb, r0.w, o_g, d_g, tx));
                if (__any(in_e || in_x)) { // wave-uniform branch
                    // alpha does not depend on the hit distance (shaders/tracer.cuh:354-357)
                    const float alpha = fminf(0.99f, response_from(A, mu, o, d, o_g, d_g) * r1.w);
                    kbuf_insert(kb, in_e ? ke : kKeyInvalid, alpha);
                    kbuf_insert(kb, in_x ? kx : kKeyInvalid, alpha);
                    bound = (kb.key[K - 1] != kKeyInvalid) ? key_t(kb.key[K - 1]) : bound;
                }
            }
            if (sp == 0) break;
            --sp;
            cur = stack_pop(s0, s1, sp);
        } else {
            const uint32_t idx = cur * 4u;
            const float4 q0 = sload4(a.nodes, idx), q1 = sload4(a.nodes, idx + 1), q2 = sload4(a.nodes, idx + 2),
                         q3 = sload4(a.nodes, idx + 3);
            if (COUNT) { c.fetches += 4; if (alive) c.node_visits++; }
            float n0, f0, n1, f1;
            box_interval(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, ri, n0, f0);
            box_interval(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, ri, n1, f1);
            const bool h0 = alive && (n0 <= f0) && (f0 >= t_lo) && (n0 <= bound);
            const bool h1 = alive && (n1 <= f1) && (f1 >= t_lo) && (n1 <= bound);
            const uint64_t m0 = __ballot(h0), m1 = __ballot(h1);
            const uint32_t c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y);
            if (m0 && m1) {
                // nearer child first, decided by the lanes that care about the order
                const uint64_t v0 = __ballot(h0 && (!h1 || n0 <= n1));
                const uint64_t v1 = __ballot(h1 && (!h0 || n1 < n0));
                const bool first0 = __popcll(v0) >= __popcll(v1);
                const uint32_t far = first0 ? c1 : c0;
                // push: lane sp of the stack register takes the (uniform) value
                if (sp < 64u) s0 = (lane_id == sp) ? far : s0;
                else s1 = (lane_id == sp - 64u) ? far : s1;
                ++sp;
                cur = first0 ? c0 : c1;
            } else if (m0) {
                cur = c0;
            } else if (m1) {
                cur = c1;
            } else {
                if (sp == 0) break;
                --sp;
                cur = stack_pop(s0, s1, sp);
            }
        }
    }
}

// trace() — shaders/tracer.cuh:328-373 — for the whole wave; lanes that are done idle in `alive`
template <bool COUNT, bool SH>
__device__ __forceinline__ void trace_gaussians_wave(const RenderArgs& a, bool have_ray, f3 o, f3 d, float t_min,
                                                     float t_max, float& density, f3& radiance, Cnt& c)
{
    float T = 1.0f - density;
    const float epsT = 1e-9f;
    float lastT = t_min;
    radiance = mk3(0.0f, 0.0f, 0.0f);
    if (COUNT && have_ray) c.segments++;
    if (a.root_ref == kNoRoot) return;
    const f3 dn = normalize3(d);
    const rayinv ri = mk_rayinv(o, d);
    uint64_t last_key = mk_key(lastT + epsT, 0x7FFFFFFFu, 1);
    const float t_hi = t_max + epsT;
    const float minT = a.p.minTransmittance;
    KBuf kb;
    bool alive = have_ray && (lastT <= t_max) && (T > minT);
    while (__any(alive)) {
        gps_round_wave<COUNT>(a, o, d, ri, alive, last_key, t_hi, kb, c);
        if (alive) {
            if (COUNT) c.rounds++;
            if (kb.key[0] == kKeyInvalid) {
                alive = false;
            } else {
#pragma unroll
                for (int i = 0; i < K; i++) {
                    if (kb.key[i] != kKeyInvalid && T > minT) {
                        if (COUNT) c.hit_evals++;
                        lastT = fmaxf(key_t(kb.key[i]), lastT);
                        const float hitAlpha = kb.alpha[i];
                        if (a.p.alpha_min < hitAlpha) {
                            const uint32_t id = key_id(kb.key[i]);
                            f3 L;
                            if (!SH) {
                                const float4 cc = a.color0[id];
                                L = mk3(cc.x, cc.y, cc.z);
                            } else {
                                L = sh_radiance(a.sh + (size_t)id * 48, dn, a.p.sh_degree_max);
                            }
                            radiance = add3(radiance, mul3s(mul3s(L, T), hitAlpha));
                            T *= (1.0f - hitAlpha);
                        }
                    }
                }
                if (kb.key[K - 1] == kKeyInvalid) alive = false;
                else {
                    last_key = kb.key[K - 1];
                    alive = (lastT <= t_max) && (T > minT);
                }
            }
        }
    }
    density = 1.0f - T;
}


// no-mesh frames: raygen -> miss -> LastGaussianPass -> write (shaders/tracer.cu:17-110 with mesh_handle == 0)
template <bool COUNT, bool SH>
__global__ __launch_bounds__(kBlock) void k_render_wave(const RenderArgs a)
{
    Cnt c;
    const uint32_t blk = a.order ? a.order[blockIdx.x] : xcd_swizzle(blockIdx.x, a.n_blocks, a.swizzle_chunk);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t lx = (wave & 1u) * 8u + (lane & 7u), ly = (wave >> 1) * 8u + (lane >> 3);
    uint32_t px, py;
    size_t out_idx;
    bool in_frame;
    if (a.mode == 0) {
        px = a.x0 + (blk % a.nbx) * 16u + lx;
        py = a.y0 + (blk / a.nbx) * 16u + ly;
        in_frame = (px < a.x1) && (py < a.y1);
        out_idx = (size_t)py * a.p.width + px;
    } else {
        const uint32_t per_tile = a.nbx * a.nby;
        const uint32_t j = blk / per_tile, sub = blk % per_tile;
        const uint32_t tile = a.first_tile + j * a.tile_stride;
        const uint32_t tx = tile % a.tiles_x, ty = tile / a.tiles_x;
        const uint32_t ox = (sub % a.nbx) * 16u + lx, oy = (sub / a.nbx) * 16u + ly;
        px = tx * a.tile_w + ox;
        py = ty * a.tile_h + oy;
        in_frame = (px < a.p.width) && (py < a.p.height);
        out_idx = ((size_t)j * a.tile_h + oy) * a.tile_w + ox;
    }
    const bool write = in_frame || (a.mode == 1);
    const f3 nU = mk3(-a.p.U[0], -a.p.U[1], -a.p.U[2]), nV = mk3(-a.p.V[0], -a.p.V[1], -a.p.V[2]);
    const f3 W = mk3(a.p.W[0], a.p.W[1], a.p.W[2]);
    const f3 eye = mk3(a.p.eye[0], a.p.eye[1], a.p.eye[2]);
    f3 dir = mk3(0.0f, 0.0f, -1.0f);
    bool have_ray = in_frame;
    if (in_frame) {
        if (!a.p.mode_fisheye) get_ray(px, py, nU, nV, W, a.p.width, a.p.height, dir);
        else have_ray = get_fisheye_ray(px, py, nU, nV, W, a.p.width, a.p.height, dir);
    }
    if (COUNT && have_ray) c.rays++;
    // bounce loop of shaders/tracer.cu:58-106 with every traceMesh a miss
    have_ray = have_ray && (length3(dir) > 0.1f) && (a.p.max_bounces > 0u);
    float density = 0.0f;
    f3 rad;
    trace_gaussians_wave<COUNT, SH>(a, have_ray, eye, dir, a.p.t_min, a.p.t_max, density, rad, c);
    f3 col = mk3(0.0f, 0.0f, 0.0f);
    if (have_ray) {
        const float alpha = density;
        const f3 directLight = mul3s(rad, alpha);                        // shaders/tracer.cu:80
        col = add3(col, mul3s(directLight, 1.0f - 0.0f));                // shaders/tracer.cu:101, blocking == 0
    }
    if (write) {
        if (a.outf) {
            a.outf[out_idx * 3] = col.x; a.outf[out_idx * 3 + 1] = col.y; a.outf[out_idx * 3 + 2] = col.z;
        }
        if (a.out8) {
            a.out8[out_idx * 3] = quantize8(col.x);
            a.out8[out_idx * 3 + 1] = quantize8(col.y);
            a.out8[out_idx * 3 + 2] = quantize8(col.z);
        }
    }
    if (a.cost && lane == 0) atomicMax(&a.cost[blk], c.iters);
    if (COUNT) {
        // fetches are wave-level events: count them once per wave
        uint32_t v[7] = {c.rays, c.segments, c.hit_evals, c.rounds, c.node_visits, c.proxy_tests, 0};
#pragma unroll
        for (int k = 0; k < 6; k++) {
            uint32_t x = v[k];
            for (int off = 32; off > 0; off >>= 1) x += (uint32_t)__shfl_xor((int)x, off);
            if (lane == 0 && x) atomicAdd(&a.counters[k], (unsigned long long)x);
        }
        if (lane == 0 && c.fetches) atomicAdd(&a.counters[6], (unsigned long long)c.fetches);
    }
}

} // namespace

int launch_render_wave(const RenderArgs& a, bool count, hipStream_t stream, std::string* err)
{
    if (a.n_blocks == 0) return GRT_OK;
    const bool sh = a.p.sh_degree_max > 0;
    auto fn = count ? (sh ? k_render_wave<true, true> : k_render_wave<true, false>)
                    : (sh ? k_render_wave<false, true> : k_render_wave<false, false>);
    hipLaunchKernelGGL(fn, dim3(a.n_blocks), dim3(kBlock), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (err) *err = std::string("k_render_wave launch: ") + hipGetErrorString(e);
        return GRT_ERR_HIP;
    }
    return GRT_OK;
}

} // namespace grt
