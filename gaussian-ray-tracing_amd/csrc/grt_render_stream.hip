// grt_render_stream.hip — single-pass, wave-cooperative "streaming" render kernel (gfx950).
//
// Same contract and per-lane arithmetic as grt_render_wave.hip (bit-identical results), but one workgroup is ONE
// wave64 = one 8x8 pixel tile, and the k = 7 re-traversal rounds of trace() (shaders/tracer.cuh:341-369) are replaced
// by ONE front-to-back pass over the 4-wide view of the LBVH (grt_bvh.hip: k_widen):
//   * records are fetched once per wave with scalar loads: a 128-B wide node, or a 64-B proxy record plus the 16-B
//     per-eye record that holds what depends on the ray origin only (grt_api.hip: k_eye_records);
//   * the wave expands the tree BEST-FIRST: the frontier (unexpanded subtrees) lives in wave registers — slot i is
//     lane i of one (lambda, ref) VGPR pair, 64 slots — keyed by lambda = the smallest box-entry distance over the
//     lanes that want the subtree; pop = DPP min-reduction + ballot + v_readlane.  A full frontier spills to a
//     depth-first stack in LDS that is drained first (finality bound unchanged);
//   * when a subtree with key lambda is popped, no unseen hit of any lane can have t < lambda, so every buffered hit
//     event with t < lambda is FINAL and is composited immediately, in key order (t, particle id, entry<exit) —
//     exactly the order the rounds would have produced;
//   * each lane buffers PARTICLES, not hits: 12 sorted 64-bit keys in registers (payload: exit t and alpha in LDS
//     cells named by the key's low bits); compositing an entry event re-keys the slot to its exit event.  The sorted
//     insert / pop are generated inline assembly (gen_slots.py -> grt_slots_gen.inc);
//   * a lane whose window overflows records the smallest key it had to drop (`cutoff`), keeps compositing below it,
//     and — only if it still has transmittance left at the end of the pass — starts another pass from its last
//     composited key (the reference's "next round", now the exception, not the rule);
//   * lanes stop wanting subtrees once T <= minTransmittance, so the pass ends early for opaque tiles.
// The body (grt_render_stream_body.inc) is instantiated twice: 12 slots at 4 waves/SIMD, and 32 slots at 2 waves/SIMD
// for the tiles the scheduling feedback marks as heavy.  DESIGN.md §5.2 has the measurements behind each choice.
//
// Style note: the per-lane slot file and the wave-level frontier are plain local scalars driven by macros
// (no structs passed by reference, no loop-indexed arrays): that is what keeps all of it in registers —
// struct/array forms of the same code were left in scratch memory by the compiler.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

#include "grt_device.h"
#include "grt_internal.h"

namespace grt {

namespace {

constexpr int kBlock = 256; // threads of a 16x16 screen block (the unit of RenderArgs::n_blocks, order[], cost[])
constexpr int kWG = 64;     // workgroup = ONE wave: a finished wave frees its slot at once instead of waiting
                            // for the slowest of four (tile costs within a block differ a lot)
constexpr uint64_t kCellMask = 31ull; // payload-cell bits of a slot key
__device__ __forceinline__ uint64_t mk_skey(float t, uint32_t id, uint32_t is_exit)
{
    return ((uint64_t)__float_as_uint(t) << 32) | (uint64_t)((id << 6) | (is_exit << 5));
}
__device__ __forceinline__ uint32_t skey_id(uint64_t k) { return ((uint32_t)k) >> 6; }

struct Cnt {
    uint32_t rays = 0, segments = 0, hit_evals = 0, rounds = 0, node_visits = 0, proxy_tests = 0, fetches = 0;
};

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

// wave64 min of non-negative floats (or +inf) -> wave-uniform value.  Their bit patterns order like unsigned
// integers, so the reduction is 4 v_min_u32 with DPP operands inside rows of 16, then 4 v_readlane + 3 s_min_u32
// (no NaN canonicalisation, which fminf would add to every step).
__device__ __forceinline__ float wave_min(float v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t x = __float_as_uint(v);
    // bound_ctrl:1 lets the compiler fold each DPP move into the v_min_u32 itself (one VALU op per step)
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, true)); // row_half_mirror
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xF, 0xF, true)); // row_mirror
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)x, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)x, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)x, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)x, 48);
    return __uint_as_float(min(min(a, b), min(c, d)));
#else
    return v;
#endif
}

// four independent wave minima in lockstep: the DPP steps of different reductions interleave, so the two wait states a
// DPP operand needs after its producer are filled with useful work instead of s_nop
__device__ __forceinline__ void wave_min4(float a, float b, float c, float d, float& ra, float& rb, float& rc, float& rd)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t x0 = __float_as_uint(a), x1 = __float_as_uint(b), x2 = __float_as_uint(c), x3 = __float_as_uint(d);
#define GRT_DPP_STEP(CTRL)                                                                                 \
    {                                                                                                      \
        const uint32_t y0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x0, CTRL, 0xF, 0xF, true);       \
        const uint32_t y1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x1, CTRL, 0xF, 0xF, true);       \
        const uint32_t y2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x2, CTRL, 0xF, 0xF, true);       \
        const uint32_t y3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x3, CTRL, 0xF, 0xF, true);       \
        x0 = min(x0, y0); x1 = min(x1, y1); x2 = min(x2, y2); x3 = min(x3, y3);                            \
    }
    GRT_DPP_STEP(0xB1)  // quad_perm [1,0,3,2]
    GRT_DPP_STEP(0x4E)  // quad_perm [2,3,0,1]
    GRT_DPP_STEP(0x141) // row_half_mirror
    GRT_DPP_STEP(0x140) // row_mirror
#undef GRT_DPP_STEP
#define GRT_ROWS(x)                                                                                        \
    __uint_as_float(min(min((uint32_t)__builtin_amdgcn_readlane((int)x, 0), (uint32_t)__builtin_amdgcn_readlane((int)x, 16)), \
                        min((uint32_t)__builtin_amdgcn_readlane((int)x, 32), (uint32_t)__builtin_amdgcn_readlane((int)x, 48))))
    ra = GRT_ROWS(x0); rb = GRT_ROWS(x1); rc = GRT_ROWS(x2); rd = GRT_ROWS(x3);
#undef GRT_ROWS
#else
    ra = a; rb = b; rc = c; rd = d;
#endif
}

// votes straight on the lane mask (the __any/__ballot wrappers go through an int and cost two extra VALU ops)
__device__ __forceinline__ uint64_t wave_ballot(bool p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(p);
#else
    return p ? 1ull : 0ull;
#endif
}
__device__ __forceinline__ bool wave_any(bool p) { return wave_ballot(p) != 0ull; }
// max(v, +0) for a non-NaN float as one integer max (negative floats are negative ints)
__device__ __forceinline__ float clamp0(float v) { return __int_as_float(max(__float_as_int(v), 0)); }

// Profiling build (make WPROF=1, never shipped): the counters then hold WAVE-level trip counts of the kernel's
// sections instead of per-lane event counts — rays: pops, node_visits: wide nodes, fetches: leaf particles,
// proxy_tests: exact slab tests run, segments: insert blocks, hit_evals: compositing steps, rounds: re-key blocks.
#ifdef GRT_WPROF
#define GRT_W_DECL Cnt w; const uint64_t w_t0 = wall_clock64();
#define GRT_W(f) w.f++;
#define GRT_W_FLUSH                                                                                        \
    c = (lane == 0) ? w : Cnt();                                                                           \
    if (a.outf && lane == 0 && write) { /* timeline: (start, end) in 100 MHz ticks and the hardware id */  \
        uint32_t hwid_;                                                                                    \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid_));                                \
        a.outf[out_idx * 3] = __uint_as_float((uint32_t)w_t0);                                             \
        a.outf[out_idx * 3 + 1] = __uint_as_float((uint32_t)wall_clock64());                               \
        a.outf[out_idx * 3 + 2] = __uint_as_float(hwid_);                                                  \
    }
#else
#define GRT_W_DECL
#define GRT_W(f)
#define GRT_W_FLUSH
#endif

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); } // v_pk_fma_f32

// two wave-uniform floats as one 64-bit scalar (an aligned SGPR pair) and back
__device__ __forceinline__ uint64_t pack2(float a, float b)
{
    return (uint64_t)__float_as_uint(a) | ((uint64_t)__float_as_uint(b) << 32);
}
__device__ __forceinline__ v2f unpack2(uint64_t u)
{
    return v2f{__uint_as_float((uint32_t)u), __uint_as_float((uint32_t)(u >> 32))};
}

// default kernel: 12-particle window, 4 waves per SIMD (128 VGPRs; a handful spill to scratch)
#define GRT_KS 12
#ifndef GRT_DEF_WAVES
#define GRT_DEF_WAVES 4
#endif
#define GRT_WAVES GRT_DEF_WAVES
#define GRT_KERNEL_NAME k_render_stream
#include "grt_render_stream_body.inc"
#undef GRT_KS
#undef GRT_WAVES
#undef GRT_KERNEL_NAME

// big-window kernel: 32-particle window, 2 waves per SIMD (no spills) — for the blocks marked heavy
#ifndef GRT_BIG_KS
#define GRT_BIG_KS 32
#endif
#define GRT_KS GRT_BIG_KS
#define GRT_WAVES 2
#define GRT_KERNEL_NAME k_render_stream_big
#include "grt_render_stream_body.inc"
#undef GRT_KS
#undef GRT_WAVES
#undef GRT_KERNEL_NAME

} // namespace

typedef void (*StreamKernel)(const RenderArgs);
static StreamKernel pick(bool big, bool count, bool sh, bool mesh)
{
#define GRT_PICK(K)                                                                                        \
    (count ? (sh ? (mesh ? K<true, true, true> : K<true, true, false>) : (mesh ? K<true, false, true> : K<true, false, false>)) \
           : (sh ? (mesh ? K<false, true, true> : K<false, true, false>) : (mesh ? K<false, false, true> : K<false, false, false>)))
    return big ? GRT_PICK(k_render_stream_big) : GRT_PICK(k_render_stream);
#undef GRT_PICK
}

int launch_render_stream(const RenderArgs& a, bool count, bool mesh, hipStream_t stream, const LaunchAux* aux,
                         std::string* err)
{
    if (a.n_blocks == 0) return GRT_OK;
    const bool sh = a.p.sh_degree_max > 0;
    hipError_t e = hipSuccess;
    const bool split = aux && aux->aux && aux->heavy_cap && a.order && a.n_heavy;
    if (!split) {
        RenderArgs b = a;
        b.heavy_role = 0;
        hipLaunchKernelGGL(pick(aux && aux->force_big, count, sh, mesh), dim3(a.n_blocks * 4u), dim3(kWG), 0, stream, b);
    } else {
        // heavy blocks (the first *n_heavy ranks of the cost-sorted order) on the big-window kernel, on a second
        // stream so that both launches share the GPU; the main stream joins it before anything else runs
        RenderArgs h = a, n = a;
        h.heavy_role = 1;
        n.heavy_role = 2;
        e = hipEventRecord(aux->fork, stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(aux->aux, aux->fork, 0);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(pick(true, count, sh, mesh), dim3(std::min(aux->heavy_cap, a.n_blocks * 4u)), dim3(kWG), 0,
                               aux->aux, h);
            e = hipEventRecord(aux->join, aux->aux);
        }
        if (e == hipSuccess) {
            hipLaunchKernelGGL(pick(false, count, sh, mesh), dim3(a.n_blocks * 4u), dim3(kWG), 0, stream, n);
            e = hipStreamWaitEvent(stream, aux->join, 0);
        }
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) {
        if (err) *err = std::string("k_render_stream launch: ") + hipGetErrorString(e);
        return GRT_ERR_HIP;
    }
    return GRT_OK;
}

} // namespace grt
