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
#include "grt_wave.h"

namespace grt {

namespace {

constexpr int kBlock = 256; // threads of a 16x16 screen block (the unit of RenderArgs::n_blocks, order[], cost[])
constexpr int kWG = 64;     // workgroup = ONE wave: a finished wave frees its slot at once instead of waiting
                            // for the slowest of four (tile costs within a block differ a lot)
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
#elif defined(GRT_MARKS)
// ISA-budget build (profiles/isa_budget.py, never shipped): each section start leaves a comment in the assembly
#define GRT_W_DECL
#define GRT_W(f) asm volatile("; GRT_MARK " #f);
#define GRT_W_FLUSH
#else
#define GRT_W_DECL
#define GRT_W(f)
#define GRT_W_FLUSH
#endif

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
