// grt_render_tile.hip — tile kernel: one wave64 = one 8x8 pixel tile, BVH culling done per CHILD BOX (gfx950).
//
// The streaming kernel (grt_render_stream.hip) walks the tree with lanes = rays: every popped 4-wide node costs the
// wave four 64-lane box tests, four wave reductions and scalar push logic (~230 VALU + ~170 SALU per node,
// profiles/r02_isa_budget.json) although the 64 rays of a tile are almost parallel and nearly always agree.  This
// kernel turns that part around:
//
//   * CULLING IS DONE WITH LANES = CHILD BOXES.  The tile's rays share the eye and lie inside a thin frustum (four
//     planes through the eye, from wave reductions over the lanes' directions).  A node step takes the (<= 8) nearest
//     unexpanded 8-wide nodes off the frontier at once; lane l loads child (l % 8) of node (l / 8) — two 16-B vector
//     loads — and tests that ONE box against the frustum (conservative; culling only) and computes a lower bound lambda
//     of the hit distance of ANY ray of the tile inside it.  64 boxes per step for ~200 VALU instead of 4 boxes for ~230.
//     A leaf step expands the (<= 16) nearest leaf ranges the same way into their (<= 4) particles, whose boxes sit in
//     pbox[], and slab-tests the survivors at once.
//   * the frontier (unexpanded subtrees and leaf ranges) lives in one (lambda, ref) register pair, slot i = lane i;
//     children are compacted into free slots through a 512-B LDS exchange buffer (rank = v_mbcnt of the ballot).  What
//     does not fit goes to a 256-entry bag in LDS; a rebalance keeps the nearest entries of (registers + bag) in
//     registers; only a full bag falls back to a depth-first stack in LDS.
//   * F = the smallest lambda on the frontier (and in the bag) is the FINALITY bound exactly as in the streaming kernel:
//     no unseen event of any lane can have t < F, so buffered events below F are composited in key order
//     (t, id, entry<exit).
//   * exact work keeps lanes = rays and the streaming kernel's arithmetic, operation for operation: a surviving particle
//     is fetched by scalar loads (64-B record + 64-B eye record) and slab-tested by all lanes; hits go into the per-lane
//     sorted window (12 keys in registers, payload cells in LDS; the same generated EXEC-masked insert/shift macros); a
//     window that overflows spills to the lane's bag in global memory (refilled by a scan; a full bag keeps its nearer
//     half) and only what is lost for good costs the lane another pass.  Frames are therefore bit-identical to the
//     other kernels'.
//   * compositing is deferred until enough lanes have a final event (or a window is about to overflow): one
//     compositing step costs the same whether 1 or 64 lanes take part.
//
// MODE 0: camera rays (window / tile modes), with or without the mesh wavefront pipeline (MESH = true: the primary
// segment ends at the per-lane mesh hit and the rays that go on are written to the continuation queue, one 64-entry chunk
// per tile).  MODE 1 / MODE 2: the bounced rays of mesh frames, as per-tile bundles / one ray per wave (below).
// DESIGN.md 5.2 and 5.5 have the numbers.  Citations (file:line) are into Ray-Studio2/gaussian-ray-tracing.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

#include "grt_device.h"
#include "grt_internal.h"
#include "grt_mesh.h"
#include "grt_wave.h"

namespace grt {

namespace {

constexpr int kBlock = 256; // threads of a 16x16 screen block (the unit of RenderArgs::n_blocks)
constexpr int kWG = 64;     // one wave per workgroup, as in the streaming kernel
constexpr uint32_t kLeafLanes = (uint32_t)kTileLeafMax; // lanes per leaf range in a leaf step (grt_internal.h)
constexpr uint32_t kBatch = 64u / kLeafLanes; // leaf ranges per leaf step (x kLeafLanes particles = 64 lanes)
constexpr float kSweepEagerT = 0.5f; // a ready lane below this transmittance keeps a compositing sweep going on its own
constexpr uint32_t kBag = 256u;  // far frontier entries parked in LDS (4 per lane when they are rebalanced)
constexpr uint32_t kStack = kTileStack; // depth-first overflow stack (only when the LDS bag is full too; guarded; the launcher
                                  // admits only trees it can hold: tile_stack_fits).  LDS per wave must stay <= 10 KB:
                                  // 10304 B gave 15 waves per CU instead of 16 and cost 4 %
constexpr uint32_t kKeep = 40u; // frontier entries kept in registers by a rebalance (the nearest ones)
constexpr uint32_t kOvf = kTileOvfEntries; // entries (16 B, global memory) a lane's overflow bag holds at most; the capacity in use is
                                           // a.ovf_entries (<= kOvf; smaller only in tests)
constexpr uint32_t kSub = kTileOvfSub;     // The pool is handed out in CHUNKS of kSub entries x 64 lanes (32 KiB).  A tile STARTS in one,
                                           // two or three in a row (three = a full bag per ray) by how deep its bags got in the frame
                                           // before: the cost word's two lowest bits say how full the fullest bag of any of its rays got
                                           // (0: at most kBagKeep1 entries, 1: at most kBagKeep2, 3: more), the launch order hands the
                                           // size class back in the part field of a whole tile's entry (grt_bvh.hip); a tile without a
                                           // cost word (a cold frame) starts in one.  A tile that outgrows its chunks MOVES to three
                                           // fresh ones (its rays' entries are copied: a wave-level loop, rare) — no class is ever a limit.
                                           // Of the tiles of the 1 M scene that overflow at all, half never hold more than 16 entries in
                                           // any bag, 79 % never more than 32, 95 % never more than 48.
constexpr uint32_t kBagKeep1 = 20u, kBagKeep2 = 48u; // (a class's bags are pruned 8 entries short of full: at 24 and 56)
static_assert(kOvf == 3u * kSub && kBagKeep1 < kSub && kBagKeep2 < 2u * kSub, "a full bag is three chunks");
constexpr int kBisect = 18;          // most bisection steps of a nearest-k selection (4 / 6 at least)
constexpr uint32_t kPruneRoom = 32u; // a window bag with less room than this is pruned between steps
constexpr int kWavesPerSimd = 4;     // waves per SIMD the camera-ray and bundle kernels are compiled for (128 VGPRs)
constexpr int kWavesQuad = 3;        // ... the quad kernel (MODE 3: per-lane records in the exact test; its launches are bound by their longest wave, not by occupancy)
// The cost word of a camera-ray tile counts, besides its steps, 2/8 of a step per particle fetched and 12/8 per exact test run: the
// launch order and the part-wave policy live on that word, and steps alone are a poor proxy of a tile's TIME (a leaf step that carries
// sixteen ranges through their exact tests and a node step count the same).  Same-box kernel ms with it: C1 0.513 -> 0.472, C2 0.955 ->
// 0.89, a rank of eight 0.757 -> 0.737; other weightings and the wave's own clock: profiles/r04_experiments_log.md 19.
constexpr uint32_t kCostFetch = 4u, kCostTest = 24u; // (in sixteenths of a step)
static_assert(kCostFetch % 4u == 0u && kCostTest % 4u == 0u, "the two lowest bits of a camera-ray tile's `work` are its bags' depth class");

// The only compile-time variants of this file: GRT_TILE_KS = 8 with GRT_TILE_SINGLE_TU (grt_render_tile_single.hip: the one-ray-per-wave
// mode as a translation unit of its own, 3 waves per SIMD), GRT_TILE_DIAG (wave-level trip counts in the counters), GRT_TILE_CHECK
// (invariant checks), GRT_MARKS (section marks in the assembly, for the ISA budget).  tests/test_isa_lint.py compiles each of them.
// The switches of experiments that lost live on as profiles/tools/r04_experiments_removed.patch.
#ifndef GRT_TILE_WAVES2
#define GRT_TILE_WAVES2 2 /* waves per SIMD of the one-ray-per-wave kernel (MODE 2; 19 KB of LDS per wave at 12 keys: 8 waves per CU) */
#endif
#ifndef GRT_TILE_KS
#define GRT_TILE_KS 12 /* keys of a lane's sorted window: 12, or 8 */
#endif
#define GRT_KS GRT_TILE_KS
#define KS GRT_TILE_KS
#if GRT_TILE_KS == 12
#define KLAST k11
#define KPRESS k9 /* a lane holding >= KS-2 keys asks for compositing before the next insert */

#define KROOM k8  /* a lane with room above this slot joins a refill scan it does not need yet */
#define GRT_KEYS_DECL                                                                                      \
    uint64_t k0 = kKeyInvalid, k1 = kKeyInvalid, k2 = kKeyInvalid, k3 = kKeyInvalid, k4 = kKeyInvalid,     \
             k5 = kKeyInvalid, k6 = kKeyInvalid, k7 = kKeyInvalid, k8 = kKeyInvalid, k9 = kKeyInvalid,     \
             k10 = kKeyInvalid, k11 = kKeyInvalid;
#define GRT_KEYS_RESET k0 = k1 = k2 = k3 = k4 = k5 = k6 = k7 = k8 = k9 = k10 = k11 = kKeyInvalid;
#elif GRT_TILE_KS == 8
#define KLAST k7
#define KPRESS k5
#define KROOM k4
#define GRT_KEYS_DECL                                                                                      \
    uint64_t k0 = kKeyInvalid, k1 = kKeyInvalid, k2 = kKeyInvalid, k3 = kKeyInvalid, k4 = kKeyInvalid,     \
             k5 = kKeyInvalid, k6 = kKeyInvalid, k7 = kKeyInvalid;
#define GRT_KEYS_RESET k0 = k1 = k2 = k3 = k4 = k5 = k6 = k7 = kKeyInvalid;
#else
#error "GRT_TILE_KS must be 8 or 12"
#endif
#include "grt_slots_gen.inc"
#define PL_OTHER(cell) pl_other[(cell) * kWG + lane]
#define PL_ALPHA(cell) pl_alpha[(cell) * kWG + lane]
#define PL_COL(cell, ch) pl_col[((ch) * KS + (cell)) * kWG + lane]

// Diagnostic build (make EXTRA=-DGRT_TILE_DIAG, never shipped; counters on): the counters hold WAVE-level trip counts —
// rays: node steps, segments: particles fetched, hit_evals: compositing steps, rounds: passes, node_visits: depth-first
// pops + window refills, proxy_tests: exact tests executed, rec_fetches: leaf steps, stall_exits: frontier rebalances.
// Checking build (make EXTRA=-DGRT_TILE_CHECK, never shipped): stall_exits counts violated invariants (an event turning
// up below the front: +1 per lane; frontier entries not conserved by a rebalance: +1000 per lane) and, when a float
// frame is rendered, row 0 of it receives the (t, 2 id + exit, T) log of the events lane GRT_TILE_CHECK_LANE composites.
#ifndef GRT_TILE_CHECK_LANE
#define GRT_TILE_CHECK_LANE 0u
#endif
#if defined(GRT_TILE_DIAG)
#define GRT_D(f, n) if (COUNT) w.f += (n);
#elif defined(GRT_MARKS)
#define GRT_D(f, n) asm volatile("; GRT_MARK " #f);
#else
#define GRT_D(f, n)
#endif
// The frustum fit uses the hardware reciprocal / reciprocal square root (v_rcp_f32 / v_rsq_f32, 1 ulp) instead of correctly rounded
// divisions and square roots (a tile re-fits its frustum every time half of its wanting lanes have finished): the per-lane
// 1 / (d . axis) — (tu, tv) move by 1.2e-7 relative, the bounds are widened by 1e-4; the plane normals — unit to 1.2e-7, against the
// 2e-5 slack of the plane tests; the per-axis slab bounds — (1 - 1e-6) / max|d| within 2.4e-7, still a lower bound.  Culling only:
// frames are bit-identical (C3 -1.3 %, round 4).  This is the build that failed in round 3 — a register-allocator defect, not
// numerics: profiles/r04_experiments_log.md 1, csrc/hipcc_via_asm.py.
#define GRT_FIT_RCP1(x) __builtin_amdgcn_rcpf(x)
#define GRT_FIT_NRM(p, x) ((p) * __builtin_amdgcn_rsqf(x))
#define GRT_FIT_DIV4(a, x) ((a) * __builtin_amdgcn_rcpf(x))

// Signed-float wave reductions: eleven per frustum fit, and a tile re-fits its frustum every time half of its wanting lanes
// have finished.  As `fminf(v, __shfl_xor(v, off))` each was six dependent LDS round trips (ds_bpermute) and eighteen VALU
// operations with their NaN canonicalisation; here a float goes through an order-preserving integer key (sign bit
// flipped for v >= 0, all bits for v < 0), the DPP integer minimum of grt_wave.h (wave_min / wave_min4: no LDS, four
// reductions interleaved) and back.  The result is wave-uniform and the exact minimum / maximum as before.
// NaN: unlike fminf / fmaxf the integer key does not drop it (a NaN would win the reduction and void the frustum for the
// whole tile).  No NaN reaches these reductions: they are fed from the rays of lanes with `alive`, which implies
// have_ray, i.e. length(d) > 0.1 (the reference's loop guard, shaders/tracer.cu:59 — false for a NaN direction, which is
// how a bounce off a zero shading normal ends), with origins that are the eye or a finite mesh hit point
// (tests/test_gpu_parity.py::test_mesh_with_zero_normals_nan_bounce_directions).
__device__ __forceinline__ uint32_t fkey(float f)
{
    const uint32_t b = __float_as_uint(f);
    return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k)
{
    return __uint_as_float(k ^ ((uint32_t)((int32_t)~k >> 31) | 0x80000000u));
}
__device__ __forceinline__ float wave_fmin(float v) { return fkey_inv(__float_as_uint(wave_min(__uint_as_float(fkey(v))))); }
__device__ __forceinline__ float wave_fmax(float v) { return fkey_inv(~__float_as_uint(wave_min(__uint_as_float(~fkey(v))))); }
// (min a, max b, min c, max d) in one go
__device__ __forceinline__ void wave_fminmax4(float a, float b, float c, float d, float& mna, float& mxb, float& mnc, float& mxd)
{
    float ra, rb, rc, rd;
    wave_min4(__uint_as_float(fkey(a)), __uint_as_float(~fkey(b)), __uint_as_float(fkey(c)), __uint_as_float(~fkey(d)), ra, rb, rc, rd);
    mna = fkey_inv(__float_as_uint(ra)); mxb = fkey_inv(~__float_as_uint(rb));
    mnc = fkey_inv(__float_as_uint(rc)); mxd = fkey_inv(~__float_as_uint(rd));
}
__device__ __forceinline__ float uni(float v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v)));
#else
    return v;
#endif
}
__device__ __forceinline__ uint32_t lanes_below(uint64_t m) // number of set bits of m below this lane
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
#else
    return 0u;
#endif
}
// wave64 minimum of 64-bit keys (two unsigned 32-bit DPP reductions: wave_min works on the bit patterns)
__device__ __forceinline__ uint64_t wave_umin64(uint64_t k)
{
    const uint32_t hi = (uint32_t)(k >> 32);
    const uint32_t mh = __float_as_uint(wave_min(__uint_as_float(hi)));
    const uint32_t lo = (hi == mh) ? (uint32_t)k : 0xFFFFFFFFu;
    const uint32_t ml = __float_as_uint(wave_min(__uint_as_float(lo)));
    return ((uint64_t)mh << 32) | (uint64_t)ml;
}
// minimum of 64-bit keys over the four lanes of a quad (lanes 4 q .. 4 q + 3), in every lane of the quad: the high words by two
// DPP minima, then the low words of the lanes that hold that high word
__device__ __forceinline__ uint64_t quad_umin64(uint64_t k)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t hi = (uint32_t)(k >> 32);
    uint32_t mh = min(hi, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0xB1, 0xF, 0xF, true)); // quad_perm [1,0,3,2]
    mh = min(mh, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mh, 0x4E, 0xF, 0xF, true));           // quad_perm [2,3,0,1]
    const uint32_t lo = (hi == mh) ? (uint32_t)k : 0xFFFFFFFFu;
    uint32_t ml = min(lo, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0xB1, 0xF, 0xF, true));
    ml = min(ml, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ml, 0x4E, 0xF, 0xF, true));
    return ((uint64_t)mh << 32) | (uint64_t)ml;
#else
    return k;
#endif
}
__device__ __forceinline__ float lane_value(float v, int l) // v of lane l (l wave-uniform)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), l));
#else
    return v;
#endif
}
__device__ __forceinline__ void wave_fence()
{
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // compiler ordering of the LDS exchange; no instruction
#endif
}

// A full overflow bag keeps its nearer half.  Without this the key that no longer fits is simply dropped, and that key is
// the window's own last one: the lane's cut-off then sits a dozen events ahead however much the bag holds, and a ray inside
// hundreds of overlapping proxies (all their exit events pending at once) needs a pass per dozen events.  Pruning instead
// keeps the cut-off at about the bag's median, ~50 events ahead, for one scan of the bag (+ five of a 16-entry sample).
// Lanes with `doit` prune; bp = the lane's column of its tile's chunk (entry i at bp[i * 64]).  Called between steps
// (few values live there), as soon as a bag has fewer than kPruneRoom free entries.
__device__ __forceinline__ void bag_prune(float4* bp, bool doit, uint32_t& nb, uint64_t& bagmin, uint64_t& lost)
{
    const uint32_t n = doit ? nb : 0u;
    uint32_t nmax = n;
    for (int off = 32; off > 0; off >>= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, off));
    nmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmax);
    // the threshold comes from a SAMPLE: the first 16 entries (the bag is in arrival order, which does not know the keys)
    const uint32_t ns = min(n, 16u), nsmax = min(nmax, 16u);
    uint32_t lo = 0xFFFFFFFFu, hi = 0u; // bit patterns of the keys' t (positive floats order like integers)
    for (uint32_t i = 0; i < nsmax; i++) {
        if (i < ns) {
            const uint32_t t = __float_as_uint(bp[(size_t)i * 64u].y);
            lo = min(lo, t);
            hi = max(hi, t);
        }
    }
    for (int it = 0; it < 4; it++) { // the largest threshold (to 1/16 of the range) that keeps at most half of the sample
        const uint32_t mid = lo + ((hi - lo) >> 1);
        uint32_t cnt = 0;
        for (uint32_t i = 0; i < nsmax; i++)
            if (i < ns) cnt += (__float_as_uint(bp[(size_t)i * 64u].y) <= mid) ? 1u : 0u;
        const bool few = cnt * 2u <= ns;
        lo = few ? mid : lo;
        hi = few ? hi : mid;
    }
    // entries with t <= lo stay
    uint32_t w = 0;
    uint64_t newmin = kKeyInvalid, dropmin = kKeyInvalid;
    for (uint32_t i = 0; i < nmax; i++) {
        if (i < n) {
            const float4 e = bp[(size_t)i * 64u];
            const uint64_t key = ((uint64_t)__float_as_uint(e.y) << 32) | (uint64_t)__float_as_uint(e.x);
            if (__float_as_uint(e.y) <= lo) {
                bp[(size_t)w * 64u] = e;
                newmin = (key < newmin) ? key : newmin;
                w++;
            } else {
                dropmin = (key < dropmin) ? key : dropmin;
            }
        }
    }
    // (equal distances, or a sample that misled: the bag must shrink whatever happens — the last quarter goes)
    const bool trunc = doit && (w * 8u > n * 7u);
    if (wave_any(trunc)) {
        const uint32_t w2 = trunc ? (w - (w >> 2)) : w;
        for (uint32_t i = 0; i < nmax; i++) {
            if (trunc && i >= w2 && i < w) {
                const float4 e = bp[(size_t)i * 64u];
                const uint64_t key = ((uint64_t)__float_as_uint(e.y) << 32) | (uint64_t)__float_as_uint(e.x);
                dropmin = (key < dropmin) ? key : dropmin;
            }
        }
        w = w2; // (newmin may now name a dropped entry: a smaller bagmin only asks for a refill early)
    }
    if (doit) {
        nb = w;
        bagmin = newmin;
        lost = (dropmin < lost) ? dropmin : lost;
    }
}

#ifdef GRT_TILE_CHECK
#define GRT_TILE_CHECK_FRONT(INS, K) if ((INS) && key_t(K) < F) c.stall_exits++; /* finality violated: an event below the front turned up late */
#else
#define GRT_TILE_CHECK_FRONT(INS, K)
#endif
// BUNDLE = true (stage 3 of the mesh wavefront pipeline): the wave's 64 rays are one chunk of the continuation queue —
// the rays of one 8x8 tile after their bounce, each with its own origin.  The frustum planes get offsets (each plane is
// pushed out to the outermost origin), the distance bounds are taken about the first ray's origin and loosened by the
// origins' spread, and a pass only takes the rays within ~16 degrees of its first ray (the others wait for a later pass),
// so that the bundle's frustum stays a useful cull.  The per-eye records do not apply: A (o - mu) is formed per lane.
// A bundle that is not one — rays spread wide AND through dense parts of the scene — makes every step pay for 64 rays that
// share nothing.  MODE 1 therefore works to a budget of steps: a chunk that exceeds it gives up, nothing of it is kept, and
// its rays go on the `heavy` list; MODE 2 traces the rays of that list ONE PER WAVE, a resident grid drawing from the list.
// There every lane holds the same ray, and the exact work turns round as well: LANES = PARTICLES.  The (<= 64) particles
// of a leaf step are slab-tested at once, each by the lane that culled its box (record by vector loads); a hit goes into
// THAT lane's window together with its colour, so the 64 windows are one pool of 768 pending events; compositing takes
// the smallest key of the pool (one 64-bit wave minimum per event) and updates the wave-uniform T / radiance.  A ray with
// a thousand events is then a few dozen steps, not a thousand — it is these rays that bound a per-lane or per-bundle
// kernel's run time.  Such a wave keeps its ray to the END: after a segment it traces the mesh itself (all lanes walk the
// small mesh tree in step on the idle overflow stack) and goes on with the next iteration of the bounce loop.  The same
// mode finishes the rays of the retry queue (a.single_own_mesh: per-lane segments that went over their budget).
// Same arithmetic per event, same order: same bits.
// PIECES = true: the tree holds pieces of split proxies (grt_api.hip: k_piece_boxes) — a piece reports a particle only when
// the lane's first pending event lies in its cell, and the repeats that are still possible are dropped.  Scenes without
// pieces run the PIECES = false instantiation, whose code is what it was before pieces existed (the few extra
// instructions cost the default scene 1.3 %, and any change to this kernel's hot loop is a lottery: see the watchdog).
// stage 1 of a mesh frame for the wave's 64 camera rays (k_primary_mesh_wave's work, record layout and all), called from the
// primary stage of the tile kernel when the two are fused (RenderArgs::mesh_primary_wave == 2)
template <bool COUNT>
__device__ __forceinline__ void mesh_primary_fused(const RenderArgs& a, uint32_t* stk, bool have_ray, f3 o, f3 d, float4& r0, float4& r1,
                                                             float4& r2, uint32_t& nv)
{
    const MeshHit mh = mesh_closest_wave<COUNT>(a, stk, have_ray, o, d, kTraceMeshTmin, kTraceMeshTmax, nv);
    int st_ = MeshPass;
    uint32_t nb_ = 0;
    float seg_tmax = a.p.t_max;
    f3 nrm_ = mk3(0, 0, 0), nextO = mk3(0, 0, 0), nextD = mk3(0, 0, 0);
    if (have_ray) mesh_shade(a, mh, o, d, st_, seg_tmax, nrm_, nextO, nextD, nb_);
    const uint32_t flags = (uint32_t)st_ | (nb_ << 8) | ((have_ray ? 1u : 0u) << 16);
    r0 = make_float4(seg_tmax, __uint_as_float(flags), nextO.x, nextO.y);
    r1 = make_float4(nextO.z, nextD.x, nextD.y, nextD.z);
    r2 = make_float4(nrm_.x, nrm_.y, nrm_.z, 0.0f);
}

template <bool COUNT, bool SH, bool MESH, int MODE, bool PIECES>
__global__ __launch_bounds__(kWG, MODE == 2 ? GRT_TILE_WAVES2 : (MODE == 3 ? kWavesQuad : kWavesPerSimd)) void k_render_tile(const RenderArgs a)
{
    constexpr bool BUNDLE = MODE == 1 || MODE == 2, SINGLE = MODE == 2;
    // MODE 3 (QUAD): camera rays of ONE 4x4 QUADRANT of a heavy tile, lanes = rays x slots: lane 4 r + s carries ray r (16 of them) and
    // is its slot s.  The exact work turns round as in MODE 2, but four-fold instead of sixty-four-fold: a trip of the exact-test loop
    // takes FOUR survivors of the leaf step, slot s of every ray tests survivor s (records by vector loads, A (o - mu) per lane) and the hits
    // go into THAT lane's window — a ray's pending events are the pool of its four windows (+ four bags), its next event the smallest
    // first key of the quad (two DPP minima).  T, radiance, last key, cut-off and `alive` are per RAY and are kept alike in the four
    // lanes.  Same arithmetic per event, same order: same bits.  What it is for: a tile whose wave bounds the frame (a rank's share of
    // a frame, a 256^2 frame) ran as four waves of 16 rays with 48 of 64 lanes idle — each a quarter of the rays but 0.78 of the time,
    // because the stream of exact tests and inserts is as long for 16 rays as for 64.  Here that stream is a quarter as long.
    constexpr bool QUAD = MODE == 3;
    const uint32_t rank = SINGLE ? blockIdx.x : (QUAD ? blockIdx.x : xcd_swizzle(blockIdx.x, gridDim.x, a.swizzle_chunk * 4u));
    const uint32_t n_in = BUNDLE ? (SINGLE ? *a.hcount : *a.qcount_in) : (QUAD ? a.qpart_count[0] : 0u); // chunks of the queue / rays of the heavy list / parts
    const uint32_t lane = threadIdx.x;
    __shared__ float pl_other[KS * kWG], pl_alpha[KS * kWG];
    __shared__ float pl_col[SINGLE ? 3 * KS * kWG : 1]; // MODE 2: the event's radiance, fetched by the lane that inserted it
    (void)pl_col;
    // The frustum's twelve plane components and three slab factors live in LDS (64 B: what was left under the 16-waves-per-CU
    // limit) instead of 21 SGPRs that are alive across every loop of the kernel; the step reads them back by four broadcast
    // ds_read_b128.  The kernel spilled 22 SGPRs to VGPR lanes; it spills 6 now, and the v_readlane / v_writelane inside loops — whose
    // static count predicts the frame across builds (profiles/r04_experiments_log.md 10, 11) — went from 171 to 141: C3 -4.5 %, C5 -4.4 %,
    // C2 -3.7 %, C3a -5.1 %, C4 -2.5 % (round 4).
    __shared__ __attribute__((aligned(16))) float fr_lds[16];
    __shared__ uint2 xch[kWG];       // children on their way to free frontier slots
    __shared__ uint32_t xsel[kBatch]; // refs of the nodes picked for this step
    __shared__ uint2 bag[kBag];      // far part of the frontier: (lambda bits, ref), unordered; its minimum is Fbag
    __shared__ float4 qstg[QUAD ? 4 * kWG : 1]; // QUAD: the records of a leaf step's survivors (slot = the lane that culled the box), 4 KB
    (void)qstg;
    __shared__ uint32_t dstack[kStack]; // depth-first overflow: the batch that overflowed (<= 64) + kTileWide - 1 siblings
                                 // per wide level below it (tile_stack_fits, grt_internal.h)
    bool first_draw = true;
    for (uint32_t unit_s = rank;;) { // (one trip; MODE 2: the waves draw the rays of the heavy list from a counter, so
                                     //  that a wave stuck with a long ray does not hold a share of the others back)
    if (SINGLE && !first_draw) { // (a wave's FIRST ray is the one of its own number: 2816 atomics on one address at the start of every
                                 //  launch took 30 us — an empty list's launch 34 us, and a mesh frame has three of those; the counter hands
                                 //  out the rays behind the grid's)
        uint32_t u_ = 0;
        if (lane == 0u) u_ = atomicAdd(a.hnext, 1u);
        unit_s = gridDim.x + (uint32_t)__builtin_amdgcn_readfirstlane((int)u_);
    }
    first_draw = false;
    // (QUAD: wave i takes entry i of the list of four-way parts that k_quad_list compacted from the launch order, heaviest first; the
    //  grid is the list's capacity — no draw loop: a loop around the whole kernel keeps 75 scalar registers alive across it)
    if ((BUNDLE || QUAD) && unit_s >= n_in) break; // wave-uniform
    // (a chunk whose tile gave up as a bundle in an earlier frame: its rays are on the early list of the one-ray-per-wave kernel already)
    if (MODE == 1 && a.qskip && a.qskip[unit_s]) break;
    Cnt c, w;
    (void)w;
    // camera rays: an entry of the launch order may name a PART of a heavy tile (grt_internal.h: kOrderUnitMask; grt_bvh.hip:
    // k_cost_order_parts) — the wave then traces the tile's upper / lower 4 rows, or one of its 4x4 quadrants, and the other
    // lanes carry no ray; entries past the last one are padding
    const uint32_t ue = QUAD ? a.qparts[unit_s] : ((a.order && !BUNDLE) ? a.order[rank] : unit_s);
    if (!BUNDLE && (a.order || QUAD) && (ue & kOrderUnitMask) >= a.n_units) break; // padding (kOrderPad), or anything that is not a tile of this launch
    // code 3 = a four-way part that k_quad_list handed to the quad kernel (MODE 3): not this kernel's when that kernel is launched
    // beside it (a.quad_parts); else a four-way part like any other
    if (MODE == 0 && a.quad_parts && (ue >> 30) == 3u) break;
    const uint32_t unit = BUNDLE ? ue : (ue & kOrderUnitMask);
    // (lane = 8 row + column: bit 5 = lower half of the tile, bit 2 = right half; formed from `ue` where it is needed — at the
    //  ray set-up and at the pixel write — so that nothing but `ue` lives across the passes)
#define GRT_IN_PART (BUNDLE || QUAD || (ue >> 30) == 0u || /* (codes 1, 2: halves, quarters as part waves of this kernel) */ (((ue >> 30) == 1u ? (lane >> 5) : (((lane >> 5) << 1) | ((lane >> 2) & 1u))) == ((ue >> 28) & 3u)))
    // the heaviest tiles of the previous frame (the head of the cost-sorted order) bound the frame: they issue first
    if (!BUNDLE && a.order && a.tile_prio_div && rank < gridDim.x / a.tile_prio_div) __builtin_amdgcn_s_setprio(2);
    if (QUAD) __builtin_amdgcn_s_setprio(3); // (the quad kernel's waves ARE the frame's critical path: they issue first on their SIMD)
    const uint32_t blk = unit >> 2, wave = unit & 3u;
    // (QUAD: ray r = lane / 4 is pixel (r % 4, r / 4) of quadrant `part`: bit 0 = right half, bit 1 = lower half of the tile)
    const uint32_t tx8 = QUAD ? (((ue >> 28) & 1u) * 4u + ((lane >> 2) & 3u)) : (lane & 7u), ty8 = QUAD ? (((ue >> 29) & 1u) * 4u + (lane >> 4)) : (lane >> 3);
    const uint32_t lx = (wave & 1u) * 8u + tx8, ly = (wave >> 1) * 8u + ty8;
    uint32_t px = 0, py = 0;
    size_t out_idx = 0;
    bool in_frame = false;
    bool aborted = false; // MODE 1: over the step budget
    // BUNDLE: this lane's entry of the incoming queue (MODE 2: the wave's ONE ray, on lane 0)
    const size_t ent = SINGLE ? (a.heavy ? (size_t)a.heavy[unit] : (size_t)unit) : (size_t)unit * 64u + lane;
    const size_t qi = ent * 4;
    if (BUNDLE) {
        const float4 q3 = a.queue_in[qi + 3];
        in_frame = (__float_as_uint(q3.y) >> 31) != 0u; // lanes that carry a ray (MODE 2: all 64 hold the SAME ray)
        out_idx = (size_t)__float_as_uint(q3.z) | ((size_t)__float_as_uint(q3.w) << 32);
    } else if (a.mode == 0) {
        px = a.x0 + (blk % a.nbx) * 16u + lx;
        py = a.y0 + (blk / a.nbx) * 16u + ly;
        in_frame = (px < a.x1) && (py < a.y1);
        out_idx = (size_t)py * a.p.width + px;
    } else {
        const uint32_t per_tile = a.nbx * a.nby;
        const uint32_t j = blk / per_tile, sub = blk % per_tile;
        const uint32_t tile = a.first_tile + j * a.tile_stride;
        const uint32_t tx_ = tile % a.tiles_x, ty_ = tile / a.tiles_x;
        const uint32_t ox = (sub % a.nbx) * 16u + lx, oy = (sub / a.nbx) * 16u + ly;
        px = tx_ * a.tile_w + ox;
        py = ty_ * a.tile_h + oy;
        in_frame = (px < a.p.width) && (py < a.p.height);
        out_idx = ((size_t)j * a.tile_h + oy) * a.tile_w + ox;
    }
    const bool tally = SINGLE ? (lane == 0u) : (!QUAD || (lane & 3u) == 0u); // per-ray counters (and the pixel): once per ray
    const bool write = BUNDLE ? (in_frame && tally) : ((in_frame || (a.mode == 1)) && tally);
    const f3 nU = mk3(-a.p.U[0], -a.p.U[1], -a.p.U[2]), nV = mk3(-a.p.V[0], -a.p.V[1], -a.p.V[2]);
    const f3 W = mk3(a.p.W[0], a.p.W[1], a.p.W[2]);
    f3 o = mk3(a.p.eye[0], a.p.eye[1], a.p.eye[2]); // wave-uniform origin (camera rays); per lane when BUNDLE
    f3 d = mk3(0.0f, 0.0f, -1.0f);
    bool have_ray = in_frame;
    float density_in = 0.0f;
    if (BUNDLE) {
        if (in_frame) {
            const float4 q0 = a.queue_in[qi], q1 = a.queue_in[qi + 1], q2 = a.queue_in[qi + 2];
            o = mk3(q0.x, q0.y, q0.z);
            d = mk3(q0.w, q1.x, q1.y);
            density_in = q2.w;
        }
    } else if (in_frame) {
        if (!a.p.mode_fisheye) get_ray(px, py, nU, nV, W, a.p.width, a.p.height, d);
        else have_ray = get_fisheye_ray(px, py, nU, nV, W, a.p.width, a.p.height, d);
    }
    have_ray = have_ray && GRT_IN_PART;
    // (mesh frames: stage 1 counts the rays — unless this kernel IS stage 1: a.mesh_primary_wave == 2, below)
    if (COUNT && have_ray && tally && (!MESH || (MODE == 0 && a.mesh_primary_wave == 2u))) c.rays++;
    have_ray = have_ray && (length3(d) > 0.1f) && (a.p.max_bounces > 0u); // loop guard, shaders/tracer.cu:59
    float seg_tmax = a.p.t_max;
    uint32_t pflags = 0;
    f3 nextO = mk3(0, 0, 0), nextD = mk3(0, 0, 0), hitN = mk3(0, 0, 0);
    if (SINGLE && a.single_own_mesh) { // a ray from the retry queue of k_bounce: no mesh-hit record yet
        if (have_ray) {
            uint32_t it_ = 0, nv_ = 0;
            const MeshHit mh = mesh_closest_t<COUNT, 1>(a, dstack, o, d, kTraceMeshTmin, kTraceMeshTmax, it_, nv_);
            if (COUNT && tally) c.node_visits += nv_;
            int st_ = MeshPass;
            uint32_t nb_ = __float_as_uint(a.queue_in[qi + 3].x);
            f3 nrm_;
            mesh_shade(a, mh, o, d, st_, seg_tmax, nrm_, nextO, nextD, nb_);
            hitN = nrm_;
            pflags = (uint32_t)st_ | (nb_ << 8) | (1u << 16);
        }
    } else if (MESH && MODE == 0 && a.mesh_primary_wave == 2u) {
        // stage 1 FUSED (round 6): the tile's 64 camera rays walk the mesh tree together right here (grt_mesh.h: mesh_closest_wave, on
        // the depth-first stack, idle before the first pass) — no launch of its own in front of the Gaussian stage (0.16 ms of a 2.3 ms
        // frame, its long waves those of the sphere's limb), no 48-B record per pixel written and read back.  The same MeshHit and
        // closest-hit shading as k_primary_mesh, bit for bit.
        // (inlined.  As a function of its own — __attribute__((noinline)), tried — the CALL costs every launch of this kernel its scratch
        //  set-up and C4 went from 2.29 to 3.0 ms whichever stage-1 form ran; inlined it costs this instantiation 63 more spilled VGPRs
        //  outside the loops and 21 more lane moves inside them, and the frame still gains: 2.29 -> 2.24 ms)
        float4 r0_, r1_, r2_;
        uint32_t nv_ = 0;
        mesh_primary_fused<COUNT>(a, dstack, have_ray, o, d, r0_, r1_, r2_, nv_);
        if (COUNT) c.node_visits += nv_;
        seg_tmax = r0_.x;
        pflags = __float_as_uint(r0_.y);
        nextO = mk3(r0_.z, r0_.w, r1_.x);
        nextD = mk3(r1_.y, r1_.z, r1_.w);
        hitN = mk3(r2_.x, r2_.y, r2_.z);
    } else if (MESH) { // stage 1 (k_primary_mesh / k_queue_mesh) already traced the mesh for this ray
        const size_t pi = BUNDLE ? ent * 3 : ((size_t)blk * kBlock + wave * 64u + lane) * 3;
        const float4 pr0 = a.prec[pi], pr1 = a.prec[pi + 1], pr2 = a.prec[pi + 2];
        seg_tmax = pr0.x;
        pflags = __float_as_uint(pr0.y);
        nextO = mk3(pr0.z, pr0.w, pr1.x);
        nextD = mk3(pr1.y, pr1.z, pr1.w);
        hitN = mk3(pr2.x, pr2.y, pr2.z);
        have_ray = have_ray && ((pflags >> 16) & 1u);
    }

    f3 col = mk3(0.0f, 0.0f, 0.0f);
    bool cont = false; // MESH: the ray goes on bouncing (stage 3)
    f3 accumColor = mk3(0, 0, 0);
    float accumAlpha = 0.0f, blocking = 0.0f;
    uint32_t timeout = 0;
    if (BUNDLE && in_frame) { // the accumulators of the iterations before this one
        const float4 q1 = a.queue_in[qi + 1], q2 = a.queue_in[qi + 2], q3 = a.queue_in[qi + 3];
        accumColor = mk3(q1.z, q1.w, q2.x);
        accumAlpha = q2.y;
        blocking = q2.z;
        timeout = __float_as_uint(q3.y) & 0x7FFFFFFFu;
        col = accumColor;
    }
    bool gave_up = false; // MODE 1: the chunk went over its budget
    float density = 0.0f;
    // MODE 2 keeps its ray until it ends: every trip of this loop is one iteration of the reference's bounce loop
    // (shaders/tracer.cu:58-106); the other modes make one trip and queue the rays that go on
    for (;;) {
    // ---- trace() for the whole wave (shaders/tracer.cuh:328-373), density starts at 0 ----
    const float minT = a.p.minTransmittance;
    float T = 1.0f - density_in; // the payload's density carries over from segment to segment (shaders/tracer.cuh:331)
    f3 radiance = mk3(0.0f, 0.0f, 0.0f);
    if (COUNT && have_ray && tally) c.segments++;
    const uint64_t raym = wave_ballot(have_ray);
    if (a.root_ref != kNoRoot && raym) {
        const float epsT = 1e-9f;
        const f3 dn = normalize3(d);
        const float t_hi = seg_tmax + epsT; // per lane when MESH (segment ends at the mesh hit)
        const float t_hi_m = __uint_as_float(__float_as_uint(t_hi) - 1u); // largest float below t_hi (t_hi > 0)

        // ---- the tile's frustum (wave-uniform; culling only) ----
        // axis = direction of the first lane that has a ray; (u, v) complete it; a lane's direction is
        // d ~ ax + tu u + tv v, and the four planes bound (tu, tv) over the lanes, widened by 1e-4 rad.
        f3 ax, uu, vv;
        f3 oc = o; // the point the boxes are measured from: the eye; BUNDLE: the origin of the pass's first ray
#define GRT_AXES(MASK)                                                                                     \
        {                                                                                                  \
            const int l0 = (int)__builtin_ctzll(MASK);                                                     \
            ax = mk3(__shfl(d.x, l0), __shfl(d.y, l0), __shfl(d.z, l0));                                   \
            if (BUNDLE) oc = mk3(__shfl(o.x, l0), __shfl(o.y, l0), __shfl(o.z, l0));                       \
            const float axx = fabsf(ax.x), ayy = fabsf(ax.y), azz = fabsf(ax.z);                           \
            const f3 e_ = (axx <= ayy && axx <= azz) ? mk3(1, 0, 0) : ((ayy <= azz) ? mk3(0, 1, 0) : mk3(0, 0, 1)); \
            uu = normalize3(cross3(ax, e_));                                                               \
            vv = cross3(ax, uu);                                                                           \
        }
        GRT_AXES(raym)
        // The frustum bounds the lanes that still WANT something (GRT_FRUSTUM(mask)): all rays at first; re-fitted when
        // half of them have finished (saturated, or past their window cut-off), so that a few straggling rays do not
        // drag the whole tile's frustum through the rest of the scene.
        float pLx, pLy, pLz, pRx, pRy, pRz, pBx, pBy, pBz, pTx, pTy, pTz;
        float ivx, ivy, ivz; // per-axis slab bound: when every ray moves the same way along an axis,
                             // t >= (near plane - eye) / (largest |d|); 0 when the directions straddle the axis
        bool shx, shy, shz;  // near plane is the box's hi side
        // BUNDLE only (all zero / one for camera rays): plane offsets mP = min over the rays of n_P . (o - oc) (<= 0: a ray
        // stays on the inner side of the plane through ITS origin), per-axis origin offsets, the origins' spread about oc
        // and the bounds of |d| (a bounced direction is a unit vector only up to rounding)
        float mL = 0.0f, mR = 0.0f, mB = 0.0f, mT = 0.0f, ofx = 0.0f, ofy = 0.0f, ofz = 0.0f, rmax = 0.0f, idmax = 1.0f, idmin = 1.0f;
#define GRT_AXIS(M, C, IV, SH_, OF, MN, MX)                                                                \
        {                                                                                                  \
            const float mn_ = uni(MN), mx_ = uni(MX);                                                      \
            SH_ = mx_ < -1e-20f;                                                                           \
            IV = (mn_ > 1e-20f) ? GRT_FIT_DIV4(1.0f - 1e-6f, mx_) : (SH_ ? GRT_FIT_DIV4(1.0f - 1e-6f, mn_) : 0.0f); \
            IV = uni(pk_ * IV);                                                                            \
            if (BUNDLE) { /* the origin nearest to the box side the rays enter through */                  \
                const float dl_ = o.C - oc.C;                                                              \
                const float q_ = SH_ ? uni(wave_fmin((M) ? dl_ : INFINITY)) : uni(wave_fmax((M) ? dl_ : -INFINITY)); \
                OF = SH_ ? (q_ - 2e-6f * fabsf(q_) - 1e-30f) : (q_ + 2e-6f * fabsf(q_) + 1e-30f);          \
            }                                                                                              \
        }
#define GRT_POFF(M, P, MP)                                                                                 \
        {                                                                                                  \
            const float dx_ = o.x - oc.x, dy_ = o.y - oc.y, dz_ = o.z - oc.z;                              \
            const float s_ = P##x * dx_ + P##y * dy_ + P##z * dz_ - 8e-6f * ((fabsf(dx_) + fabsf(dy_)) + fabsf(dz_)); \
            MP = uni(wave_fmin((M) ? s_ : INFINITY));                                                      \
        }
#define GRT_PNORM(P, X, Y, Z)                                                                              \
        {                                                                                                  \
            const float x_ = (X), y_ = (Y), z_ = (Z);                                                      \
            const float il_ = GRT_FIT_NRM(pk_, __builtin_fmaf(x_, x_, __builtin_fmaf(y_, y_, z_ * z_))); \
            P##x = uni(x_ * il_); P##y = uni(y_ * il_); P##z = uni(z_ * il_);                              \
        }
#define GRT_PK_OF_PASS uni(fr_lds[15])
#define GRT_PLANES_TO_LDS_(REFIT)                                                                          \
            if (lane == 0u) {                                                                              \
                float4* q_ = (float4*)fr_lds;                                                              \
                q_[0] = make_float4(pLx, pLy, pLz, pRx); q_[1] = make_float4(pRy, pRz, pBx, pBy);          \
                q_[2] = make_float4(pBz, pTx, pTy, pTz);                                                   \
                if (!(REFIT)) q_[3] = make_float4(ivx, ivy, ivz, pk_); /* .w: the pass's width check, for its re-fits */ \
            }                                                                                              \
            wave_fence();
// REFIT (camera rays): a re-fit inside a pass narrows the four planes only.  The per-axis slab factors and
// the width check of the pass's first fit bound a superset of the lanes that are left, so they stay valid (culling only, and the
// largest |d| over an 8x8 tile moves in its fourth digit), and eight of a fit's twelve wave reductions are not run again.
#define GRT_FRUSTUM(M) GRT_FRUSTUM_(M, false)
#define GRT_FRUSTUM_(M, REFIT)                                                                             \
        {                                                                                                  \
            const float da = dot3(d, ax);                                                                  \
            const float ida = GRT_FIT_RCP1(fmaxf(da, 1e-6f));                                              \
            const float tu = dot3(d, uu) * ida, tv = dot3(d, vv) * ida;                                    \
            /* a tile wider than ~75 degrees (tiny fisheye frames) gets no culling at all: every box passes */ \
            float mnx_ = 0.0f, mxx_ = 0.0f, mny_ = 0.0f, mxy_ = 0.0f, mnz_ = 0.0f, mxz_ = 0.0f, damin_ = 1.0f, spare_; \
            if (!(REFIT)) {                                                                                \
            wave_fminmax4((M) ? d.x : INFINITY, (M) ? d.x : -INFINITY, (M) ? d.y : INFINITY, (M) ? d.y : -INFINITY, mnx_, mxx_, mny_, mxy_); \
            wave_fminmax4((M) ? d.z : INFINITY, (M) ? d.z : -INFINITY, (M) ? da : 1.0f, -INFINITY, mnz_, mxz_, damin_, spare_); \
            (void)spare_;                                                                                  \
            }                                                                                              \
            const float pk_ = (REFIT) ? GRT_PK_OF_PASS : ((uni(damin_) >= 0.25f) ? 1.0f : 0.0f);            \
            float tu0, tu1, tv0, tv1;                                                                      \
            wave_fminmax4((M) ? tu : INFINITY, (M) ? tu : -INFINITY, (M) ? tv : INFINITY, (M) ? tv : -INFINITY, tu0, tu1, tv0, tv1); \
            tu0 = uni(tu0); tu1 = uni(tu1); tv0 = uni(tv0); tv1 = uni(tv1);                                \
            tu0 -= 1e-4f * (1.0f + fabsf(tu0)); tu1 += 1e-4f * (1.0f + fabsf(tu1));                        \
            tv0 -= 1e-4f * (1.0f + fabsf(tv0)); tv1 += 1e-4f * (1.0f + fabsf(tv1));                        \
            /* unit normals (the leaf step sets a bounding sphere's radius against them); pk_ = 0: no plane at all */ \
            GRT_PNORM(pL, uu.x - tu0 * ax.x, uu.y - tu0 * ax.y, uu.z - tu0 * ax.z)                             \
            GRT_PNORM(pR, tu1 * ax.x - uu.x, tu1 * ax.y - uu.y, tu1 * ax.z - uu.z)                             \
            GRT_PNORM(pB, vv.x - tv0 * ax.x, vv.y - tv0 * ax.y, vv.z - tv0 * ax.z)                             \
            GRT_PNORM(pT, tv1 * ax.x - vv.x, tv1 * ax.y - vv.y, tv1 * ax.z - vv.z)                             \
            if (!(REFIT)) {                                                                                \
            GRT_AXIS(M, x, ivx, shx, ofx, mnx_, mxx_)                                                      \
            GRT_AXIS(M, y, ivy, shy, ofy, mny_, mxy_)                                                      \
            GRT_AXIS(M, z, ivz, shz, ofz, mnz_, mxz_)                                                      \
            }                                                                                              \
            GRT_PLANES_TO_LDS_(REFIT)                                                                      \
            if (BUNDLE) {                                                                                  \
                GRT_POFF(M, pL, mL) GRT_POFF(M, pR, mR) GRT_POFF(M, pB, mB) GRT_POFF(M, pT, mT)            \
                const float r_ = length3(sub3(o, oc)), ld_ = length3(d);                                   \
                rmax = uni(wave_fmax((M) ? r_ : 0.0f)) * (1.0f + 4e-6f) + 1e-30f;                          \
                idmax = 1.0f / (uni(wave_fmax((M) ? ld_ : 0.0f)) * (1.0f + 4e-6f));                        \
                idmin = (1.0f + 4e-6f) / uni(wave_fmin((M) ? ld_ : INFINITY));                             \
            }                                                                                              \
        }

        uint64_t last_key = mk_skey(a.p.t_min + epsT, 0x03FFFFFFu, 1) | kCellMask; // last composited event (exclusive bound)
        bool alive = have_ray && (T > minT);
        uint32_t stalls = 0;
        GRT_KEYS_DECL
        uint32_t pmask = 0; // payload cells in use
        uint32_t iters = 0; // wave-uniform work measure for the scheduling feedback
        uint32_t work = 0;  // MODE 1: particles fetched + 2 x exact tests run (wave-uniform), against the budget
        bool watchdog = false;
#ifdef GRT_TILE_CHECK
        uint32_t dbg_n = 0, dbg_m = 0;
#endif
        uint32_t chunk = kNoRoot; // this tile's first chunk of the overflow pool (taken at the first window overflow)
        // (QUAD: a ray's four bags TOGETHER hold what one bag of the camera-ray kernel holds.  The capacity is a cut-off, not just room:
        //  a ray whose bag is full stops wanting what lies beyond, the tile's reach shrinks with it and the frontier stays clear of far
        //  entries — with four full-size bags a quadrant of a cluster core went on for 242 steps where the part wave took 164)
        // (a whole tile whose bags stayed shallow in the frame before — part field 1 or 2 of its order entry — starts in one or two chunks;
        //  part field 0: a tile without a cost word — a cold frame — starts in ONE (a.ovf_cls0; in three when the order's entries are bare
        //  unit numbers: mesh frames, part waves off).  A tile that outgrows its chunks moves: three fresh
        //  chunks, what its rays hold is copied over (rare: a wave-level copy of at most 64 entries per ray), and it goes on with a full bag.
        //  bag_cap: the capacity in use; bit 16: no more moves — the pool had nothing left.  QUAD: a ray's four bags hold a quarter each)
        const uint32_t fld_ = (MODE == 0 && a.order && (ue >> 30) == 0u) ? ((ue >> 28) & 3u) : 3u;
        uint32_t bag_cap = QUAD ? max(a.ovf_entries >> 2, 1u) : min(a.ovf_entries, (fld_ ? fld_ : a.ovf_cls0) * kSub);
#define GRT_BAG_CAP (bag_cap & 0xFFFFu)
#define GRT_PRUNE_ROOM (QUAD ? kPruneRoom / 4u : (GRT_BAG_CAP > 2u * kSub ? kPruneRoom : kSub / 4u))
        const uint32_t ready_min = SINGLE ? 1u : a.tile_ready_min; // lanes with a final event before a compositing sweep starts
        // a lone ray meets few boxes per level: it looks much further ahead, so that a step still has 64 boxes to cull
        const float look_ = SINGLE ? a.single_look : a.tile_look, band_ = SINGLE ? a.single_band : a.tile_band;

        uint32_t npass = 0;
        while (wave_any(alive)) { // one iteration = one front-to-back pass
            npass++;
            bool parked = false; // BUNDLE: alive, but outside this pass's cone of directions
            if (BUNDLE) {
                GRT_AXES(wave_ballot(alive))
                const bool in_cone = dot3(d, ax) >= 0.96f * length3(d) * length3(ax);
                parked = alive && !in_cone;
                alive = alive && in_cone;
            }
            if (COUNT && alive && tally) c.rounds++;
            GRT_D(rounds, 1)
            const uint64_t pass_lo = last_key; // events with key <= pass_lo were composited by an earlier pass
            const float t_lo = key_t(pass_lo);
            // Window overflow: the particle that no longer fits (the farthest of the 12 + 1) goes to the lane's BAG in
            // global memory (16 B: key, exit t, alpha) instead of being dropped; `cutoff` is the smallest key that is not
            // in the window (bag or lost) and compositing never passes it; when a lane's next event sits in its bag the
            // bag is scanned once and the 12 smallest keys of (window + bag) are back in the window.  Only a full bag (or
            // an exhausted pool) really loses an event (`lost`), which costs that lane another pass as before.
            uint64_t bagmin = kKeyInvalid; // smallest key in this lane's bag
            uint64_t lost = kKeyInvalid;   // smallest key this lane had to drop for good in this pass
            uint32_t nb = 0;               // entries in this lane's bag
            bool bags = false;             // some lane has a non-empty bag (wave-uniform)
            GRT_KEYS_RESET
            pmask = 0;
            // wave-level interval of interest: nothing beyond LIM, nothing that ends before LO (stale values are
            // conservative: LIM only shrinks, LO only grows)
            float LIM = uni(wave_fmax(alive ? t_hi_m : 0.0f));
            const float LO = uni(wave_fmin(alive ? t_lo : INFINITY));
            bool lim_dirty = false;
            GRT_FRUSTUM(alive)
            uint32_t nact_ref = (uint32_t)__popcll(wave_ballot(alive)); // wanting lanes the frustum was fitted to
            uint32_t nact_cur = nact_ref;                               // wanting lanes now (loop top)
            // frontier: slot i = lane i; free slot: (inf, kNoRoot)
            float fl = (lane == 0u) ? 0.0f : INFINITY;
            uint32_t fr = (lane == 0u) ? a.root_ref : kNoRoot;
            uint32_t dsp = 0;
            uint32_t nbag = 0;       // entries in the LDS bag
            float Fbag = INFINITY;   // smallest lambda in the bag
            bool rebal = false;      // children were parked in the bag: re-split near / far before going on
            float F = 0.0f;
            float Ff_cur = 0.0f; // minimum of the register part of the frontier (loop top)
            bool done = false;
            // (the rebalance of frontier + LDS bag, as a lambda: called where the step loop finds it due)
            auto do_rebalance = [&](const uint32_t nocc_, float& Ff) {
                            // ---- rebalance: the nearest kKeep entries of (frontier + bag) stay in registers, the rest
                            //      goes (back) to the bag.  Everything passes through registers: 4 bag entries per lane.
                            GRT_D(stall_exits, 1)
                            float bl0, bl1, bl2, bl3;
                            uint32_t br0, br1, br2, br3;
#define GRT_BLD(K)                                                                                         \
                            {                                                                              \
                                const uint32_t i_ = (K) * 64u + lane;                                      \
                                const uint2 e_ = (i_ < nbag) ? bag[i_] : make_uint2(0x7F800000u, kNoRoot);  \
                                bl##K = __uint_as_float(e_.x);                                             \
                                br##K = e_.y;                                                              \
                            }
                            GRT_BLD(0) GRT_BLD(1) GRT_BLD(2) GRT_BLD(3)
#undef GRT_BLD
                            wave_fence();
                            const float lo0 = wave_min(fminf(fminf(fl, bl0), fminf(bl1, fminf(bl2, bl3))));
                            float th = INFINITY;
                            if (nocc_ + nbag > kKeep) { // six bisection steps on the distance threshold
                                const float f0 = (fl < INFINITY) ? fl : 0.0f, f1 = (bl0 < INFINITY) ? bl0 : 0.0f,
                                            f2 = (bl1 < INFINITY) ? bl1 : 0.0f, f3 = (bl2 < INFINITY) ? bl2 : 0.0f,
                                            f4 = (bl3 < INFINITY) ? bl3 : 0.0f;
                                float lo_ = lo0, hi_ = uni(wave_fmax(fmaxf(fmaxf(f0, f1), fmaxf(f2, fmaxf(f3, f4)))));
                                // (up to kBisect steps, until at least half of kKeep qualify: in a dense cluster hundreds of
                                //  entries lie within 1e-3 of each other while the farthest one stretches the interval)
                                for (int it = 0; it < kBisect; it++) {
                                    const float mid = 0.5f * (lo_ + hi_);
                                    const uint32_t n_ = (uint32_t)__popcll(wave_ballot(fl <= mid)) + (uint32_t)__popcll(wave_ballot(bl0 <= mid)) +
                                                        (uint32_t)__popcll(wave_ballot(bl1 <= mid)) + (uint32_t)__popcll(wave_ballot(bl2 <= mid)) +
                                                        (uint32_t)__popcll(wave_ballot(bl3 <= mid));
                                    const bool few = n_ <= kKeep;
                                    lo_ = few ? mid : lo_;
                                    hi_ = few ? hi_ : mid;
                                    if (few && n_ * 2u >= kKeep && it >= 5) break;
                                }
                                th = lo_;
                            }
                            // near entries -> xch by rank (ties beyond 56 stay far), far entries -> bag by rank
                            uint32_t nk = 0, nfar = 0;
                            float far_min = INFINITY;
#define GRT_SPLIT(LAM, REF)                                                                                \
                            {                                                                              \
                                const bool v_ = (REF) != kNoRoot;                                          \
                                const bool near_ = v_ && ((LAM) <= th);                                    \
                                const uint64_t nm_ = wave_ballot(near_);                                   \
                                const uint32_t kr_ = nk + lanes_below(nm_);                                \
                                const bool keep_ = near_ && (kr_ < 56u);                                   \
                                const uint64_t km_ = wave_ballot(keep_);                                   \
                                const bool far_ = v_ && !keep_;                                            \
                                const uint64_t fm_ = wave_ballot(far_);                                    \
                                if (keep_) xch[kr_] = make_uint2(__float_as_uint(LAM), (REF));             \
                                if (far_) bag[nfar + lanes_below(fm_)] = make_uint2(__float_as_uint(LAM), (REF)); \
                                far_min = fminf(far_min, far_ ? (LAM) : INFINITY);                         \
                                nk += (uint32_t)__popcll(km_);                                             \
                                nfar += (uint32_t)__popcll(fm_);                                           \
                            }
                            GRT_SPLIT(fl, fr) GRT_SPLIT(bl0, br0) GRT_SPLIT(bl1, br1) GRT_SPLIT(bl2, br2) GRT_SPLIT(bl3, br3)
#undef GRT_SPLIT
                            wave_fence();
                            {
                                const uint2 v_ = (lane < nk) ? xch[lane] : make_uint2(0x7F800000u, kNoRoot);
                                fl = __uint_as_float(v_.x);
                                fr = v_.y;
                            }
                            wave_fence();
#ifdef GRT_TILE_CHECK
                            if (nk + nfar != nocc_ + nbag) {
                                c.stall_exits += 1000u; // conservation of the frontier entries
                                if (a.outf && lane == 0u) {
                                    float* q_ = a.outf + (size_t)a.p.width * 3 + dbg_m * 12; // row 1 of the frame
                                    q_[0] = (float)nk; q_[1] = (float)nfar; q_[2] = (float)nocc_; q_[3] = (float)nbag; q_[4] = th; q_[5] = lo0;
                                    q_[6] = (float)__popcll(wave_ballot(br0 != kNoRoot)); q_[7] = (float)__popcll(wave_ballot(br1 != kNoRoot));
                                    q_[8] = (float)__popcll(wave_ballot(br2 != kNoRoot)); q_[9] = (float)__popcll(wave_ballot(br3 != kNoRoot));
                                    q_[10] = (float)iters;
                                }
                                dbg_m++;
                            }
                            if ((uint32_t)__popcll(wave_ballot(fr != kNoRoot)) != nk) c.stall_exits += 100000u;
#endif
                            nbag = nfar;
                            Fbag = nfar ? wave_min(far_min) : INFINITY;
                            rebal = false;
                            Ff = wave_min(fl);
                            Ff_cur = Ff;
            };

            while (true) {
                uint32_t cur = kNoRoot; // entry taken off the overflow stack (depth-first mode; F stays as it is)
                if (dsp && !wave_any(alive)) dsp = 0; // every lane is done: nothing on the stack matters any more
                const bool dfs = dsp != 0u;
                if (dfs) {
                    --dsp;
                    cur = dstack[dsp];
                    cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur);
                    GRT_D(node_visits, 1)
                } else {
                    float Ff = wave_min(fl);
                    Ff_cur = Ff;
                    if (nbag) {
                        const uint32_t nocc_ = (uint32_t)__popcll(wave_ballot(fr != kNoRoot));
                        // (everything but one entry may end up in the bag: only when frontier + bag fit it)
                        if ((rebal || !(Ff < INFINITY) || ((Fbag <= Ff + Ff * look_) && (nocc_ + 8u <= kKeep))) &&
                            (nocc_ + nbag <= kBag)) {
                            do_rebalance(nocc_, Ff);
                        }
                    }
                    F = fminf(Ff, Fbag);
                    // lanes that still want something in this pass: alive and not yet past their window cut-off
                    const float ct_ = (lost != kKeyInvalid) ? key_t(lost) : t_hi_m;
                    const bool act = alive && (ct_ >= F);
                    const uint32_t nact = (uint32_t)__popcll(wave_ballot(act));
                    nact_cur = nact;
                    if (nact == 0u) F = INFINITY; // nothing left to find: the pass is over
                    done = !(F < INFINITY);
                    if (!done && (nact * 2u <= nact_ref)) { // half of them have finished: re-fit the frustum
                        GRT_FRUSTUM_(act, !BUNDLE)
                        LIM = uni(wave_fmax(act ? ct_ : 0.0f));
                        lim_dirty = false;
                        nact_ref = nact;
                    }
                }

                // ---- composite buffered events with t < F (and key < cutoff), in key order; deferred until
                //      ready_min lanes have one, a window is nearly full, or the pass is over ----
                // A bag is pruned only where an overflow would hurt: while the front still stands at the start of the pass
                // (the particles that CONTAIN the origin arrive in no order at all, by the hundred in a dense cluster), and in
                // any later pass (a lane came back: its cut-off did fall short).  Behind a moving front the arrivals are
                // ordered, what overflows lies far ahead, and the scans would be wasted (100 k-Gaussian frame: 10-35 % slower).
                if (!SINGLE && !dfs && bags && ((F <= LO) || npass > 1u)) {
                    const bool pr_ = alive && (chunk < a.ovf_chunks) && (nb + GRT_PRUNE_ROOM >= GRT_BAG_CAP);
                    if (wave_any(pr_)) { // wave-uniform, rare
                        work |= 3u; // (deep bags: below)
                        bag_prune(a.ovf_pool + (size_t)chunk * (kSub * 64u) + lane, pr_, nb, bagmin, lost);
                        if (QUAD) lost = quad_umin64(lost);
                        lim_dirty = true;
                    }
                }
                if (SINGLE && !dfs) {
                    // ---- MODE 2: the pool's smallest final key, one event per trip; T / radiance are wave-uniform ----
                    while (true) {
                        const bool cl_ = alive && (k0 != kKeyInvalid) && (key_t(k0) < F) && (k0 < lost);
                        if (!wave_any(cl_)) break;
                        GRT_D(hit_evals, 1)
                        const uint64_t ek = wave_umin64(cl_ ? k0 : kKeyInvalid);
                        const bool own = cl_ && (k0 == ek); // exactly one lane: a particle is tested once per pass
                        const uint64_t om = wave_ballot(own);
                        const int ol = (int)__builtin_ctzll(om);
                        const uint32_t cell = (uint32_t)(ek & kCellMask);
                        const uint32_t id = skey_id(ek);
                        float ea = 0.0f, eo = INFINITY, cr = 0.0f, cg = 0.0f, cb = 0.0f;
                        if (own) { ea = PL_ALPHA(cell); eo = PL_OTHER(cell); cr = PL_COL(cell, 0); cg = PL_COL(cell, 1); cb = PL_COL(cell, 2); }
                        SLOT_SHIFT_ALL(om)
                        ea = lane_value(ea, ol); eo = lane_value(eo, ol);
                        cr = lane_value(cr, ol); cg = lane_value(cg, ol); cb = lane_value(cb, ol);
                        const bool dup_ = PIECES && ((ek | kCellMask) == last_key); // the same event again (another piece of a split particle)
                        if (COUNT && tally && !dup_) c.hit_evals++;
                        last_key = ek | kCellMask;
                        if (!dup_ && a.p.alpha_min < ea) { // shaders/tracer.cuh:352-367
                            radiance = add3(radiance, mul3s(mul3s(mk3(cr, cg, cb), T), ea));
                            T *= (1.0f - ea);
                        }
                        if (!(T > minT)) alive = false;
                        const bool rekey = own && !dup_ && ((((uint32_t)ek) & 32u) == 0u) && (eo < t_hi);
                        const uint64_t nk = rekey ? (mk_skey(eo, id, 1) | (uint64_t)cell) : kKeyInvalid;
                        pmask = (own && !rekey) ? (pmask & ~(1u << cell)) : pmask;
                        if (wave_any(rekey)) { // wave-uniform branch
                            if (rekey) PL_OTHER(cell) = INFINITY;
                            SLOT_INSERT(nk) // a slot was just freed: it fits
                        }
                    }
                }
                if (!SINGLE && !dfs) {
                    // a lane's next event may be composited when its key lies below ONE limit: the smallest of the front (as a
                    // key: t < F <=> key < F's bits << 32, both non-negative), the lane's bag minimum and its cut-off.  The limit
                    // changes once per trip of the outer loop (and after a refill); the test below runs once per compositing step
                    // and once more per trip: one 64-bit compare instead of four compares (k0 = ~0, a free slot, is never below it).
                    // (QUAD: the RAY's next event is the smallest first key of its four windows, its limit the smallest of the four lanes' — q0_ and
                    //  limk_ are alike in the lanes of a quad, and so is everything derived from them)
                    const uint64_t fkey_ = (uint64_t)__float_as_uint(F) << 32;
                    uint64_t limk_ = (bagmin < lost) ? bagmin : lost;
                    if (QUAD) limk_ = quad_umin64(limk_);
                    limk_ = (fkey_ < limk_) ? fkey_ : limk_;
                    while (true) {
                        uint64_t q0_ = QUAD ? quad_umin64(k0) : k0;
                        bool can_ = alive & (q0_ < limk_);
                        uint64_t cm_ = wave_ballot(alive) & wave_ballot(q0_ < limk_);
                        // a lane whose next final event sits in its bag needs a refill before it can go on
                        const bool need = bags && alive && (nb != 0u) && !can_ && (key_t(bagmin) < F) && (bagmin < lost) &&
                                          ((q0_ == kKeyInvalid) || (q0_ >= bagmin));
                        const uint64_t nm_ = bags ? wave_ballot(need) : 0ull;
                        if (!(cm_ | nm_)) break;
                        // A sweep starts — and goes on — while enough lanes can take part: half of the WANTING lanes, at most
                        // ready_min (a tile down to a few wanting lanes never has ready_min of them ready: its last rays would go
                        // on gathering events they end before, until a window filled), or a window is nearly full, or one of the
                        // ready lanes is past half of its transmittance (a ray near its end keeps the tile's frustum and cut-off
                        // open for as long as it waits; a fresh one does not: C2 1.51 -> 1.04 ms, C3 -1 %, C5 +0.5 %).  The
                        // condition is taken again before every step: the tail of a sweep, one or two lanes per step, was a
                        // third of the 100 k frame.
                        if (!done) {
                            const uint32_t rmin_ = min(ready_min, max(1u, (nact_cur + 1u) >> 1));
                            // (QUAD: the count is in lanes, four per ray — a ray that waits for the refill of ONE of its windows counts as a ray,
                            //  or the last ray of a quadrant, 1 lane against a threshold of 2, would wait for the end of the pass)
                            uint64_t nmq_ = nm_;
                            if (QUAD && nm_) { nmq_ |= (nmq_ & 0xAAAAAAAAAAAAAAAAull) >> 1; nmq_ |= (nmq_ & 0x5555555555555555ull) << 1;
                                               nmq_ |= (nmq_ & 0xCCCCCCCCCCCCCCCCull) >> 2; nmq_ |= (nmq_ & 0x3333333333333333ull) << 2; }
                            const bool go_ = ((uint32_t)__popcll(cm_ | nmq_) >= rmin_) ||
                                             wave_any(can_ && ((KPRESS != kKeyInvalid) || (T < kSweepEagerT)));
                            if (!go_) break;
                        }
                        if (!cm_) {
                            // ---- refill: one scan of the bags of the lanes in need; entry by entry, whatever is smaller
                            //      than the window's last key goes in (sorted insert) and the displaced last key takes
                            //      its place in the bag (compacted in place: position w <= i) ----
                            GRT_D(node_visits, 1)
                            // lanes that do not need it yet but have room for four more keys come along: one scan instead
                            // of one per lane a few steps apart
                            const bool rf = need || (alive && (nb != 0u) && (KROOM == kKeyInvalid) && (bagmin < lost));
                            uint32_t nmax = rf ? nb : 0u;
                            for (int off = 32; off > 0; off >>= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, off));
                            nmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmax);
                            work |= (nmax > kBagKeep2) ? 3u : ((nmax > kBagKeep1) ? 1u : 0u);
                            uint32_t w_ = 0;
                            uint64_t newmin = kKeyInvalid;
                            float4* bp = a.ovf_pool + (size_t)chunk * (kSub * 64u) + lane;
                            for (uint32_t i = 0; i < nmax; i++) {
                                const bool v_ = rf && (i < nb);
                                float4 e_ = make_float4(0.f, 0.f, 0.f, 0.f);
                                if (v_) e_ = bp[(size_t)i * 64u];
                                const uint64_t ekey = ((uint64_t)__float_as_uint(e_.y) << 32) | (uint64_t)__float_as_uint(e_.x);
                                const bool fullw = KLAST != kKeyInvalid;
                                const bool tk = v_ && (!fullw || (ekey < KLAST));
                                float4 st_ = e_; // what stays in the bag at position w
                                if (wave_any(tk)) { // wave-uniform branch
                                    const uint32_t lcell = (uint32_t)(KLAST & kCellMask);
                                    if (tk && fullw) { // the displaced last key, with its payload
                                        const uint64_t dk = KLAST | kCellMask;
                                        st_ = make_float4(__uint_as_float((uint32_t)dk), __uint_as_float((uint32_t)(dk >> 32)),
                                                          PL_OTHER(lcell), PL_ALPHA(lcell));
                                    }
                                    const uint32_t cell = fullw ? lcell : (uint32_t)__builtin_ctz(~pmask);
                                    KLAST = (tk && fullw) ? kKeyInvalid : KLAST;
                                    pmask = tk ? (pmask | (1u << cell)) : pmask;
                                    if (tk) { PL_OTHER(cell) = e_.z; PL_ALPHA(cell) = e_.w; }
                                    SLOT_INSERT(tk ? ((ekey & ~kCellMask) | (uint64_t)cell) : kKeyInvalid)
                                }
                                const bool keep = v_ && (!tk || fullw);
                                if (keep) {
                                    bp[(size_t)w_ * 64u] = st_;
                                    const uint64_t sk = ((uint64_t)__float_as_uint(st_.y) << 32) | (uint64_t)__float_as_uint(st_.x);
                                    newmin = (sk < newmin) ? sk : newmin;
                                    w_++;
                                }
                            }
                            if (rf) {
                                nb = w_;
                                bagmin = newmin;
                            }
                            // (the refill and the compositing step are two if-thens in a row, not the arms of an if / else:
                            //  the arms of a structurised if / else keep BOTH versions of the window alive, 27 register copies
                            //  per compositing step)
                            limk_ = (bagmin < lost) ? bagmin : lost;
                            if (QUAD) limk_ = quad_umin64(limk_);
                            limk_ = (fkey_ < limk_) ? fkey_ : limk_;
                            q0_ = QUAD ? quad_umin64(k0) : k0;
                            can_ = alive & (q0_ < limk_);
                            cm_ = wave_ballot(alive) & wave_ballot(q0_ < limk_);
                        }
                        if (!cm_) continue;
                        GRT_D(hit_evals, 1)
                        const uint64_t ek = q0_;
                        const uint32_t cell = (uint32_t)(ek & kCellMask);
                        const uint32_t id = skey_id(ek);
                        float ea = 0.0f, eo = INFINITY, T_old = 0.0f;
                        bool blend_ = false;
                        float4 cc = make_float4(0.f, 0.f, 0.f, 0.f);
                        // QUAD: the event sits in ONE of the ray's four windows (a particle is tested by one slot per pass: the keys of a ray
                        // are distinct); that lane pops it, all four read its payload cell and take the same compositing step
                        const bool own_ = !QUAD || (k0 == q0_);
                        const uint64_t ownm_ = QUAD ? (cm_ & wave_ballot(k0 == q0_)) : cm_;
                        uint32_t ol_ = lane; // the lane whose window holds the event
                        if (QUAD) ol_ = (lane & 60u) + (uint32_t)__builtin_ctz(((uint32_t)(ownm_ >> (lane & 60u)) & 15u) | 16u);
                        if (can_) { // payload from LDS and (degree 0) the colour, both in flight while the window is popped
                            ea = QUAD ? pl_alpha[cell * kWG + ol_] : PL_ALPHA(cell); eo = QUAD ? pl_other[cell * kWG + ol_] : PL_OTHER(cell);
                            if (!SH) cc = a.color0[id];
                        }
                        SLOT_SHIFT_ALL(ownm_)
#ifdef GRT_TILE_CHECK
                        if (can_ && a.outf && lane == GRT_TILE_CHECK_LANE && dbg_n < 1900u) { // event log of one lane
                            a.outf[dbg_n * 3] = key_t(ek); a.outf[dbg_n * 3 + 1] = (float)(id * 2u + ((((uint32_t)ek) >> 5) & 1u)); a.outf[dbg_n * 3 + 2] = T;
                            dbg_n++;
                        }
#endif
                        // equal keys meet in the window when a split particle was inserted through two of its pieces: the first is
                        // composited, the repeat only gives its cell back
                        const bool dup_ = PIECES && can_ && ((ek | kCellMask) == last_key);
                        const uint64_t dupm_ = PIECES ? wave_ballot((ek | kCellMask) == last_key) : 0ull;
                        if (can_ && !dup_) { // shaders/tracer.cuh:352-367
                            if (COUNT && (!QUAD || own_)) c.hit_evals++;
                            last_key = ek | kCellMask; // nothing with the same (t, id, exit) can compare above it
                            if (a.p.alpha_min < ea) {
                                if (!SH) { // degree 0: the colour load is still in flight; its use waits until the re-key is done
                                    blend_ = true;
                                    T_old = T;
                                } else {
                                    f3 dl = dn; // keep the SH basis out of loop-invariant hoisting (it would spill)
                                    asm volatile("" : "+v"(dl.x), "+v"(dl.y), "+v"(dl.z));
                                    const f3 L = sh_radiance(a.sh + (size_t)id * 48, dl, a.p.sh_degree_max);
                                    radiance = add3(radiance, mul3s(mul3s(L, T), ea));
                                }
                                T *= (1.0f - ea);
                            }
                            if (!(T > minT)) alive = false;
                        }
                        // an entry whose exit lies inside the segment is re-keyed to its exit event and keeps its
                        // payload cell, otherwise the cell is released
                        const bool rekey = can_ && own_ && !dup_ && ((((uint32_t)ek) & 32u) == 0u) && (eo < t_hi);
                        const uint64_t nk = rekey ? (mk_skey(eo, id, 1) | (uint64_t)cell) : kKeyInvalid;
                        pmask = (can_ && own_ && !rekey) ? (pmask & ~(1u << cell)) : pmask;
                        const uint64_t rkm_ = ownm_ & ~dupm_ & vote_eq_u32(((uint32_t)ek) & 32u, 0u) & vote_lt_f32(eo, t_hi);
                        if (rkm_) { // wave-uniform branch
                            if (rekey) PL_OTHER(cell) = INFINITY;
                            SLOT_INSERT(nk) // a slot was just freed: it fits
                        }
                        // (same value, (L T) alpha per channel with the T of before the event; placed here so that the gather of
                        //  color0 has the window pop and the re-key to hide behind)
                        if (!SH && blend_) radiance = add3(radiance, mul3s(mul3s(mk3(cc.x, cc.y, cc.z), T_old), ea));
                    }
                }
                if (done) break;
                // watchdog: never reached by design; a reported failure beats a hung GPU.  HOW it is reported matters: any
                // store to an error word from this kernel (in the loop, behind it, atomic or plain) re-shuffled the register
                // allocation of the hot loop and cost MODE 0 12-16 % (2.18 -> 2.44-2.53 ms on C3, six formulations measured).
                // So a camera-ray tile's reasons travel in the cost word it writes anyway (iters > max_iters = watchdog, high
                // bits = stack guard / stalled passes) and k_check_costs (grt_api.hip) turns them into the sticky error
                // word right behind the frame.  A bundle (MODE 1) in trouble gives up as if over budget and its rays go one
                // per wave; that last resort (MODE 2: 2 waves per SIMD, registers to spare) reports directly.
                if (++iters > a.max_iters) {
                    if (MODE == 1) { aborted = true; break; } // a bundle gives up as if over budget: its rays go one per wave
                    c.stall_exits += alive ? 1u : 0u;
                    if (SINGLE && lane == 0u && wave_any(alive)) atomicOr(a.err_word, kErrWatchdog);
                    watchdog = true;
                    break;
                }
                if (MODE == 1 && iters + work > a.bundle_budget) { // not a bundle worth the name: it is split, or its rays go one per wave
                    aborted = true;
                    break;
                }

                // ---- one step: the entries at the front, four lanes each.  LEAF step: leaf ranges -> their particles'
                //      boxes are culled here and the survivors slab-tested at once (lanes = rays).  NODE step: internal
                //      nodes -> their children's boxes are culled and the survivors join the frontier ----
                const bool occ_l = fr != kNoRoot;
                const bool rng_l = occ_l && ((fr & kLeafBit) != 0u);
                bool leaf_step;
                uint32_t nref; // the entry this lane's group expands
                uint32_t ngrp; // groups in this step (wave-uniform): group g is valid when g < ngrp
                uint32_t g, j; // group of this lane and its child slot in the group: 4 lanes per leaf range, kTileWide per node
                if (cur != kNoRoot) { // depth-first mode: one entry
                    leaf_step = (cur & kLeafBit) != 0u;
                    g = leaf_step ? (lane / kLeafLanes) : (lane / kTileWide);
                    j = leaf_step ? (lane % kLeafLanes) : (lane % kTileWide);
                    nref = cur;
                    ngrp = 1u;
                } else {
                    // (no reductions here: the frontier minimum Ff of the loop top and two votes decide the step)
                    // (votes on compound conditions as ANDs of votes on single compares: the vote of an AND goes through a
                    //  0 / 1 register and a second compare, two VALU operations each)
                    const uint64_t occm_ = wave_ballot(occ_l), rngm_ = wave_ballot(rng_l);
                    const bool have_rng = rngm_ != 0ull;
                    // a nearly full frontier takes leaf steps whatever lies in front (testing particles early is always
                    // legal; spilling children to the depth-first stack stalls the front)
                    const uint32_t nocc = (uint32_t)__popcll(occm_);
                    const bool crowded = (nocc > 64u - a.tile_reserve) && have_rng;
                    // nodes within the look-ahead of the FRONT are expanded first, so that leaf steps find full batches;
                    // then the nearest ranges (within a band behind the nearest one) are tested together
                    const float hz = F + (PIECES ? fmaxf(F * look_, a.tile_band_abs) : F * look_);
                    const bool node_near = (occm_ & ~rngm_ & wave_ballot(fl <= hz)) != 0ull;
                    leaf_step = have_rng && (!node_near || crowded);
                    // the nearest range / node: the frontier minimum when it is of that kind (the common case), else one
                    // reduction
                    float Fr = Ff_cur, Fn = Ff_cur;
                    if (leaf_step && !(rngm_ & wave_ballot(fl <= Ff_cur))) Fr = wave_min(rng_l ? fl : INFINITY);
                    if (!leaf_step && !node_near) Fn = wave_min(rng_l ? INFINITY : fl);
                    // (a tree with pieces holds sheets and needles that reach up to the eye: near the eye a band RELATIVE to the
                    //  front is a sliver and a leaf step would take one range at a time — there the band has an absolute floor)
                    const float tau = leaf_step ? (Fr + (PIECES ? fmaxf(Fr * band_, a.tile_band_abs) : Fr * band_)) : fmaxf(hz, Fn);
                    const bool cand = occ_l && (rng_l == leaf_step);
                    // a node step frees one slot per node and may need four: expand only what is sure to fit (at least
                    // one node: a frontier full of internal nodes overflows to the depth-first stack)
                    const uint32_t maxb = leaf_step ? kBatch : max(min(64u / kTileWide, (64u - nocc) / (kTileWide - 1u)), 1u);
                    g = leaf_step ? (lane / kLeafLanes) : (lane / kTileWide);
                    j = leaf_step ? (lane % kLeafLanes) : (lane % kTileWide);
                    float th = tau;
                    const uint64_t candm_ = leaf_step ? rngm_ : (occm_ & ~rngm_);
                    uint64_t sm = candm_ & wave_ballot(fl <= th);
                    if ((uint32_t)__popcll(sm) > maxb) {
                        // more candidates than the step can take: the NEAREST ones go first (four bisection steps on
                        // the distance threshold; lane order only breaks what is left of the tie)
                        float lo_ = wave_min(cand ? fl : INFINITY), hi_ = tau; // the nearest candidate itself always qualifies
                        // (MODE 2 looks far ahead: bisect between the nearest and the farthest candidate, twice as finely)
                        if (SINGLE) hi_ = fminf(tau, uni(wave_fmax((cand && (fl <= tau)) ? fl : 0.0f)));
                        for (int it = 0; it < kBisect; it++) {
                            const float mid = 0.5f * (lo_ + hi_);
                            const uint32_t n_ = (uint32_t)__popcll(candm_ & wave_ballot(fl <= mid));
                            const bool few = n_ <= maxb;
                            lo_ = few ? mid : lo_;
                            hi_ = few ? hi_ : mid;
                            if (few && n_ * 2u >= maxb && it >= 3) break;
                        }
                        th = lo_;
                        sm = candm_ & wave_ballot(fl <= th);
                    }
                    const bool selm = cand && (fl <= th);
                    const uint32_t rk = lanes_below(sm);
                    const bool sel = selm && (rk < maxb);
                    const uint32_t cnt = min((uint32_t)__popcll(sm), maxb);
                    if (sel) xsel[rk] = fr;
                    fl = sel ? INFINITY : fl;
                    fr = sel ? kNoRoot : fr;
                    wave_fence();
                    nref = xsel[g];
                    ngrp = cnt;
                }
                const uint32_t first = leaf_first(nref);
                // (one compare per condition, made where it is voted on: a condition that arrives from another block as a
                //  bool is voted on through a 0 / 1 register and a second compare)
                const uint32_t jmax_ = leaf_step ? leaf_count(nref) : kTileWide;
                const bool gv = g < ngrp;
                const bool cv = gv & (j < jmax_);
                float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
                if (cv) {
                    const float4* src = leaf_step ? (a.pbox + (size_t)(first + j) * 2) : (a.qnodes + (size_t)nref * (2u * kTileWide) + j * 2u);
                    b0 = src[0];
                    b1 = src[1];
                }
                // QUAD, leaf step: the candidates' RECORDS are asked for together with their boxes — speculatively, for the ones the cull
                // will drop too: a second, dependent round trip to memory per leaf step is what bounds a heavy tile's wave (a vector load
                // returns after ~0.5 us; the first form of this kernel, which fetched the survivors' records behind the cull, ran a
                // 256^2 frame in 0.67 ms instead of 0.46)
                float4 rq0 = b0, rq1 = b0, rq2 = b0, rq3 = b0;
                if (QUAD && leaf_step && cv) {
                    const float4* rp = a.rec + (size_t)(first + j) * 4;
                    rq0 = rp[0]; rq1 = rp[1]; rq2 = rp[2]; rq3 = rp[3];
                }
                const uint32_t cref = leaf_step ? (first + j) : __float_as_uint(b0.w); // particle index / child ref
                const bool valid = cv && (cref != kNoRoot);
                if (COUNT && valid) c.node_visits++; // one 32-B child box per lane
                if (lim_dirty) { // a window overflowed: lanes past their cutoff want nothing any more
                    const float ct2_ = (lost != kKeyInvalid) ? key_t(lost) : t_hi_m;
                    LIM = uni(wave_fmax(alive ? ct2_ : 0.0f));
                    lim_dirty = false;
                }
                // box relative to the eye
                const float lx_ = b0.x - oc.x, ly_ = b0.y - oc.y, lz_ = b0.z - oc.z;
                const float hx_ = b1.x - oc.x, hy_ = b1.y - oc.y, hz_ = b1.z - oc.z;
                // four frustum planes (unit normals n), everything times two: 2 n.centre + min(|n|.extent, 2 radius) is twice the
                // farthest reach of (box AND bounding sphere) along n.  hi.w = the radius of a sphere about the box centre that
                // holds the primitive (a proxy's vertices: grt_api.hip k_proxy_boxes; +inf for child boxes and pieces): for a round
                // proxy the box corner reaches up to sqrt 3 times further along an oblique normal than the proxy does, a third
                // of the particles a leaf step used to fetch.  The slack covers the rounding of the sums and products:
                // 2e-5 x the L1 size of the box about the eye
                const float epsM = -2e-5f * (((fabsf(lx_) + fabsf(hx_)) + (fabsf(ly_) + fabsf(hy_))) + (fabsf(lz_) + fabsf(hz_)));
                const float cx_ = lx_ + hx_, cy_ = ly_ + hy_, cz_ = lz_ + hz_;
                const float gx_ = hx_ - lx_, gy_ = hy_ - ly_, gz_ = hz_ - lz_;
                const float rs_ = b1.w + b1.w;
                // (shadows of the pass-level values: read back from LDS; the near side is the box's hi side where the slab factor is negative)
                const float4 fq0 = ((const float4*)fr_lds)[0], fq1 = ((const float4*)fr_lds)[1], fq2 = ((const float4*)fr_lds)[2], fq3 = ((const float4*)fr_lds)[3];
                const float pLx = fq0.x, pLy = fq0.y, pLz = fq0.z, pRx = fq0.w, pRy = fq1.x, pRz = fq1.y, pBx = fq1.z, pBy = fq1.w;
                const float pBz = fq2.x, pTx = fq2.y, pTy = fq2.z, pTz = fq2.w, ivx = fq3.x, ivy = fq3.y, ivz = fq3.z;
                const bool shx = ivx < 0.0f, shy = ivy < 0.0f, shz = ivz < 0.0f;
#define GRT_PSIDE(P, MP)                                                                                   \
                ((__builtin_fmaf(P##x, cx_, __builtin_fmaf(P##y, cy_, P##z * cz_)) +                        \
                  fminf(__builtin_fmaf(fabsf(P##x), gx_, __builtin_fmaf(fabsf(P##y), gy_, fabsf(P##z) * gz_)), rs_)) >= \
                 (BUNDLE ? epsM + 2.0f * (MP) : epsM))
                // (all four, no short circuit: a plane test costs the wave the same for one lane as for 64, and the votes on the
                //  single compares AND together for nothing)
                const bool in0_ = GRT_PSIDE(pL, mL), in1_ = GRT_PSIDE(pR, mR), in2_ = GRT_PSIDE(pB, mB), in3_ = GRT_PSIDE(pT, mT);
                const bool inside = in0_ & in1_ & in2_ & in3_;
                const uint64_t insidem_ = wave_ballot(in0_) & wave_ballot(in1_) & wave_ballot(in2_) & wave_ballot(in3_);
#undef GRT_PSIDE
                // lower bound of t over the tile: Euclidean distance to the box, and the per-axis slab bound
                const float ex_ = fmaxf(fmaxf(lx_, -hx_), 0.0f), ey_ = fmaxf(fmaxf(ly_, -hy_), 0.0f),
                            ez_ = fmaxf(fmaxf(lz_, -hz_), 0.0f);
                // (v_sqrt_f32, 1 ulp, instead of the correctly rounded sqrtf and its 15 instructions: a bound that is cut by 2e-6
                //  below; a denormal argument gives 0, a smaller bound still)
                float euc = __builtin_amdgcn_sqrtf(__builtin_fmaf(ex_, ex_, __builtin_fmaf(ey_, ey_, ez_ * ez_)));
                float sx_, sy_, sz_;
                if (BUNDLE) { // |d| t >= dist(o, box) >= dist(oc, box) - |o - oc|;  t >= (side - o.x) / d.x per axis
                    euc = fmaxf(euc - rmax, 0.0f) * idmax;
                    sx_ = ((shx ? hx_ : lx_) - ofx) * ivx; sy_ = ((shy ? hy_ : ly_) - ofy) * ivy; sz_ = ((shz ? hz_ : lz_) - ofz) * ivz;
                } else {
                    sx_ = (shx ? hx_ : lx_) * ivx; sy_ = (shy ? hy_ : ly_) * ivy; sz_ = (shz ? hz_ : lz_) * ivz;
                }
                float lam = fmaxf(fmaxf(euc, sx_), fmaxf(sy_, sz_)) * (1.0f - 2e-6f);
                lam = fmaxf(lam, F); // never below the current front (keeps the frontier monotone)
                bool want = valid & inside & (lam <= LIM);
                uint64_t wm = wave_ballot(g < ngrp) & wave_ballot(j < jmax_) & wave_ballot(cref != kNoRoot) & insidem_ & wave_ballot(lam <= LIM);
                if (LO > 0.0f) { // later passes: skip what ends before the restart point
                    const float fx_ = fmaxf(fabsf(lx_), fabsf(hx_)), fy_ = fmaxf(fabsf(ly_), fabsf(hy_)),
                                fz_ = fmaxf(fabsf(lz_), fabsf(hz_));
                    float far = sqrtf(__builtin_fmaf(fx_, fx_, __builtin_fmaf(fy_, fy_, fz_ * fz_))) * (1.0f + 2e-6f);
                    if (BUNDLE) far = (far + rmax) * idmin;
                    want = want && (far >= LO);
                    wm &= wave_ballot(far >= LO);
                }

                if (leaf_step) {
                    GRT_D(fetches, 1)
                    if (MODE == 0 || QUAD) work += kCostFetch * (uint32_t)__popcll(wm);
                    if (QUAD && wm) { // the survivors' records to LDS, each by the lane that culled its box: slot s of every ray reads survivor s's
                        if (want) { qstg[lane * 4u] = rq0; qstg[lane * 4u + 1u] = rq1; qstg[lane * 4u + 2u] = rq2; qstg[lane * 4u + 3u] = rq3; }
                        wave_fence();
                    }
                    // ---- exact tests of the surviving particles, all lanes = rays (grt_render_stream's arithmetic) ----
                    bool trip = wm != 0ull; // MODE 2: ONE trip, lanes = particles
                    const uint64_t alivem_ = wave_ballot(alive); // (nothing in this loop changes it)
                    while (SINGLE ? trip : (wm != 0ull)) {
                        trip = false;
                        float4 r0, r1, r2, r3, e0, e1, e2, e3;
                        bool act_; // lanes the exact test is meant for
                        if (SINGLE) { // every surviving lane fetches and tests ITS particle
                            act_ = want && alive;
                            r0 = r1 = r2 = r3 = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (act_) {
                                const float4* rp = a.rec + (size_t)cref * 4;
                                r0 = rp[0]; r1 = rp[1]; r2 = rp[2]; r3 = rp[3];
                            }
                            if (COUNT) c.fetches += 4u * (uint32_t)__popcll(wm);
                            wm = 0ull;
                        } else if (QUAD) { // up to four survivors at once: slot s of every ray fetches and tests survivor s
                            const uint32_t p0_ = (uint32_t)__builtin_ctzll(wm);
                            wm = clear_bit64(wm, p0_);
                            uint32_t p1_ = p0_, p2_ = p0_, p3_ = p0_, nsv = 1u; // (the survivors' lanes = their slots of the staged records)
                            if (wm) {
                                p1_ = (uint32_t)__builtin_ctzll(wm); wm = clear_bit64(wm, p1_); nsv = 2u;
                                if (wm) {
                                    p2_ = (uint32_t)__builtin_ctzll(wm); wm = clear_bit64(wm, p2_); nsv = 3u;
                                    if (wm) { p3_ = (uint32_t)__builtin_ctzll(wm); wm = clear_bit64(wm, p3_); nsv = 4u; }
                                }
                            }
                            const uint32_t sl_ = lane & 3u;
                            const uint32_t sv_ = (sl_ == 0u) ? p0_ : ((sl_ == 1u) ? p1_ : ((sl_ == 2u) ? p2_ : p3_));
                            act_ = alive && (sl_ < nsv);
                            r0 = qstg[sv_ * 4u]; r1 = qstg[sv_ * 4u + 1u]; r2 = qstg[sv_ * 4u + 2u]; r3 = qstg[sv_ * 4u + 3u];
                            if (COUNT) c.fetches += 4u * nsv;
                        } else {
                            const uint32_t b = (uint32_t)__builtin_ctzll(wm);
                            wm = clear_bit64(wm, b);
                            const uint32_t pidx = (uint32_t)__builtin_amdgcn_readlane((int)cref, (int)b);
                            // (a 32-bit BYTE offset: the two 64-B scalar loads take it as their SGPR offset, no 64-bit address
                            //  arithmetic; the launcher sends scenes of 2^26 primitives and more elsewhere, kTileMaxPrims)
                            const uint32_t roff = pidx << 6;
                            sload64(a.rec, roff, r0, r1, r2, r3);
                            if (!BUNDLE) sload64(a.erec, roff, e0, e1, e2, e3);
                            if (COUNT) c.fetches += BUNDLE ? 4 : 8; // wave-uniform: 64-B record (+ 64-B eye record), in 16-B units
                            act_ = alive;
                        }
                        GRT_D(segments, 1)
                        if (MODE == 1) work++;
                        const f3 mu = mk3(r0.x, r0.y, r0.z);
                        m33 A;
                        A.a[0] = r1.x; A.a[1] = r1.y; A.a[2] = r1.z;
                        A.a[3] = r2.x; A.a[4] = r2.y; A.a[5] = r2.z;
                        A.a[6] = r3.x; A.a[7] = r3.y; A.a[8] = r3.z;
                        // A (o - mu): from the eye record (wave-uniform), or per lane for a bundle / a quad's own particle (the very operation
                        // sequence the eye records were made with: the same bits)
                        const f3 o_g = (BUNDLE || QUAD) ? matvec(A, sub3(o, mu)) : mk3(e0.x, e0.y, e0.z);
                        const float cc_ = (BUNDLE || QUAD) ? proxy_sphere_cc(o_g, r0.w) : e0.w;
                        const f3 d_g = matvec(A, d);
                        {   // conservative sphere pre-test (proxy_sphere_maybe_pre) as lane masks
                            const float b_ = dot3(o_g, d_g), aa_ = dot3(d_g, d_g);
                            const uint64_t m_ = (wave_ballot(cc_ <= 0.0f) | wave_ballot(b_ * b_ * (1.0f + 4e-6f) >= aa_ * cc_)) &
                                                ((SINGLE || QUAD) ? wave_ballot(act_) : alivem_);
                            if (!m_) continue;
                            // (QUAD: the cost word counts a trip's particles as the camera-ray kernel counts them — those some lane can touch —
                            //  so that a tile costs the same word on either kernel and the launch order splits the same tiles)
                            if (QUAD) {
                                uint32_t f_ = (uint32_t)m_ | (uint32_t)(m_ >> 32);
                                f_ |= f_ >> 16; f_ |= f_ >> 8; f_ |= f_ >> 4;
                                work += kCostTest * (uint32_t)__builtin_popcount(f_ & 15u);
                            }
                        }
                        if (COUNT && act_) c.proxy_tests++;
                        if (MODE == 1) work += 2u;
                        if (MODE == 0) work += kCostTest;
                        GRT_D(proxy_tests, 1)
                        float te, tx;
                        float pa[10]; // slab_project(o_g)
                        if (BUNDLE || QUAD) {
                            slab_project(o_g, pa);
                        } else {
                            pa[0] = e1.x; pa[1] = e1.y; pa[2] = e1.z; pa[3] = e1.w; pa[4] = e2.x; pa[5] = e2.y; pa[6] = e2.z;
                            pa[7] = e2.w; pa[8] = e3.x; pa[9] = e3.y;
                        }
                        const bool hit = proxy_slabs_pre(pa, d_g, r0.w, te, tx) && act_;
                        // ---- the particle's events into the lanes' windows: keys of the entry / exit events inside the lane's interval,
                        //      piece ownership, the response (computed only when some lane inserts), window overflow into the lane's bag,
                        //      sorted insert.  (> last_key, not just > pass_lo, with pieces: a particle that entered the tree as several pieces
                        //      is met once per piece the tile crosses, with the same keys; float compares first: te / tx may be negative or
                        //      NaN, the unsigned key compares assume t > 0; alpha does not depend on the hit distance, shaders/tracer.cuh:
                        //      354-357; window full: the largest pending key leaves — into the lane's bag in global memory, or for good: the
                        //      lane is then lossy beyond it.)
                        const uint32_t id = __float_as_uint(r2.w);
                        const uint64_t ke = mk_skey(te, id, 0), kx = mk_skey(tx, id, 1);
                        const uint64_t seen_ = PIECES ? last_key : pass_lo;
                        bool in_e = hit && (te >= t_lo) && (te < t_hi) && (ke > seen_);
                        bool in_x = hit && (tx >= t_lo) && (tx < t_hi) && (kx > seen_);
                        const uint32_t cellb = PIECES ? __float_as_uint(r3.w) : 0u;
                        if (PIECES && cellb) {
                            const bool own_ = piece_owns(cellb, r0.w, o_g, d_g, in_e ? te : tx);
                            in_e = in_e && own_;
                            in_x = in_x && own_;
                        }
                        const uint64_t k_first = in_e ? ke : (in_x ? kx : kKeyInvalid);
                        const bool ins = (k_first != kKeyInvalid) && (k_first < lost);
                        GRT_TILE_CHECK_FRONT(ins, k_first)
                        if (wave_any(ins)) {
                            const float alpha = fminf(0.99f, response_from(A, mu, o, d, o_g, d_g) * r1.w);
                            const float other = (in_e && in_x) ? tx : INFINITY;
                            const bool full = KLAST != kKeyInvalid;
                            const bool take = ins && (!full || k_first < KLAST);
                            const bool drop = ins && full;
                            const uint32_t cell = full ? (uint32_t)(KLAST & kCellMask) : (uint32_t)__builtin_ctz(~pmask);
                            if (wave_any(drop)) {
                                if (!SINGLE && chunk == kNoRoot) {
                                    // (as many chunks as the bags may grow to: QUAD — a ray's four bags hold a quarter each — one always)
                                    const uint32_t nch = (GRT_BAG_CAP + kSub - 1u) / kSub;
                                    uint32_t ch = 0;
                                    if (lane == 0u) ch = atomicAdd(a.ovf_next, nch);
                                    ch = (uint32_t)__builtin_amdgcn_readfirstlane((int)ch);
                                    chunk = (ch + nch <= a.ovf_chunks) ? ch : (kNoRoot - 1u);
                                }
                                const uint64_t dk = take ? (KLAST | kCellMask) : (k_first | kCellMask);
                                // the tile has outgrown its size class: a full bag elsewhere in the pool, the rays' entries move over
                                if (MODE == 0 && GRT_BAG_CAP < a.ovf_entries && !(bag_cap >> 16) && chunk < a.ovf_chunks &&
                                    wave_any(drop && (nb >= GRT_BAG_CAP) && (dk < lost))) {
                                    uint32_t ch = 0;
                                    if (lane == 0u) ch = atomicAdd(a.ovf_next, kOvf / kSub);
                                    ch = (uint32_t)__builtin_amdgcn_readfirstlane((int)ch);
                                    if (ch + kOvf / kSub <= a.ovf_chunks) {
                                        uint32_t nmx = nb;
                                        for (int off = 32; off > 0; off >>= 1) nmx = max(nmx, (uint32_t)__shfl_xor((int)nmx, off));
                                        nmx = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmx);
                                        const float4* src_ = a.ovf_pool + (size_t)chunk * (kSub * 64u) + lane;
                                        float4* dst_ = a.ovf_pool + (size_t)ch * (kSub * 64u) + lane;
                                        for (uint32_t i = 0; i < nmx; i++)
                                            if (i < nb) dst_[(size_t)i * 64u] = src_[(size_t)i * 64u];
                                        chunk = ch;
                                        bag_cap = a.ovf_entries;
                                    } else {
                                        bag_cap |= 0x10000u;
                                    }
                                }
                                const bool to_bag = !SINGLE && drop && (chunk < a.ovf_chunks) && (nb < GRT_BAG_CAP) && (dk < lost);
                                if (to_bag) {
                                    const float d_o = take ? PL_OTHER(cell) : other, d_a = take ? PL_ALPHA(cell) : alpha;
                                    a.ovf_pool[((size_t)chunk * kSub + nb) * 64u + lane] =
                                        make_float4(__uint_as_float((uint32_t)dk), __uint_as_float((uint32_t)(dk >> 32)), d_o, d_a);
                                    nb++;
                                    bagmin = (dk < bagmin) ? dk : bagmin;
                                }
                                const bool gone = drop && !to_bag;
                                lost = (gone && (dk < lost)) ? dk : lost;
                                if (SINGLE) lost = wave_umin64(lost);
                                else bags = true;
                                if (QUAD) lost = quad_umin64(lost); // the cut-off is the RAY's: an event one of its windows lost bounds all four
                                if (wave_any(gone)) lim_dirty = true;
                            }
                            KLAST = (take && full) ? kKeyInvalid : KLAST;
                            pmask = take ? (pmask | (1u << cell)) : pmask;
                            if (take) { PL_OTHER(cell) = other; PL_ALPHA(cell) = alpha; }
                            if (SINGLE && take) {
                                f3 L;
                                if (!SH) {
                                    const float4 cc = a.color0[id];
                                    L = mk3(cc.x, cc.y, cc.z);
                                } else {
                                    L = sh_radiance(a.sh + (size_t)id * 48, dn, a.p.sh_degree_max);
                                }
                                PL_COL(cell, 0) = L.x; PL_COL(cell, 1) = L.y; PL_COL(cell, 2) = L.z;
                            }
                            SLOT_INSERT(take ? (k_first | (uint64_t)cell) : kKeyInvalid)
                        }
                    }
                    continue; // (the step is over: on to the next trip of the step loop)
                }
                GRT_D(rays, 1)
                // ---- node step: compaction of the wanted children into free frontier slots; what does not fit goes
                //      to the depth-first stack ----
                if (wm) {
                    const uint64_t fm = wave_ballot(fr == kNoRoot);
                    const uint32_t nc = (uint32_t)__popcll(wm), nf = (uint32_t)__popcll(fm);
                    const uint32_t crk = lanes_below(wm), frk = lanes_below(fm);
                    // (a rebalance may send every entry but one to the bag: frontier + bag + these must fit it)
                    if (nc > nf && !dfs && nbag + nc + 64u <= kBag) {
                        // no room: park all of them in the bag; the next iteration keeps the nearest entries of
                        // (frontier + bag) in registers
                        if (want) bag[nbag + crk] = make_uint2(__float_as_uint(lam), cref);
                        Fbag = fminf(Fbag, wave_min(want ? lam : INFINITY));
                        nbag += nc;
                        rebal = true;
                        wave_fence();
                    } else {
                        if (nc > nf && dsp + (nc - nf) > kStack) { // cannot happen for the tree heights the launcher admits
                            c.stall_exits += alive ? 1u : 0u;
                            if (MODE == 1) { aborted = true; break; }
                            watchdog = true;
                            iters |= kCostStackBit;
                            if (SINGLE && lane == 0u && wave_any(alive)) atomicOr(a.err_word, kErrStack);
                            break;
                        }
                        if (want) {
                            if (crk < nf) xch[crk] = make_uint2(__float_as_uint(lam), cref);
                            else dstack[dsp + (crk - nf)] = cref; // bag full too: depth-first from here on
                        }
                        wave_fence();
                        if ((fr == kNoRoot) && (frk < nc)) {
                            const uint2 v = xch[frk];
                            fl = __uint_as_float(v.x);
                            fr = v.y;
                        }
                        dsp += (nc > nf) ? (nc - nf) : 0u;
                        wave_fence();
                    }
                }
            }
            // a lane goes again only if it dropped something and still has transmittance left
            if (aborted) break;
            const bool progressed = last_key != pass_lo;
            stalls = parked ? stalls : (progressed ? 0u : stalls + 1u);
            const bool again = alive && (lost != kKeyInvalid);
            // HOW DEEP the bags got (0: no ray's bag held more than kBagKeep1 entries, 1: none more than kBagKeep2, 3: more — the tile wants
            // a full bag per ray next frame too) is noted in the two lowest bits of `work`, which the camera-ray kernel only ever raises by
            // multiples of four: at a prune, at a refill scan, and here — what the bags hold at the end of a pass was never scanned
            if (MODE == 0 && bags) work |= wave_any(nb > kBagKeep2) ? 3u : (wave_any(nb > kBagKeep1) ? 1u : 0u);
            if (COUNT && again && stalls >= 2u) c.stall_exits++;
            if (wave_any(again && stalls >= 2u)) { // (never seen)
                if (MODE == 1) { aborted = true; break; }
                iters |= kCostStallBit;
                if (SINGLE && lane == 0u) atomicOr(a.err_word, kErrStall);
            }
            alive = ((again && (stalls < 2u)) || parked) && !watchdog;
        }
        // (unit and part code are taken from the ONE scalar that lives across the passes, the order entry)
        // the cost word: steps + the weighted particle work (kCostFetch, kCostTest).  The watchdog's reading of it, "steps > max_iters"
        // (k_check_costs), stays exact: the word is kept at or below max_iters unless the watchdog fired.
        if (!BUNDLE && a.cost && lane == 0) {
            uint32_t cw = min((iters & kCostStepsMask) + (work >> 4), kCostStepsMask);
            // (max_iters <= kCostStepsMask - 1: grt_set_option; a stack-guard give-up has its own bit and is not a step watchdog)
            const bool over = watchdog && !(iters & kCostStackBit);
            cw = over ? max(cw, a.max_iters + 1u) : min(cw, a.max_iters);
            // (two lowest bits: how deep the bags got)
            if ((MODE == 0 || QUAD) && !over) { // (the part of a split tile: deep whatever its own bags did — the tile's word is the maximum of its parts')
                cw = (cw & ~3u) | ((QUAD || (ue >> 30) != 0u) ? 3u : (work & 3u));
                cw = (cw > a.max_iters && cw >= 4u) ? cw - 4u : cw; // (still at or below max_iters, the bits kept)
            }
            atomicMax(&a.cost[ue & kOrderUnitMask], (iters & (kCostStackBit | kCostStallBit)) | cw | (min(ue >> 30, 2u) << kCostPartShift));
        }
    }
    if (MODE == 1 && aborted) { // wave-uniform: nothing is written, the chunk's rays join the heavy list
        const uint64_t vm = wave_ballot(in_frame);
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(a.hcount, (uint32_t)__popcll(vm));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (in_frame) a.heavy[base + lanes_below(vm)] = (uint32_t)ent;
        if (a.bverdict && lane == 0) a.bverdict[a.qunit[unit_s]] = a.bverdict_epoch; // the tile is no bundle under THIS view: remembered (RenderArgs::bverdict)
        gave_up = true;
        break;
    }
    // (no hittable particle: the density is left as it came, shaders/tracer.cuh:328-373 never runs)
    density = (BUNDLE && a.root_ref == kNoRoot) ? density_in : 1.0f - T;
    cont = false;
    const uint32_t numBounces = (pflags >> 8) & 0xFFu;
    if (have_ray) {
        const float alpha = density;
        if (!MESH) {
            const f3 directLight = mul3s(radiance, alpha);     // shaders/tracer.cu:80
            col = add3(col, mul3s(directLight, 1.0f - 0.0f));  // shaders/tracer.cu:101 with blocking == 0
        } else {
            // first iteration of the bounce loop (shaders/tracer.cu:58-106) with all accumulators at zero
            const uint32_t state = pflags & 0xFFu;
            f3 directLight = mk3(0, 0, 0);
            if (state == 3u) { // Terminate: renderNormal, shaders/tracer.cuh:417-428
                accumColor = add3(accumColor, radiance);
                accumAlpha += alpha;
                const f3 normalColor = mul3s(add3(hitN, mk3(1.0f, 1.0f, 1.0f)), 0.5f);
                accumColor = add3(accumColor, mul3s(normalColor, 1.0f - alpha));
            } else {
                if (state == 0u) { // LastGaussianPass, shaders/tracer.cu:68-82
                    directLight = mul3s(radiance, alpha);
                    accumAlpha = clampf(accumAlpha + alpha, 0.0f, 1.0f);
                } else {           // shaders/tracer.cu:84-98
                    accumColor = add3(accumColor, mul3s(radiance, 1.0f - accumAlpha));
                    accumAlpha = clampf(accumAlpha + alpha, 0.0f, 1.0f);
                    blocking = clampf(blocking + alpha, 0.0f, 1.0f);
                }
                accumColor = add3(accumColor, mul3s(directLight, 1.0f - blocking)); // shaders/tracer.cu:101
                timeout += 1u;                                                      // shaders/tracer.cu:103-104
                cont = (length3(nextD) > 0.1f) && (numBounces < a.p.max_bounces) && !(timeout > kTimeoutIterations);
            }
            col = accumColor;
        }
    }
    if (!(SINGLE && cont)) break;
    // ---- MODE 2, next iteration: the mesh hit of the new ray.  All 64 lanes hold the same ray and walk the (small) mesh
    //      tree in step, on ONE stack (the depth-first overflow stack of the frontier is idle between segments) ----
    o = nextO;
    d = nextD;
    density_in = density;
    {
        uint32_t it_ = 0, nv_ = 0;
        const MeshHit mh = mesh_closest_t<COUNT, 1>(a, dstack, o, d, kTraceMeshTmin, kTraceMeshTmax, it_, nv_);
        if (COUNT && tally) c.node_visits += nv_;
        int st_ = MeshPass;
        uint32_t nb_ = numBounces;
        f3 nrm_;
        mesh_shade(a, mh, o, d, st_, seg_tmax, nrm_, nextO, nextD, nb_);
        hitN = nrm_;
        pflags = (uint32_t)st_ | (nb_ << 8) | (1u << 16);
    }
    have_ray = length3(d) > 0.1f;
    } // bounce loop
    if (gave_up) break;
    const uint32_t numBounces = (pflags >> 8) & 0xFFu;
    if (MESH) {
        // ---- the rays that go on: the wave takes ONE 64-entry chunk of the queue (one atomic) and every lane writes
        //      its own slot, so that stage 3 finds the rays of a tile together, as a bundle; bit 31 of the timeout word
        //      marks the slots that carry a ray ----
        const uint64_t mask = wave_ballot(cont);
        if (!SINGLE && mask) { // wave-uniform (MODE 2 never gets here with a ray that goes on)
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(a.qcount, 1u);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            float4* q = a.queue + ((size_t)base * 64u + lane) * 4;
            // whose chunk it is (bundle verdicts): the tile's number, and it travels with the tile's rays from round to round
            if (MODE == 0 && a.qunit && lane == 0) a.qunit[base] = ue & kOrderUnitMask;
            if (MODE == 1 && a.qunit_out && lane == 0) a.qunit_out[base] = a.qunit[unit_s];
            if (cont) {
                q[0] = make_float4(nextO.x, nextO.y, nextO.z, nextD.x);
                q[1] = make_float4(nextD.y, nextD.z, accumColor.x, accumColor.y);
                q[2] = make_float4(accumColor.z, accumAlpha, blocking, density);
            }
            q[3] = make_float4(__uint_as_float(numBounces), __uint_as_float(timeout | (cont ? 0x80000000u : 0u)),
                               __uint_as_float((uint32_t)out_idx), __uint_as_float((uint32_t)(out_idx >> 32)));
        }
    }
    const bool write_px = write && !cont && GRT_IN_PART; // queued rays write their pixel in stage 3
    if (write_px) {
        if (a.outf) {
            a.outf[out_idx * 3] = col.x; a.outf[out_idx * 3 + 1] = col.y; a.outf[out_idx * 3 + 2] = col.z;
        }
        if (a.out8) {
            a.out8[out_idx * 3] = quantize8(col.x);
            a.out8[out_idx * 3 + 1] = quantize8(col.y);
            a.out8[out_idx * 3 + 2] = quantize8(col.z);
        }
    }
#ifdef GRT_TILE_DIAG
    if (COUNT) { c = (lane == 0) ? w : Cnt(); c.fetches = w.fetches; }
#endif
    if (COUNT) {
        uint32_t v0 = c.rays, v1 = c.segments, v2 = c.hit_evals, v3 = c.rounds, v4 = c.node_visits, v5 = c.proxy_tests;
        for (int off = 32; off > 0; off >>= 1) {
            v0 += (uint32_t)__shfl_xor((int)v0, off); v1 += (uint32_t)__shfl_xor((int)v1, off);
            v2 += (uint32_t)__shfl_xor((int)v2, off); v3 += (uint32_t)__shfl_xor((int)v3, off);
            v4 += (uint32_t)__shfl_xor((int)v4, off); v5 += (uint32_t)__shfl_xor((int)v5, off);
        }
        if (lane == 0) {
            if (v0) atomicAdd(&a.counters[0], (unsigned long long)v0);
            if (v1) atomicAdd(&a.counters[1], (unsigned long long)v1);
            if (v2) atomicAdd(&a.counters[2], (unsigned long long)v2);
            if (v3) atomicAdd(&a.counters[3], (unsigned long long)v3);
            if (v4) atomicAdd(&a.counters[4], (unsigned long long)v4);
            if (v5) atomicAdd(&a.counters[5], (unsigned long long)v5);
            // record bytes in 16-B units: particle records once per wave, child boxes (32 B) once per lane that loaded one
            const unsigned long long fb = (unsigned long long)c.fetches + 2ull * v4;
            if (fb) atomicAdd(&a.counters[6], fb);
        }
        if (c.stall_exits) atomicAdd(&a.counters[7], (unsigned long long)c.stall_exits);
    }
    if (!SINGLE) break;
    } // for unit_s
}

#undef GRT_IN_PART
#undef GRT_BAG_CAP
#undef GRT_PRUNE_ROOM
#undef GRT_TILE_CHECK_FRONT
#undef KS
#undef KLAST
#undef KPRESS
#undef KROOM
#undef PL_OTHER
#undef PL_ALPHA
#undef PL_COL

} // namespace

typedef void (*TileKernel)(const RenderArgs);
#ifdef GRT_TILE_QUAD_TU
// ---- this translation unit (grt_render_tile_quad.hip) holds the quad mode alone (MODE 3: the four-way parts of heavy tiles,
//      lanes = rays x slots) ----
int launch_render_tile_quad(const RenderArgs& a, bool count, hipStream_t stream, std::string* err)
{
    if (!a.qparts || !a.qpart_count) return GRT_OK;
    const bool sh = a.p.sh_degree_max > 0;
    RenderArgs b = a;
    b.heavy_role = 0;
    TileKernel k = count ? (sh ? k_render_tile<true, true, false, 3, false> : k_render_tile<true, false, false, 3, false>)
                         : (sh ? k_render_tile<false, true, false, 3, false> : k_render_tile<false, false, false, 3, false>);
    // (one wave per entry of the list when the host knows its length, else per entry it can hold: the waves past its end exit at once)
    hipLaunchKernelGGL(k, dim3(a.quad_known ? a.quad_known - 1u : kQuadListCap), dim3(kWG), 0, stream, b);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (err) *err = std::string("k_render_tile (quad parts) launch: ") + hipGetErrorString(e);
        return GRT_ERR_HIP;
    }
    return GRT_OK;
}
#elif defined(GRT_TILE_SINGLE_TU)
// ---- this translation unit (grt_render_tile_single.hip) holds the one-ray-per-wave mode alone, compiled with an 8-key
//      window: its LDS per wave is 14 KB instead of 19 KB (the window's payload cells carry the events' radiance there) and
//      it fits 168 VGPRs, so 11 waves per CU are resident instead of 8.  The mode waits on memory for 46 % of its wave
//      cycles (profiles/r03_C4_counters.json): C4 3.98 -> 3.82 ms.  The camera-ray and bundle kernels keep 12 keys (with 8
//      they lose 4-10 %). ----
static TileKernel pick_single(bool count, bool sh, bool pieces)
{
#define GRT_PICK2(C, S) (pieces ? k_render_tile<C, S, true, 2, true> : k_render_tile<C, S, true, 2, false>)
    return count ? (sh ? GRT_PICK2(true, true) : GRT_PICK2(true, false)) : (sh ? GRT_PICK2(false, true) : GRT_PICK2(false, false));
#undef GRT_PICK2
}

int launch_render_tile_single(const RenderArgs& a, bool count, hipStream_t stream, std::string* err)
{
    const bool sh = a.p.sh_degree_max > 0;
    RenderArgs b = a;
    b.heavy_role = 0;
    hipLaunchKernelGGL(pick_single(count, sh, a.has_pieces != 0u), dim3(GRT_TILE_GRID2), dim3(kWG), 0, stream, b);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (err) *err = std::string("k_render_tile (one ray per wave) launch: ") + hipGetErrorString(e);
        return GRT_ERR_HIP;
    }
    return GRT_OK;
}
#else
static TileKernel pick_tile(bool count, bool sh, bool mesh, int mode, bool pieces)
{
#define GRT_PICK3(C, S, P)                                                                                 \
    (mode == 1 ? k_render_tile<C, S, true, 1, P> : (mesh ? k_render_tile<C, S, true, 0, P> : k_render_tile<C, S, false, 0, P>))
#define GRT_PICK2(C, S) (pieces ? GRT_PICK3(C, S, true) : GRT_PICK3(C, S, false))
    return count ? (sh ? GRT_PICK2(true, true) : GRT_PICK2(true, false)) : (sh ? GRT_PICK2(false, true) : GRT_PICK2(false, false));
#undef GRT_PICK2
#undef GRT_PICK3
}

// mode 0: camera rays (mesh = stage 2 of the wavefront pipeline: up to their mesh hit); mode 1: stage 3, one wave per
// chunk of a.queue_in (grid = the most chunks there can be: one per 8x8 tile of the launch); mode 2: one wave per ray of
// the heavy list, a resident grid drawing from it (grt_render_tile_single.hip)
int launch_render_tile(const RenderArgs& a, bool count, bool mesh, int mode, hipStream_t stream, std::string* err, const LaunchAux* aux)
{
    if (a.n_blocks == 0) return GRT_OK;
    if (a.root_ref != kNoRoot && (!a.pbox || (!(a.root_ref & kLeafBit) && !a.qnodes) || (mode == 0 && !a.erec))) {
        if (err) *err = "tile kernel: per-child BVH layout or eye records missing";
        return GRT_ERR_INVALID;
    }
    if (mode != 0 && (!a.queue_in || !a.qcount_in || !a.prec || !a.queue || !a.qcount || (!a.heavy && mode == 1) || !a.hcount || !a.fqueue ||
                      !a.fcount || !a.hnext)) {
        if (err) *err = "tile kernel: continuation queues missing";
        return GRT_ERR_INVALID;
    }
    if (mode == 2) return launch_render_tile_single(a, count, stream, err);
    const bool sh = a.p.sh_degree_max > 0;
    RenderArgs b = a;
    b.heavy_role = 0;
    // mode 0: one wave per entry of the launch order (tiles + the parts of split tiles, padded) or per tile; mode 1: one wave per
    // chunk the queue can hold (launch_render passes it as n_launch)
    const uint32_t grid = ((mode == 1 || a.order) && a.n_launch) ? a.n_launch : a.n_blocks * 4u;
    // The four-way parts of heavy tiles run on the quad kernel BESIDE this launch (camera rays without meshes or pieces).  The quad
    // kernel goes on the frame's own stream, this kernel on the second one behind a fork event: the parts are the frame's longest waves
    // and must be dispatched FIRST — launched the other way round (or with the second stream at high priority) they found the machine
    // already full of this kernel's waves and waited 0.2-0.4 ms for registers (a 720p frame: 0.76 -> 0.9-1.0 ms).
    const bool quad = mode == 0 && !mesh && a.quad_parts && a.order && a.n_launch && a.qparts && a.qpart_count && aux && aux->aux && aux->fork && aux->join;
    b.quad_parts = quad ? 1u : 0u;
    hipStream_t main_stream = stream;
    if (quad) {
        hipError_t eq = hipEventRecord(aux->fork, stream);
        if (eq == hipSuccess) eq = hipStreamWaitEvent(aux->aux, aux->fork, 0);
        if (eq != hipSuccess) {
            if (err) *err = std::string("tile kernel: fork to the second stream: ") + hipGetErrorString(eq);
            return GRT_ERR_HIP;
        }
        const int rq = launch_render_tile_quad(b, count, stream, err);
        if (rq != GRT_OK) return rq;
        main_stream = aux->aux;
    }
    hipLaunchKernelGGL(pick_tile(count, sh, mesh || mode != 0, mode, a.has_pieces != 0u), dim3(grid), dim3(kWG), 0, main_stream, b);
    if (quad) {
        hipError_t eq = hipEventRecord(aux->join, aux->aux);
        if (eq == hipSuccess) eq = hipStreamWaitEvent(stream, aux->join, 0);
        if (eq != hipSuccess) {
            if (err) *err = std::string("tile kernel: join of the second stream: ") + hipGetErrorString(eq);
            return GRT_ERR_HIP;
        }
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (err) *err = std::string("k_render_tile launch: ") + hipGetErrorString(e);
        return GRT_ERR_HIP;
    }
    return GRT_OK;
}
#endif

} // namespace grt
