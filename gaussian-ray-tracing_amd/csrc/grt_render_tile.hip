// grt_render_tile.hip — tile kernel: one wave64 = one 8x8 pixel tile, BVH culling done per CHILD BOX (gfx950).
//
// The streaming kernel (grt_render_stream.hip) walks the tree with lanes = rays: every popped 4-wide node costs the
// wave four 64-lane box tests, four wave reductions and scalar push logic (~230 VALU + ~170 SALU per node,
// profiles/r02_isa_budget.json) although the 64 rays of a tile are almost parallel and nearly always agree.  This
// kernel turns that part around:
//
//   * CULLING IS DONE WITH LANES = CHILD BOXES.  The tile's rays share the eye and lie inside a thin frustum (four
//     planes through the eye, from wave reductions over the lanes' directions).  A step takes up to 16 unexpanded
//     nodes off the frontier at once; lane l loads child (l & 3) of node (l >> 2) — two 16-B vector loads — and tests
//     that ONE box against the frustum (conservative; culling only) and computes a lower bound lambda of the hit
//     distance of ANY ray of the tile inside it.  64 boxes per step for ~70 VALU instead of 4 boxes for ~230.
//     A leaf range expands the same way into its (<= 4) particles, whose boxes sit in pbox[].
//   * the frontier (unexpanded subtrees AND untested particles) lives in one (lambda, ref) register pair, slot i =
//     lane i; children are compacted into free slots through a 512-B LDS exchange buffer (rank = v_mbcnt of the
//     ballot).  A full frontier spills to a depth-first stack in LDS that is drained first with the bound unchanged.
//   * F = the smallest lambda on the frontier is the FINALITY bound exactly as in the streaming kernel: no unseen
//     event of any lane can have t < F, so buffered events below F are composited in key order (t, id, entry<exit).
//   * exact work keeps lanes = rays and the streaming kernel's arithmetic, operation for operation: a particle whose
//     lambda reaches the front is fetched by scalar loads (64-B record + 16-B eye record) and slab-tested by all
//     lanes; hits go into the per-lane sorted window (12 keys in registers, payload cells in LDS; the same generated
//     EXEC-masked insert/shift macros), overflow sets the lane's cutoff and costs another pass.  Frames are therefore
//     bit-identical to the other kernels'.
//   * compositing is deferred until enough lanes have a final event (or a window is about to overflow): one
//     compositing step costs the same whether 1 or 64 lanes take part.
//
// Used for camera rays (window / tile modes), with or without the mesh wavefront pipeline (MESH = true: the primary
// segment ends at the per-lane mesh hit and the rays that go on are compacted into the continuation queue).
// Citations (file:line) are into Ray-Studio2/gaussian-ray-tracing.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

#include "grt_device.h"
#include "grt_internal.h"
#include "grt_wave.h"

namespace grt {

namespace {

constexpr int kBlock = 256; // threads of a 16x16 screen block (the unit of RenderArgs::n_blocks)
constexpr int kWG = 64;     // one wave per workgroup, as in the streaming kernel
constexpr uint32_t kBatch = 16u; // nodes expanded per step (x 4 children = 64 lanes)
constexpr uint32_t kBag = 256u;  // far frontier entries parked in LDS (4 per lane when they are rebalanced)
constexpr uint32_t kMaxIters = 1u << 21; // steps of one tile before the watchdog gives up (a heavy C3 tile takes ~2000)
constexpr uint32_t kStack = 288u; // depth-first overflow stack (only when the LDS bag is full too; guarded).  LDS per wave
                                  // must stay <= 10 KB: 10304 B gave 15 waves per CU instead of 16 and cost 4 %
constexpr uint32_t kKeep = 40u;
constexpr uint32_t kOvf = kTileOvfEntries; // per-lane capacity of the window's overflow bag (entries of 16 B, in global memory)  // frontier entries kept in registers by a rebalance (the nearest ones)

#ifndef GRT_TILE_WAVES
#define GRT_TILE_WAVES 4
#endif
#define GRT_KS 12
#define KS 12
#define KLAST k11
#define KPRESS k9 /* a lane holding >= KS-2 keys asks for compositing before the next insert */
#include "grt_slots_gen.inc"
#define PL_OTHER(cell) pl_other[(cell) * kWG + lane]
#define PL_ALPHA(cell) pl_alpha[(cell) * kWG + lane]

// Diagnostic build (make EXTRA=-DGRT_TILE_DIAG, never shipped; counters on): the counters hold WAVE-level trip counts —
// rays: node steps, segments: particles fetched, hit_evals: compositing steps, rounds: passes, node_visits: depth-first
// pops, proxy_tests: exact tests executed, rec_fetches: leaf steps, stall_exits: frontier rebalances.
// Checking build (make EXTRA=-DGRT_TILE_CHECK, never shipped): stall_exits counts violated invariants (an event turning
// up below the front: +1 per lane; frontier entries not conserved by a rebalance: +1000 per lane) and, when a float
// frame is rendered, row 0 of it receives the (t, 2 id + exit, T) log of the events lane GRT_TILE_CHECK_LANE composites.
#ifndef GRT_TILE_CHECK_LANE
#define GRT_TILE_CHECK_LANE 0u
#endif
#ifdef GRT_TILE_DIAG
#define GRT_D(f, n) if (COUNT) w.f += (n);
#elif defined(GRT_MARKS)
#define GRT_D(f, n) asm volatile("; GRT_MARK " #f);
#else
#define GRT_D(f, n)
#endif

// signed-float wave reductions (set-up only: ten of them per tile)
__device__ __forceinline__ float wave_fmin(float v)
{
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_fmax(float v)
{
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
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
__device__ __forceinline__ void wave_fence()
{
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // compiler ordering of the LDS exchange; no instruction
#endif
}

template <bool COUNT, bool SH, bool MESH>
__global__ __launch_bounds__(kWG, GRT_TILE_WAVES) void k_render_tile(const RenderArgs a)
{
    const uint32_t rank = xcd_swizzle(blockIdx.x, gridDim.x, a.swizzle_chunk * 4u);
    __shared__ float pl_other[KS * kWG], pl_alpha[KS * kWG];
    __shared__ uint2 xch[kWG];       // children on their way to free frontier slots
    __shared__ uint32_t xsel[kBatch]; // refs of the nodes picked for this step
    __shared__ uint2 bag[kBag];      // far part of the frontier: (lambda bits, ref), unordered; its minimum is Fbag
    __shared__ uint32_t dstack[kStack]; // depth-first overflow: the batch that overflowed (<= 64) + 3 siblings per level
                                 // below it (<= 3 * 62 for the tree heights the launcher sends here)
    Cnt c, w;
    (void)w;
    const uint32_t unit = a.order ? a.order[rank] : rank;
    // the heaviest tiles of the previous frame (the head of the cost-sorted order) bound the frame: they issue first
    if (a.order && a.tile_prio_div && rank < gridDim.x / a.tile_prio_div) __builtin_amdgcn_s_setprio(2);
    const uint32_t blk = unit >> 2, wave = unit & 3u, lane = threadIdx.x;
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
        const uint32_t tx_ = tile % a.tiles_x, ty_ = tile / a.tiles_x;
        const uint32_t ox = (sub % a.nbx) * 16u + lx, oy = (sub / a.nbx) * 16u + ly;
        px = tx_ * a.tile_w + ox;
        py = ty_ * a.tile_h + oy;
        in_frame = (px < a.p.width) && (py < a.p.height);
        out_idx = ((size_t)j * a.tile_h + oy) * a.tile_w + ox;
    }
    const bool write = in_frame || (a.mode == 1);
    const f3 nU = mk3(-a.p.U[0], -a.p.U[1], -a.p.U[2]), nV = mk3(-a.p.V[0], -a.p.V[1], -a.p.V[2]);
    const f3 W = mk3(a.p.W[0], a.p.W[1], a.p.W[2]);
    const f3 o = mk3(a.p.eye[0], a.p.eye[1], a.p.eye[2]); // wave-uniform origin (camera rays)
    f3 d = mk3(0.0f, 0.0f, -1.0f);
    bool have_ray = in_frame;
    if (in_frame) {
        if (!a.p.mode_fisheye) get_ray(px, py, nU, nV, W, a.p.width, a.p.height, d);
        else have_ray = get_fisheye_ray(px, py, nU, nV, W, a.p.width, a.p.height, d);
    }
    if (COUNT && have_ray && !MESH) c.rays++;
    have_ray = have_ray && (length3(d) > 0.1f) && (a.p.max_bounces > 0u); // loop guard, shaders/tracer.cu:59
    float seg_tmax = a.p.t_max;
    uint32_t pflags = 0;
    f3 nextO = mk3(0, 0, 0), nextD = mk3(0, 0, 0), hitN = mk3(0, 0, 0);
    if (MESH) { // stage 1 (k_primary_mesh) already traced the mesh for this pixel
        const size_t pi = ((size_t)blk * kBlock + wave * 64u + lane) * 3;
        const float4 pr0 = a.prec[pi], pr1 = a.prec[pi + 1], pr2 = a.prec[pi + 2];
        seg_tmax = pr0.x;
        pflags = __float_as_uint(pr0.y);
        nextO = mk3(pr0.z, pr0.w, pr1.x);
        nextD = mk3(pr1.y, pr1.z, pr1.w);
        hitN = mk3(pr2.x, pr2.y, pr2.z);
        have_ray = have_ray && ((pflags >> 16) & 1u);
    }

    // ---- trace() for the whole wave (shaders/tracer.cuh:328-373), density starts at 0 ----
    const float minT = a.p.minTransmittance;
    float T = 1.0f;
    f3 radiance = mk3(0.0f, 0.0f, 0.0f);
    if (COUNT && have_ray) c.segments++;
    const uint64_t raym = wave_ballot(have_ray);
    if (a.root_ref != kNoRoot && raym) {
        const float epsT = 1e-9f;
        const f3 dn = normalize3(d);
        const float t_hi = seg_tmax + epsT; // per lane when MESH (segment ends at the mesh hit)
        const float t_hi_m = __uint_as_float(__float_as_uint(t_hi) - 1u); // largest float below t_hi (t_hi > 0)

        // ---- the tile's frustum (wave-uniform; culling only) ----
        // axis = direction of the first lane that has a ray; (u, v) complete it; a lane's direction is
        // d ~ ax + tu u + tv v, and the four planes bound (tu, tv) over the lanes, widened by 1e-4 rad.
        const int l0 = (int)__builtin_ctzll(raym);
        const f3 ax = mk3(__shfl(d.x, l0), __shfl(d.y, l0), __shfl(d.z, l0));
        f3 e_;
        {
            const float axx = fabsf(ax.x), ayy = fabsf(ax.y), azz = fabsf(ax.z);
            e_ = (axx <= ayy && axx <= azz) ? mk3(1, 0, 0) : ((ayy <= azz) ? mk3(0, 1, 0) : mk3(0, 0, 1));
        }
        const f3 uu = normalize3(cross3(ax, e_)), vv = cross3(ax, uu);
        // The frustum bounds the lanes that still WANT something (GRT_FRUSTUM(mask)): all rays at first; re-fitted when
        // half of them have finished (saturated, or past their window cut-off), so that a few straggling rays do not
        // drag the whole tile's frustum through the rest of the scene.
        float pLx, pLy, pLz, pRx, pRy, pRz, pBx, pBy, pBz, pTx, pTy, pTz;
        float ivx, ivy, ivz; // per-axis slab bound: when every ray moves the same way along an axis,
                             // t >= (near plane - eye) / (largest |d|); 0 when the directions straddle the axis
        bool shx, shy, shz;  // near plane is the box's hi side
#define GRT_AXIS(M, C, IV, SH_)                                                                            \
        {                                                                                                  \
            const float mn_ = uni(wave_fmin((M) ? d.C : INFINITY)), mx_ = uni(wave_fmax((M) ? d.C : -INFINITY)); \
            SH_ = mx_ < -1e-20f;                                                                           \
            IV = (mn_ > 1e-20f) ? (1.0f - 1e-6f) / mx_ : (SH_ ? (1.0f - 1e-6f) / mn_ : 0.0f);              \
            IV = uni(pk_ * IV);                                                                            \
        }
#define GRT_FRUSTUM(M)                                                                                     \
        {                                                                                                  \
            const float da = dot3(d, ax);                                                                  \
            const float ida = 1.0f / fmaxf(da, 1e-6f);                                                     \
            const float tu = dot3(d, uu) * ida, tv = dot3(d, vv) * ida;                                    \
            /* a tile wider than ~75 degrees (tiny fisheye frames) gets no culling at all: every box passes */ \
            const float pk_ = (uni(wave_fmin((M) ? da : 1.0f)) >= 0.25f) ? 1.0f : 0.0f;                    \
            float tu0 = uni(wave_fmin((M) ? tu : INFINITY)), tu1 = uni(wave_fmax((M) ? tu : -INFINITY));   \
            float tv0 = uni(wave_fmin((M) ? tv : INFINITY)), tv1 = uni(wave_fmax((M) ? tv : -INFINITY));   \
            tu0 -= 1e-4f * (1.0f + fabsf(tu0)); tu1 += 1e-4f * (1.0f + fabsf(tu1));                        \
            tv0 -= 1e-4f * (1.0f + fabsf(tv0)); tv1 += 1e-4f * (1.0f + fabsf(tv1));                        \
            pLx = uni(pk_ * (uu.x - tu0 * ax.x)); pLy = uni(pk_ * (uu.y - tu0 * ax.y)); pLz = uni(pk_ * (uu.z - tu0 * ax.z)); \
            pRx = uni(pk_ * (tu1 * ax.x - uu.x)); pRy = uni(pk_ * (tu1 * ax.y - uu.y)); pRz = uni(pk_ * (tu1 * ax.z - uu.z)); \
            pBx = uni(pk_ * (vv.x - tv0 * ax.x)); pBy = uni(pk_ * (vv.y - tv0 * ax.y)); pBz = uni(pk_ * (vv.z - tv0 * ax.z)); \
            pTx = uni(pk_ * (tv1 * ax.x - vv.x)); pTy = uni(pk_ * (tv1 * ax.y - vv.y)); pTz = uni(pk_ * (tv1 * ax.z - vv.z)); \
            GRT_AXIS(M, x, ivx, shx)                                                                       \
            GRT_AXIS(M, y, ivy, shy)                                                                       \
            GRT_AXIS(M, z, ivz, shz)                                                                       \
        }

        uint64_t last_key = mk_skey(a.p.t_min + epsT, 0x03FFFFFFu, 1) | kCellMask; // last composited event (exclusive bound)
        bool alive = have_ray && (T > minT);
        uint32_t stalls = 0;
        uint64_t k0 = kKeyInvalid, k1 = kKeyInvalid, k2 = kKeyInvalid, k3 = kKeyInvalid, k4 = kKeyInvalid,
                 k5 = kKeyInvalid, k6 = kKeyInvalid, k7 = kKeyInvalid, k8 = kKeyInvalid, k9 = kKeyInvalid,
                 k10 = kKeyInvalid, k11 = kKeyInvalid;
        uint32_t pmask = 0; // payload cells in use
        uint32_t iters = 0; // wave-uniform work measure for the scheduling feedback
        bool watchdog = false;
#ifdef GRT_TILE_CHECK
        uint32_t dbg_n = 0, dbg_m = 0;
#endif
        uint32_t chunk = kNoRoot; // this tile's chunk of the overflow pool (taken at the first window overflow)
        const uint32_t ready_min = a.tile_ready_min; // lanes with a final event before a compositing sweep starts

        while (wave_any(alive)) { // one iteration = one front-to-back pass
            if (COUNT && alive) c.rounds++;
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
            k0 = k1 = k2 = k3 = k4 = k5 = k6 = k7 = k8 = k9 = k10 = k11 = kKeyInvalid;
            pmask = 0;
            // wave-level interval of interest: nothing beyond LIM, nothing that ends before LO (stale values are
            // conservative: LIM only shrinks, LO only grows)
            float LIM = uni(wave_fmax(alive ? t_hi_m : 0.0f));
            const float LO = uni(wave_fmin(alive ? t_lo : INFINITY));
            bool lim_dirty = false;
            GRT_FRUSTUM(alive)
            uint32_t nact_ref = (uint32_t)__popcll(wave_ballot(alive)); // wanting lanes the frustum was fitted to
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
                        if ((rebal || !(Ff < INFINITY) || ((Fbag <= Ff + Ff * a.tile_look) && (nocc_ + 8u <= kKeep))) &&
                            (nocc_ + nbag <= kBag)) {
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
                                for (int it = 0; it < 6; it++) {
                                    const float mid = 0.5f * (lo_ + hi_);
                                    const uint32_t n_ = (uint32_t)__popcll(wave_ballot(fl <= mid)) + (uint32_t)__popcll(wave_ballot(bl0 <= mid)) +
                                                        (uint32_t)__popcll(wave_ballot(bl1 <= mid)) + (uint32_t)__popcll(wave_ballot(bl2 <= mid)) +
                                                        (uint32_t)__popcll(wave_ballot(bl3 <= mid));
                                    const bool few = n_ <= kKeep;
                                    lo_ = few ? mid : lo_;
                                    hi_ = few ? hi_ : mid;
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
                        }
                    }
                    F = fminf(Ff, Fbag);
                    // lanes that still want something in this pass: alive and not yet past their window cut-off
                    const float ct_ = (lost != kKeyInvalid) ? key_t(lost) : t_hi_m;
                    const bool act = alive && (ct_ >= F);
                    const uint32_t nact = (uint32_t)__popcll(wave_ballot(act));
                    if (nact == 0u) F = INFINITY; // nothing left to find: the pass is over
                    done = !(F < INFINITY);
                    if (!done && (nact * 2u <= nact_ref)) { // half of them have finished: re-fit the frustum
                        GRT_FRUSTUM(act)
                        LIM = uni(wave_fmax(act ? ct_ : 0.0f));
                        lim_dirty = false;
                        nact_ref = nact;
                    }
                }

                // ---- composite buffered events with t < F (and key < cutoff), in key order; deferred until
                //      ready_min lanes have one, a window is nearly full, or the pass is over ----
                if (!dfs) {
                    bool sweep = done;
                    while (true) {
                        const bool can_ = alive && (k0 != kKeyInvalid) && (key_t(k0) < F) && (k0 < bagmin) && (k0 < lost);
                        const uint64_t cm_ = wave_ballot(can_);
                        // a lane whose next final event sits in its bag needs a refill before it can go on
                        const bool need = bags && alive && (nb != 0u) && !can_ && (key_t(bagmin) < F) && (bagmin < lost) &&
                                          ((k0 == kKeyInvalid) || (k0 >= bagmin));
                        const uint64_t nm_ = bags ? wave_ballot(need) : 0ull;
                        if (!(cm_ | nm_)) break;
                        if (!sweep) {
                            sweep = ((uint32_t)__popcll(cm_ | nm_) >= ready_min) || wave_any(can_ && (KPRESS != kKeyInvalid));
                            if (!sweep) break;
                        }
                        if (!cm_) {
                            // ---- refill: one scan of the bags of the lanes in need; entry by entry, whatever is smaller
                            //      than the window's last key goes in (sorted insert) and the displaced last key takes
                            //      its place in the bag (compacted in place: position w <= i) ----
                            GRT_D(node_visits, 1)
                            // lanes that do not need it yet but have room for four more keys come along: one scan instead
                            // of one per lane a few steps apart
                            const bool rf = need || (alive && (nb != 0u) && (k8 == kKeyInvalid) && (bagmin < lost));
                            uint32_t nmax = rf ? nb : 0u;
                            for (int off = 32; off > 0; off >>= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, off));
                            nmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmax);
                            uint32_t w_ = 0;
                            uint64_t newmin = kKeyInvalid;
                            float4* bp = a.ovf_pool + (size_t)chunk * (kOvf * 64u) + lane;
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
                            continue;
                        }
                        GRT_D(hit_evals, 1)
                        const uint64_t ek = k0;
                        const uint32_t cell = (uint32_t)(ek & kCellMask);
                        const uint32_t id = skey_id(ek);
                        float ea = 0.0f, eo = INFINITY;
                        float4 cc = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (can_) { // payload from LDS and (degree 0) the colour, both in flight while the window is popped
                            ea = PL_ALPHA(cell); eo = PL_OTHER(cell);
                            if (!SH) cc = a.color0[id];
                        }
                        SLOT_SHIFT_ALL(cm_)
#ifdef GRT_TILE_CHECK
                        if (can_ && a.outf && lane == GRT_TILE_CHECK_LANE && dbg_n < 1900u) { // event log of one lane
                            a.outf[dbg_n * 3] = key_t(ek); a.outf[dbg_n * 3 + 1] = (float)(id * 2u + ((((uint32_t)ek) >> 5) & 1u)); a.outf[dbg_n * 3 + 2] = T;
                            dbg_n++;
                        }
#endif
                        if (can_) { // shaders/tracer.cuh:352-367
                            if (COUNT) c.hit_evals++;
                            last_key = ek | kCellMask; // nothing with the same (t, id, exit) can compare above it
                            if (a.p.alpha_min < ea) {
                                f3 L;
                                if (!SH) {
                                    L = mk3(cc.x, cc.y, cc.z);
                                } else {
                                    f3 dl = dn; // keep the SH basis out of loop-invariant hoisting (it would spill)
                                    asm volatile("" : "+v"(dl.x), "+v"(dl.y), "+v"(dl.z));
                                    L = sh_radiance(a.sh + (size_t)id * 48, dl, a.p.sh_degree_max);
                                }
                                radiance = add3(radiance, mul3s(mul3s(L, T), ea));
                                T *= (1.0f - ea);
                            }
                            if (!(T > minT)) alive = false;
                        }
                        // an entry whose exit lies inside the segment is re-keyed to its exit event and keeps its
                        // payload cell, otherwise the cell is released
                        const bool rekey = can_ && ((((uint32_t)ek) & 32u) == 0u) && (eo < t_hi);
                        const uint64_t nk = rekey ? (mk_skey(eo, id, 1) | (uint64_t)cell) : kKeyInvalid;
                        pmask = (can_ && !rekey) ? (pmask & ~(1u << cell)) : pmask;
                        if (wave_any(rekey)) { // wave-uniform branch
                            if (rekey) PL_OTHER(cell) = INFINITY;
                            SLOT_INSERT(nk) // a slot was just freed: it fits
                        }
                    }
                }
                if (done) break;
                if (++iters > kMaxIters) { // watchdog: never reached by design; a counted, visible failure beats a hung GPU
                    c.stall_exits += alive ? 1u : 0u;
                    watchdog = true;
                    break;
                }

                // ---- one step: the entries at the front, four lanes each.  LEAF step: leaf ranges -> their particles'
                //      boxes are culled here and the survivors slab-tested at once (lanes = rays).  NODE step: internal
                //      nodes -> their children's boxes are culled and the survivors join the frontier ----
                const bool occ_l = fr != kNoRoot;
                const bool rng_l = occ_l && ((fr & kLeafBit) != 0u);
                bool leaf_step;
                uint32_t nref; // the entry this lane's group expands
                bool gv;       // group valid
                uint32_t g, j; // group of this lane and its child slot in the group: 4 lanes per leaf range, kTileWide per node
                if (cur != kNoRoot) { // depth-first mode: one entry
                    leaf_step = (cur & kLeafBit) != 0u;
                    g = leaf_step ? (lane >> 2) : (lane / kTileWide);
                    j = leaf_step ? (lane & 3u) : (lane % kTileWide);
                    nref = cur;
                    gv = g == 0u;
                } else {
                    // (no reductions here: the frontier minimum Ff of the loop top and two votes decide the step)
                    const bool have_rng = wave_any(rng_l);
                    // a nearly full frontier takes leaf steps whatever lies in front (testing particles early is always
                    // legal; spilling children to the depth-first stack stalls the front)
                    const uint32_t nocc = (uint32_t)__popcll(wave_ballot(occ_l));
                    const bool crowded = (nocc > 64u - a.tile_reserve) && have_rng;
                    // nodes within the look-ahead of the FRONT are expanded first, so that leaf steps find full batches;
                    // then the nearest ranges (within a band behind the nearest one) are tested together
                    const float hz = F + F * a.tile_look;
                    const bool node_near = wave_any(occ_l && !rng_l && (fl <= hz));
                    leaf_step = have_rng && (!node_near || crowded);
                    // the nearest range / node: the frontier minimum when it is of that kind (the common case), else one
                    // reduction
                    float Fr = Ff_cur, Fn = Ff_cur;
                    if (leaf_step && !wave_any(rng_l && (fl <= Ff_cur))) Fr = wave_min(rng_l ? fl : INFINITY);
                    if (!leaf_step && !node_near) Fn = wave_min(rng_l ? INFINITY : fl);
                    const float tau = leaf_step ? (Fr + Fr * a.tile_band) : fmaxf(hz, Fn);
                    const bool cand = occ_l && (rng_l == leaf_step);
                    // a node step frees one slot per node and may need four: expand only what is sure to fit (at least
                    // one node: a frontier full of internal nodes overflows to the depth-first stack)
                    const uint32_t maxb = leaf_step ? kBatch : max(min(64u / kTileWide, (64u - nocc) / (kTileWide - 1u)), 1u);
                    g = leaf_step ? (lane >> 2) : (lane / kTileWide);
                    j = leaf_step ? (lane & 3u) : (lane % kTileWide);
                    float th = tau;
                    uint64_t sm = wave_ballot(cand && (fl <= th));
                    if ((uint32_t)__popcll(sm) > maxb) {
                        // more candidates than the step can take: the NEAREST ones go first (four bisection steps on
                        // the distance threshold; lane order only breaks what is left of the tie)
                        float lo_ = wave_min(cand ? fl : INFINITY), hi_ = tau; // the nearest candidate itself always qualifies
                        for (int it = 0; it < 4; it++) {
                            const float mid = 0.5f * (lo_ + hi_);
                            const bool few = (uint32_t)__popcll(wave_ballot(cand && (fl <= mid))) <= maxb;
                            lo_ = few ? mid : lo_;
                            hi_ = few ? hi_ : mid;
                        }
                        th = lo_;
                        sm = wave_ballot(cand && (fl <= th));
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
                    gv = g < cnt;
                }
                const uint32_t first = leaf_first(nref);
                const bool cv = gv && (!leaf_step || (j < leaf_count(nref)));
                float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
                if (cv) {
                    const float4* src = leaf_step ? (a.pbox + (size_t)(first + j) * 2) : (a.qnodes + (size_t)nref * (2u * kTileWide) + j * 2u);
                    b0 = src[0];
                    b1 = src[1];
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
                const float lx_ = b0.x - o.x, ly_ = b0.y - o.y, lz_ = b0.z - o.z;
                const float hx_ = b1.x - o.x, hy_ = b1.y - o.y, hz_ = b1.z - o.z;
                // four frustum planes, each at the box corner farthest along its normal (wave-uniform choice); the
                // slack covers the rounding of the three products (|n| < 2): 1e-5 x the L1 size of the box about the eye
                const float epsM = -1e-5f * (((fabsf(lx_) + fabsf(hx_)) + (fabsf(ly_) + fabsf(hy_))) + (fabsf(lz_) + fabsf(hz_)));
#define GRT_PSIDE(P)                                                                                       \
                (__builtin_fmaf(P##x, (P##x >= 0.0f) ? hx_ : lx_,                                          \
                 __builtin_fmaf(P##y, (P##y >= 0.0f) ? hy_ : ly_, P##z * ((P##z >= 0.0f) ? hz_ : lz_))) >= epsM)
                const bool inside = GRT_PSIDE(pL) && GRT_PSIDE(pR) && GRT_PSIDE(pB) && GRT_PSIDE(pT);
#undef GRT_PSIDE
                // lower bound of t over the tile: Euclidean distance to the box, and the per-axis slab bound
                const float ex_ = fmaxf(fmaxf(lx_, -hx_), 0.0f), ey_ = fmaxf(fmaxf(ly_, -hy_), 0.0f),
                            ez_ = fmaxf(fmaxf(lz_, -hz_), 0.0f);
                const float euc = sqrtf(__builtin_fmaf(ex_, ex_, __builtin_fmaf(ey_, ey_, ez_ * ez_)));
                const float sx_ = (shx ? hx_ : lx_) * ivx, sy_ = (shy ? hy_ : ly_) * ivy, sz_ = (shz ? hz_ : lz_) * ivz;
                float lam = fmaxf(fmaxf(euc, sx_), fmaxf(sy_, sz_)) * (1.0f - 2e-6f);
                lam = fmaxf(lam, F); // never below the current front (keeps the frontier monotone)
                bool want = valid && inside && (lam <= LIM);
                if (LO > 0.0f) { // later passes: skip what ends before the restart point
                    const float fx_ = fmaxf(fabsf(lx_), fabsf(hx_)), fy_ = fmaxf(fabsf(ly_), fabsf(hy_)),
                                fz_ = fmaxf(fabsf(lz_), fabsf(hz_));
                    const float far = sqrtf(__builtin_fmaf(fx_, fx_, __builtin_fmaf(fy_, fy_, fz_ * fz_))) * (1.0f + 2e-6f);
                    want = want && (far >= LO);
                }
                uint64_t wm = wave_ballot(want);

                if (leaf_step) {
                    GRT_D(fetches, 1)
                    // ---- exact tests of the surviving particles, all lanes = rays (grt_render_stream's arithmetic) ----
                    while (wm) {
                        const uint32_t b = (uint32_t)__builtin_ctzll(wm);
                        wm &= wm - 1ull;
                        const uint32_t pidx = (uint32_t)__builtin_amdgcn_readlane((int)cref, (int)b);
                        GRT_D(segments, 1)
                        const uint32_t ridx = pidx * 4u;
                        const float4 r0 = sload4(a.rec, ridx), r1 = sload4(a.rec, ridx + 1), r2 = sload4(a.rec, ridx + 2),
                                     r3 = sload4(a.rec, ridx + 3);
                        const float4 e0 = sload4(a.erec, ridx), e1 = sload4(a.erec, ridx + 1), e2 = sload4(a.erec, ridx + 2),
                                     e3 = sload4(a.erec, ridx + 3);
                        if (COUNT) c.fetches += 8; // wave-uniform: 64-B record + 64-B eye record, in 16-B units
                        const f3 mu = mk3(r0.x, r0.y, r0.z);
                        m33 A;
                        A.a[0] = r1.x; A.a[1] = r1.y; A.a[2] = r1.z;
                        A.a[3] = r2.x; A.a[4] = r2.y; A.a[5] = r2.z;
                        A.a[6] = r3.x; A.a[7] = r3.y; A.a[8] = r3.z;
                        const f3 o_g = mk3(e0.x, e0.y, e0.z); // A (o - mu), from the eye record (wave-uniform)
                        const f3 d_g = matvec(A, d);
                        {   // conservative sphere pre-test (proxy_sphere_maybe_pre) as lane masks
                            const float b_ = dot3(o_g, d_g), aa_ = dot3(d_g, d_g);
                            const uint64_t m_ = (wave_ballot(e0.w <= 0.0f) | wave_ballot(b_ * b_ * (1.0f + 4e-6f) >= aa_ * e0.w)) &
                                                wave_ballot(alive);
                            if (!m_) continue;
                        }
                        if (COUNT && alive) c.proxy_tests++;
                        GRT_D(proxy_tests, 1)
                        float te, tx;
                        const float pa[10] = {e1.x, e1.y, e1.z, e1.w, e2.x, e2.y, e2.z, e2.w, e3.x, e3.y}; // slab_project(o_g)
                        const bool hit = proxy_slabs_pre(pa, d_g, r0.w, te, tx) && alive;
                        const uint32_t id = __float_as_uint(r2.w);
                        const uint64_t ke = mk_skey(te, id, 0), kx = mk_skey(tx, id, 1);
                        // float compares first: te/tx may be negative or NaN, the unsigned key compares assume t > 0
                        const bool in_e = hit && (te >= t_lo) && (te < t_hi) && (ke > pass_lo);
                        const bool in_x = hit && (tx >= t_lo) && (tx < t_hi) && (kx > pass_lo);
                        const uint64_t k_first = in_e ? ke : (in_x ? kx : kKeyInvalid); // the slot's first pending event
                        const bool ins = (k_first != kKeyInvalid) && (k_first < lost);
#ifdef GRT_TILE_CHECK
                        if (ins && key_t(k_first) < F) c.stall_exits++; // finality violated: an event below the front turned up late
#endif
                        if (wave_any(ins)) { // wave-uniform branch
                            // alpha does not depend on the hit distance (shaders/tracer.cuh:354-357)
                            const float alpha = fminf(0.99f, response_from(A, mu, o, d, o_g, d_g) * r1.w);
                            const float other = (in_e && in_x) ? tx : INFINITY;
                            // window full: the largest pending key is dropped (the new one or the last slot's) and
                            // the lane becomes lossy beyond it
                            const bool full = KLAST != kKeyInvalid;
                            const bool take = ins && (!full || k_first < KLAST);
                            const bool drop = ins && full; // the farthest of (window + new particle) leaves the window
                            const uint32_t cell = full ? (uint32_t)(KLAST & kCellMask) : (uint32_t)__builtin_ctz(~pmask);
                            if (wave_any(drop)) { // wave-uniform branch: into the lane's bag
                                if (chunk == kNoRoot) { // first overflow of this tile: take a chunk of the pool
                                    uint32_t ch = 0;
                                    if (lane == 0u) ch = atomicAdd(a.ovf_next, 1u);
                                    ch = (uint32_t)__builtin_amdgcn_readfirstlane((int)ch);
                                    chunk = (ch < a.ovf_chunks) ? ch : (kNoRoot - 1u); // pool exhausted: drop for good
                                }
                                const uint64_t dk = take ? (KLAST | kCellMask) : (k_first | kCellMask);
                                const bool to_bag = drop && (chunk < a.ovf_chunks) && (nb < kOvf);
                                if (to_bag) {
                                    const float d_o = take ? PL_OTHER(cell) : other, d_a = take ? PL_ALPHA(cell) : alpha;
                                    a.ovf_pool[((size_t)chunk * kOvf + nb) * 64u + lane] =
                                        make_float4(__uint_as_float((uint32_t)dk), __uint_as_float((uint32_t)(dk >> 32)), d_o, d_a);
                                    nb++;
                                    bagmin = (dk < bagmin) ? dk : bagmin;
                                }
                                const bool gone = drop && !to_bag;
                                lost = (gone && (dk < lost)) ? dk : lost;
                                bags = true;
                                if (wave_any(gone)) lim_dirty = true;
                            }
                            KLAST = (take && full) ? kKeyInvalid : KLAST;
                            pmask = take ? (pmask | (1u << cell)) : pmask;
                            if (take) { PL_OTHER(cell) = other; PL_ALPHA(cell) = alpha; }
                            SLOT_INSERT(take ? (k_first | (uint64_t)cell) : kKeyInvalid)
                        }
                    }
                    continue;
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
                            watchdog = true;
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
            const bool progressed = last_key != pass_lo;
            stalls = progressed ? 0u : stalls + 1u;
            const bool again = alive && (lost != kKeyInvalid);
            if (COUNT && again && stalls >= 2u) c.stall_exits++;
            alive = again && (stalls < 2u) && !watchdog;
        }
        if (a.cost && lane == 0) atomicMax(&a.cost[unit], iters);
    }
    const float density = 1.0f - T;

    f3 col = mk3(0.0f, 0.0f, 0.0f);
    bool cont = false; // MESH: the ray goes on bouncing (stage 3)
    f3 accumColor = mk3(0, 0, 0);
    float accumAlpha = 0.0f, blocking = 0.0f;
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
                cont = (length3(nextD) > 0.1f) && (numBounces < a.p.max_bounces);
            }
            col = accumColor;
        }
    }
    if (MESH) {
        // ---- compaction of the rays that go on: wave ballot + popcount prefix + ONE atomic per wave ----
        const uint64_t mask = wave_ballot(cont);
        if (mask) { // wave-uniform
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(a.qcount, (uint32_t)__popcll(mask));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (cont) {
                const uint32_t slot = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                float4* q = a.queue + (size_t)slot * 4;
                q[0] = make_float4(nextO.x, nextO.y, nextO.z, nextD.x);
                q[1] = make_float4(nextD.y, nextD.z, accumColor.x, accumColor.y);
                q[2] = make_float4(accumColor.z, accumAlpha, blocking, density);
                q[3] = make_float4(__uint_as_float(numBounces), __uint_as_float(1u), // timeout after one iteration
                                   __uint_as_float((uint32_t)out_idx), __uint_as_float((uint32_t)(out_idx >> 32)));
            }
        }
    }
    const bool write_px = write && !cont; // queued rays write their pixel in stage 3
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
}

#undef KS
#undef KLAST
#undef KPRESS
#undef PL_OTHER
#undef PL_ALPHA

} // namespace

typedef void (*TileKernel)(const RenderArgs);
static TileKernel pick_tile(bool count, bool sh, bool mesh)
{
#define GRT_PICK(K)                                                                                        \
    (count ? (sh ? (mesh ? K<true, true, true> : K<true, true, false>) : (mesh ? K<true, false, true> : K<true, false, false>)) \
           : (sh ? (mesh ? K<false, true, true> : K<false, true, false>) : (mesh ? K<false, false, true> : K<false, false, false>)))
    return GRT_PICK(k_render_tile);
#undef GRT_PICK
}

int launch_render_tile(const RenderArgs& a, bool count, bool mesh, hipStream_t stream, std::string* err)
{
    if (a.n_blocks == 0) return GRT_OK;
    if (a.root_ref != kNoRoot && (!a.pbox || (!(a.root_ref & kLeafBit) && !a.qnodes) || !a.erec)) {
        if (err) *err = "tile kernel: per-child BVH layout or eye records missing";
        return GRT_ERR_INVALID;
    }
    const bool sh = a.p.sh_degree_max > 0;
    RenderArgs b = a;
    b.heavy_role = 0;
    hipLaunchKernelGGL(pick_tile(count, sh, mesh), dim3(a.n_blocks * 4u), dim3(kWG), 0, stream, b);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (err) *err = std::string("k_render_tile launch: ") + hipGetErrorString(e);
        return GRT_ERR_HIP;
    }
    return GRT_OK;
}

} // namespace grt
