// grt_render_stream.hip — single-pass, wave-cooperative "streaming" render kernel (gfx950).
//
// Same contract and per-lane arithmetic as grt_render_wave.hip (one wave64 = one 8x8 pixel tile, scalar
// record fetches, bit-identical results), but the k = 7 re-traversal rounds of trace()
// (shaders/tracer.cuh:341-369) are replaced by ONE front-to-back pass:
//   * the wave expands the LBVH BEST-FIRST: the frontier (unexpanded subtrees) lives in wave registers —
//     slot i is lane i of (lambda, ref) VGPR pairs, 128 slots — keyed by lambda = the smallest box-entry
//     distance over the lanes that want the subtree; pop = DPP min-reduction + ballot + v_readlane;
//   * when a subtree with key lambda is popped, no unseen hit of any lane can have t < lambda, so every
//     buffered hit event with t < lambda is FINAL and is composited immediately, in key order
//     (t, particle id, entry<exit) — exactly the order the rounds would have produced;
//   * each lane buffers PARTICLES, not hits: a slot holds (current key, other t, alpha); compositing an entry
//     event re-keys the slot to its exit event.  7 slots cover a window of ~14 hits;
//   * when the frontier is full, children go to a depth-first wave-register stack instead (popped before
//     any frontier entry, finality bound unchanged), so the frontier can never overflow;
//   * a lane whose window overflows records the smallest key it had to drop (`cutoff`), keeps compositing
//     below it, and — only if it still has transmittance left at the end of the pass — starts another
//     pass from its last composited key (the reference's "next round", now the exception, not the rule);
//   * lanes stop wanting subtrees once T <= minTransmittance, so the pass ends early for opaque tiles.
//
// Style note: the per-lane slot file and the wave-level frontier are plain local scalars driven by macros
// (no structs passed by reference, no loop-indexed arrays): that is what keeps all of it in registers —
// struct/array forms of the same code were left in scratch memory by the compiler.
#include <hip/hip_runtime.h>

#include <string>

#include "grt_device.h"
#include "grt_internal.h"

namespace grt {

namespace {

constexpr int kBlock = 256;
#ifndef GRT_STREAM_WAVES
#define GRT_STREAM_WAVES 4 // waves per SIMD the register allocator must fit (128 VGPRs; a handful spill to scratch)
#endif

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

// wave64 min of non-negative floats -> wave-uniform value (4 DPP steps inside rows of 16, then 4 readlanes)
__device__ __forceinline__ float wave_min(float v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int x = __float_as_int(v);
    x = __float_as_int(fminf(__int_as_float(x), __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false))));  // quad_perm [1,0,3,2]
    x = __float_as_int(fminf(__int_as_float(x), __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false))));  // quad_perm [2,3,0,1]
    x = __float_as_int(fminf(__int_as_float(x), __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false)))); // row_half_mirror
    x = __float_as_int(fminf(__int_as_float(x), __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false)))); // row_mirror
    const float a = __int_as_float(__builtin_amdgcn_readlane(x, 0)), b = __int_as_float(__builtin_amdgcn_readlane(x, 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(x, 32)), d = __int_as_float(__builtin_amdgcn_readlane(x, 48));
    return fminf(fminf(a, b), fminf(c, d));
#else
    return v;
#endif
}

// ---- per-lane particle slots.  Registers hold only the 12 sorted 64-bit keys k0..k11 (~0 = free); each key
// carries, in its 4 lowest bits, the index of its payload cell in LDS (exit t while the key is the entry event —
// +inf = no exit event — and alpha), so sorting moves 2 registers per slot and the window is 12 particles
// (~24 hits) for fewer registers than 7 register-resident (key, other, alpha) slots.
//   key = t bits << 32 | particle id << 5 | exit << 4 | payload cell          (particle ids < 2^27)
constexpr int KS = 12;
constexpr uint64_t kCellMask = 15ull;
__device__ __forceinline__ uint64_t mk_skey(float t, uint32_t id, uint32_t is_exit)
{
    return ((uint64_t)__float_as_uint(t) << 32) | (uint64_t)((id << 5) | (is_exit << 4));
}
__device__ __forceinline__ uint32_t skey_id(uint64_t k) { return ((uint32_t)k) >> 5; }
#define SLOT_DECL                                                                                          \
    uint64_t k0 = kKeyInvalid, k1 = kKeyInvalid, k2 = kKeyInvalid, k3 = kKeyInvalid, k4 = kKeyInvalid,   \
             k5 = kKeyInvalid, k6 = kKeyInvalid, k7 = kKeyInvalid, k8 = kKeyInvalid, k9 = kKeyInvalid,   \
             k10 = kKeyInvalid, k11 = kKeyInvalid;                                                         \
    uint32_t pmask = 0; /* payload cells in use */
#define SLOT_CLEAR                                                                                         \
    k0 = k1 = k2 = k3 = k4 = k5 = k6 = k7 = k8 = k9 = k10 = k11 = kKeyInvalid;                            \
    pmask = 0;
#define SLOT_STEP(i)                                                                                       \
    {                                                                                                      \
        const bool lt_ = ik_ < k##i;                                                                       \
        const uint64_t tk_ = k##i;                                                                         \
        k##i = lt_ ? ik_ : tk_;                                                                            \
        ik_ = lt_ ? tk_ : ik_;                                                                             \
    }
// branch-free sorted insert of KEY (with its cell bits); KEY == ~0 is a no-op.  The caller guarantees room.
#define SLOT_INSERT(KEY)                                                                                   \
    {                                                                                                      \
        uint64_t ik_ = (KEY);                                                                              \
        SLOT_STEP(0) SLOT_STEP(1) SLOT_STEP(2) SLOT_STEP(3) SLOT_STEP(4) SLOT_STEP(5) SLOT_STEP(6)         \
        SLOT_STEP(7) SLOT_STEP(8) SLOT_STEP(9) SLOT_STEP(10) SLOT_STEP(11)                                 \
    }
#define SLOT_SHIFT(i, j) { k##i = can_ ? k##j : k##i; }
#define PL_OTHER(cell) pl_other[(cell) * kBlock + threadIdx.x]
#define PL_ALPHA(cell) pl_alpha[(cell) * kBlock + threadIdx.x]

// ---- wave-level frontier: slot i (0..63) = lane i of (fl0, fr0), slot 64+i = lane i of (fl1, fr1);
// fu0/fu1 = wave-uniform occupancy masks ----
#define FRONTIER_PUSH(LAMBDA, REF, OK)                                                                     \
    {                                                                                                      \
        const bool hi_ = (fu0 == ~0ull);                    /* first 64 slots full: use the second bank */ \
        const uint64_t free_ = hi_ ? ~fu1 : ~fu0;                                                          \
        OK = free_ != 0ull;                                                                                \
        const uint32_t slot_ = OK ? (uint32_t)__builtin_ctzll(free_) : 64u; /* 64 matches no lane */       \
        const bool me_ = lane == slot_;                                                                    \
        fl0 = (me_ && !hi_) ? (LAMBDA) : fl0;                                                              \
        fr0 = (me_ && !hi_) ? (REF) : fr0;                                                                 \
        fl1 = (me_ && hi_) ? (LAMBDA) : fl1;                                                               \
        fr1 = (me_ && hi_) ? (REF) : fr1;                                                                  \
        const uint64_t bit_ = OK ? (1ull << (slot_ & 63u)) : 0ull;                                         \
        fu0 |= hi_ ? 0ull : bit_;                                                                          \
        fu1 |= hi_ ? bit_ : 0ull;                                                                          \
    }

// depth-first overflow stack (wave-register stack: entry i = lane i of ds0 / ds1), depth <= 128
#define DFS_PUSH(REF)                                                                                      \
    {                                                                                                      \
        if (dsp < 64u) ds0 = (lane == dsp) ? (REF) : ds0;                                                  \
        else ds1 = (lane == dsp - 64u) ? (REF) : ds1;                                                      \
        ++dsp;                                                                                             \
    }

template <bool COUNT, bool SH, bool MESH>
__global__ __launch_bounds__(kBlock, GRT_STREAM_WAVES) void k_render_stream(const RenderArgs a)
{
    __shared__ float pl_other[KS * kBlock], pl_alpha[KS * kBlock];
    Cnt c;
    const uint32_t blk = [&] { // workgroup -> screen block, XCD-contiguous (speed only)
        return a.order ? a.order[blockIdx.x] : xcd_swizzle(blockIdx.x, a.n_blocks, a.swizzle_chunk);
    }();
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
    // MESH: stage 1 (k_primary_mesh) already traced the mesh for this pixel
    float seg_tmax = a.p.t_max;
    uint32_t pflags = 0;
    f3 nextO = mk3(0, 0, 0), nextD = mk3(0, 0, 0), hitN = mk3(0, 0, 0);
    if (MESH) {
        const size_t pi = ((size_t)blk * kBlock + threadIdx.x) * 3;
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
    if (a.root_ref != kNoRoot) {
        const float epsT = 1e-9f;
        const f3 dn = normalize3(d);
        const rayinv ri = mk_rayinv(o, d);
        const float t_hi = seg_tmax + epsT; // per lane when MESH (segment ends at the mesh hit)
        uint64_t last_key = mk_skey(a.p.t_min + epsT, 0x07FFFFFFu, 1) | kCellMask; // last composited event (exclusive bound)
        bool alive = have_ray && (T > minT);
        uint32_t stalls = 0;
        SLOT_DECL
        uint32_t iters = 0; // wave-uniform work measure for the scheduling feedback
        while (__any(alive)) { // one iteration = one front-to-back pass
            if (COUNT && alive) c.rounds++;
            const uint64_t pass_lo = last_key; // events with key <= pass_lo were composited by an earlier pass
            const float t_lo = key_t(pass_lo);
            uint64_t cutoff = kKeyInvalid;     // smallest key this lane had to drop in this pass
            SLOT_CLEAR
            float fl0 = INFINITY, fl1 = INFINITY;
            uint32_t fr0 = 0, fr1 = 0, ds0 = 0, ds1 = 0, dsp = 0;
            uint64_t fu0 = 0, fu1 = 0;
            {
                bool ok_;
                FRONTIER_PUSH(0.0f, a.root_ref, ok_)
                (void)ok_;
            }
            float F = 0.0f;
            bool final_sweep = false;
            uint32_t bypass = kNoRoot; // a child whose key equals F is a frontier minimum: expand it next, no push/pop
            while (true) {
                uint32_t cur = 0;
                if (!final_sweep) {
                    if (bypass != kNoRoot && __any(alive)) {
                        cur = bypass;
                        bypass = kNoRoot;
                    } else if (!__any(alive) || (dsp == 0 && !(fu0 | fu1))) {
                        // frontier exhausted (or every lane done): everything still buffered is final
                        F = INFINITY;
                        final_sweep = true;
                    } else if (dsp) { // depth-first overflow entries first; the finality bound F is unchanged
                        --dsp;
                        cur = dsp < 64u ? (uint32_t)__builtin_amdgcn_readlane((int)ds0, (int)dsp)
                                        : (uint32_t)__builtin_amdgcn_readlane((int)ds1, (int)(dsp - 64u));
                    } else { // pop the frontier minimum
                        F = wave_min(fminf(fl0, fl1));
                        const uint64_t b0 = __ballot(fl0 == F);
                        if (b0) {
                            const uint32_t slot = (uint32_t)__builtin_ctzll(b0);
                            cur = (uint32_t)__builtin_amdgcn_readlane((int)fr0, (int)slot);
                            fl0 = (lane == slot) ? INFINITY : fl0;
                            fu0 &= ~(1ull << slot);
                        } else {
                            const uint64_t b1 = __ballot(fl1 == F);
                            const uint32_t slot = (uint32_t)__builtin_ctzll(b1);
                            cur = (uint32_t)__builtin_amdgcn_readlane((int)fr1, (int)slot);
                            fl1 = (lane == slot) ? INFINITY : fl1;
                            fu1 &= ~(1ull << slot);
                        }
                    }
                }
                cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur);
                ++iters;
                // issue the popped record's 64-B scalar fetch now: its latency overlaps the compositing below
                const bool is_leaf = (cur & kLeafBit) != 0u;
                const float4* pbase = is_leaf ? a.rec : a.nodes;
                const uint32_t pidx = (is_leaf ? leaf_first(cur) : cur) * 4u;
                const float4 p0 = sload4(pbase, pidx), p1 = sload4(pbase, pidx + 1), p2 = sload4(pbase, pidx + 2),
                             p3 = sload4(pbase, pidx + 3);

                // ---- composite every buffered event with t < F (and key < cutoff), in key order ----
                while (true) {
                    const bool can_ = alive && (k0 != kKeyInvalid) && (key_t(k0) < F) && (k0 < cutoff);
                    if (!__any(can_)) break;
                    const uint64_t ek = k0;
                    const uint32_t cell = (uint32_t)(ek & kCellMask);
                    const uint32_t id = skey_id(ek);
                    float ea = 0.0f, eo = INFINITY;
                    if (can_) { ea = PL_ALPHA(cell); eo = PL_OTHER(cell); }
                    if (can_) { // shaders/tracer.cuh:352-367
                        if (COUNT) c.hit_evals++;
                        last_key = ek | kCellMask; // nothing with the same (t, id, exit) can compare above it
                        if (a.p.alpha_min < ea) {
                            f3 L;
                            if (!SH) {
                                const float4 cc = a.color0[id];
                                L = mk3(cc.x, cc.y, cc.z);
                            } else {
                                L = sh_radiance(a.sh + (size_t)id * 48, dn, a.p.sh_degree_max);
                            }
                            radiance = add3(radiance, mul3s(mul3s(L, T), ea));
                            T *= (1.0f - ea);
                        }
                        if (!(T > minT)) alive = false;
                    }
                    // pop slot 0; an entry whose exit lies inside the segment is re-keyed to its exit event and
                    // keeps its payload cell, otherwise the cell is released
                    const bool rekey = can_ && ((((uint32_t)ek) & 16u) == 0u) && (eo < t_hi);
                    const uint64_t nk = rekey ? (mk_skey(eo, id, 1) | (uint64_t)cell) : kKeyInvalid;
                    pmask = (can_ && !rekey) ? (pmask & ~(1u << cell)) : pmask;
                    SLOT_SHIFT(0, 1) SLOT_SHIFT(1, 2) SLOT_SHIFT(2, 3) SLOT_SHIFT(3, 4) SLOT_SHIFT(4, 5) SLOT_SHIFT(5, 6)
                    SLOT_SHIFT(6, 7) SLOT_SHIFT(7, 8) SLOT_SHIFT(8, 9) SLOT_SHIFT(9, 10) SLOT_SHIFT(10, 11)
                    k11 = can_ ? kKeyInvalid : k11;
                    if (__any(rekey)) { // wave-uniform branch
                        if (rekey) PL_OTHER(cell) = INFINITY;
                        SLOT_INSERT(nk) // a slot was just freed: it fits
                    }
                }
                if (final_sweep) break;

                const float cut_t = (cutoff != kKeyInvalid) ? key_t(cutoff) : t_hi;
                if (is_leaf) {
                    const uint32_t first = leaf_first(cur), cnt = leaf_count(cur);
                    float4 r0 = p0, r1 = p1, r2 = p2, r3 = p3;
                    for (uint32_t j = 0; j < cnt; j++) {
                        if (j) {
                            const uint32_t idx = (first + j) * 4u;
                            r0 = sload4(a.rec, idx); r1 = sload4(a.rec, idx + 1); r2 = sload4(a.rec, idx + 2);
                            r3 = sload4(a.rec, idx + 3);
                        }
                        if (COUNT) c.fetches++;
                        const f3 mu = mk3(r0.x, r0.y, r0.z);
                        m33 A;
                        A.a[0] = r1.x; A.a[1] = r1.y; A.a[2] = r1.z;
                        A.a[3] = r2.x; A.a[4] = r2.y; A.a[5] = r2.z;
                        A.a[6] = r3.x; A.a[7] = r3.y; A.a[8] = r3.z;
                        const f3 o_g = matvec(A, sub3(o, mu));
                        const f3 d_g = matvec(A, d);
                        if (!__any(alive && proxy_sphere_maybe(o_g, d_g, r0.w))) continue; // no lane can touch it
                        if (COUNT && alive) c.proxy_tests++;
                        float te, tx;
                        const bool hit = proxy_slabs(o_g, d_g, r0.w, te, tx) && alive;
                        const uint32_t id = __float_as_uint(r2.w);
                        const uint64_t ke = mk_skey(te, id, 0), kx = mk_skey(tx, id, 1);
                        // float compares first: te/tx may be negative or NaN, the unsigned key compares assume t > 0
                        const bool in_e = hit && (te >= t_lo) && (te < t_hi) && (ke > pass_lo);
                        const bool in_x = hit && (tx >= t_lo) && (tx < t_hi) && (kx > pass_lo);
                        const uint64_t k_first = in_e ? ke : (in_x ? kx : kKeyInvalid); // the slot's first pending event
                        const bool ins = (k_first != kKeyInvalid) && (k_first < cutoff);
                        if (__any(ins)) { // wave-uniform branch
                            // alpha does not depend on the hit distance (shaders/tracer.cuh:354-357)
                            const float alpha = fminf(0.99f, response_from(A, mu, o, d, o_g, d_g) * r1.w);
                            const float other = (in_e && in_x) ? tx : INFINITY;
                            // window full: the largest pending key is dropped (the new one or slot 11's) and the lane
                            // becomes lossy beyond it
                            const bool full = k11 != kKeyInvalid;
                            const bool take = ins && (!full || k_first < k11);
                            const uint64_t dropped = (ins && full) ? (take ? (k11 | kCellMask) : k_first) : kKeyInvalid;
                            cutoff = (dropped < cutoff) ? dropped : cutoff;
                            const uint32_t cell = full ? (uint32_t)(k11 & kCellMask) : (uint32_t)__builtin_ctz(~pmask);
                            k11 = (take && full) ? kKeyInvalid : k11;
                            pmask = take ? (pmask | (1u << cell)) : pmask;
                            if (take) { PL_OTHER(cell) = other; PL_ALPHA(cell) = alpha; }
                            SLOT_INSERT(take ? (k_first | (uint64_t)cell) : kKeyInvalid)
                        }
                    }
                } else {
                    const float4 q0 = p0, q1 = p1, q2 = p2, q3 = p3;
                    if (COUNT) { c.fetches++; if (alive) c.node_visits++; }
                    float n0, f0, n1, f1;
                    box_interval(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, ri, n0, f0);
                    box_interval(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, ri, n1, f1);
                    // a lane wants a child when its box overlaps the lane's open interval (last composited t, cutoff)
                    const float lo = key_t(last_key);
                    const bool h0 = alive && (n0 <= f0) && (f0 >= lo) && (n0 <= cut_t) && (n0 < t_hi);
                    const bool h1 = alive && (n1 <= f1) && (f1 >= lo) && (n1 <= cut_t) && (n1 < t_hi);
                    const uint32_t c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y);
                    const bool any0 = __any(h0), any1 = __any(h1);
                    // both reductions back to back (independent chains overlap)
                    const float lam0 = wave_min(h0 ? fmaxf(n0, 0.0f) : INFINITY);
                    const float lam1 = wave_min(h1 ? fmaxf(n1, 0.0f) : INFINITY);
                    if (dsp) { // already depth-first below a full frontier: stay depth-first
                        if (any1) DFS_PUSH(c1)
                        if (any0) DFS_PUSH(c0)
                    } else {
                        // a child whose key equals the bound F just popped is a minimum of the frontier (every entry
                        // is >= F): expand it right away instead of pushing and popping it
                        const bool by0 = any0 && (lam0 <= F);
                        const bool by1 = any1 && !by0 && (lam1 <= F);
                        bypass = by0 ? c0 : (by1 ? c1 : kNoRoot);
                        if (any0 && !by0) {
                            bool ok;
                            FRONTIER_PUSH(lam0, c0, ok)
                            if (!ok) DFS_PUSH(c0)
                        }
                        if (any1 && !by1) {
                            bool ok = false;
                            if (!dsp) FRONTIER_PUSH(lam1, c1, ok)
                            if (!ok) DFS_PUSH(c1)
                        }
                    }
                }
            }
            // a lane goes again only if it dropped something and still has transmittance left
            const bool progressed = last_key != pass_lo;
            stalls = progressed ? 0u : stalls + 1u;
            alive = alive && (cutoff != kKeyInvalid) && (stalls < 2u);
        }
        if (a.cost && lane == 0) atomicMax(&a.cost[blk], iters);
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
        const uint64_t mask = __ballot(cont);
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
            if (c.fetches) atomicAdd(&a.counters[6], (unsigned long long)c.fetches);
        }
    }
}

} // namespace

int launch_render_stream(const RenderArgs& a, bool count, bool mesh, hipStream_t stream, std::string* err)
{
    if (a.n_blocks == 0) return GRT_OK;
    const bool sh = a.p.sh_degree_max > 0;
    void (*fn)(const RenderArgs);
    if (!mesh)
        fn = count ? (sh ? k_render_stream<true, true, false> : k_render_stream<true, false, false>)
                   : (sh ? k_render_stream<false, true, false> : k_render_stream<false, false, false>);
    else
        fn = count ? (sh ? k_render_stream<true, true, true> : k_render_stream<true, false, true>)
                   : (sh ? k_render_stream<false, true, true> : k_render_stream<false, false, true>);
    hipLaunchKernelGGL(fn, dim3(a.n_blocks), dim3(kBlock), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (err) *err = std::string("k_render_stream launch: ") + hipGetErrorString(e);
        return GRT_ERR_HIP;
    }
    return GRT_OK;
}

} // namespace grt
