// grt_render.hip — the render kernel: raygen + mesh closest-hit + k-nearest Gaussian traversal +
// integration, one launch per frame (replaces optixLaunch of shaders/tracer.cu:17-187 and the OptiX
// RT-core traversal behind optixTrace, src/GaussianTracer.cpp:525-534).
//
// Mapping: one ray per lane; a wave64 is an 8x8 pixel tile (coherent rays), a 256-thread
// workgroup a 16x16 block; workgroups are dealt to XCDs so that each XCD's L2 sees a contiguous
// run of screen blocks.  Traversal is per-lane with a short stack in LDS (stack[level][thread],
// conflict-free), near child first, far child deferred; the k = 7 hit buffer lives in registers
// as 64-bit keys (t, particle id, entry<exit) + alpha.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

#include "grt_device.h"
#include "grt_internal.h"
#include "grt_mesh.h"

namespace grt {

constexpr int K = 7;           // MaxNumHitPerTrace, shaders/tracer.cuh:11
#ifndef GRT_BOUNCE_K
#define GRT_BOUNCE_K 10
#endif
constexpr int kBounceK = GRT_BOUNCE_K; // k-buffer of the bounce stage: same hits in the same order, fewer re-traversal rounds
constexpr int kBlock = 256;

struct Cnt {
    uint32_t rays = 0, segments = 0, hit_evals = 0, rounds = 0, node_visits = 0, proxy_tests = 0, iters = 0;
};

template <int KK>
struct KBuf {
    uint64_t key[KK];
    float alpha[KK];
};

template <int KK>
__device__ __forceinline__ void kbuf_insert(KBuf<KK>& kb, uint64_t key, float alpha)
{
    // same effect as the 7 compare-and-swap steps of __anyhit__anyhit (shaders/tracer.cu:124-146)
    if (key >= kb.key[KK - 1]) return;
#pragma unroll
    for (int i = 0; i < KK; i++) {
        if (key == kb.key[i]) return; // the same event again: a split particle met through another of its pieces
        if (key < kb.key[i]) {
            const uint64_t tk = kb.key[i];
            const float ta = kb.alpha[i];
            kb.key[i] = key;
            kb.alpha[i] = alpha;
            key = tk;
            alpha = ta;
        }
    }
}

// one k-nearest round: traceGPs + __anyhit__ (shaders/tracer.cuh:289-326, shaders/tracer.cu:136-153)
// (limit: value of c.iters at which the segment gives up; false = gave up)
template <bool COUNT, int KK>
__device__ __forceinline__ bool gps_round(const RenderArgs& a, uint32_t* __restrict__ stk, f3 o, f3 d,
                                          const rayinv& ri, uint64_t last_key, float t_hi, KBuf<KK>& kb, Cnt& c,
                                          uint32_t limit)
{
#pragma unroll
    for (int i = 0; i < KK; i++) {
        kb.key[i] = kKeyInvalid;
        kb.alpha[i] = 0.0f;
    }
    const float t_lo = key_t(last_key);
    float bound = t_hi; // nothing beyond the current k-th nearest hit can enter the buffer
    uint32_t sp = 0;
    uint32_t cur = a.root_ref;
    while (true) {
        c.iters++;
        if (c.iters > limit) return false;
        if (cur & kLeafBit) {
            const uint32_t first = leaf_first(cur), cnt = leaf_count(cur);
            for (uint32_t j = 0; j < cnt; j++) {
                const float4* __restrict__ r = a.rec + (size_t)(first + j) * 4;
                const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
                if (COUNT) c.proxy_tests++;
                const f3 mu = mk3(r0.x, r0.y, r0.z);
                m33 A;
                A.a[0] = r1.x; A.a[1] = r1.y; A.a[2] = r1.z;
                A.a[3] = r2.x; A.a[4] = r2.y; A.a[5] = r2.z;
                A.a[6] = r3.x; A.a[7] = r3.y; A.a[8] = r3.z;
                const f3 o_g = matvec(A, sub3(o, mu));
                const f3 d_g = matvec(A, d);
                float te, tx;
                if (proxy_sphere_maybe(o_g, d_g, r0.w) && proxy_slabs(o_g, d_g, r0.w, te, tx)) {
                    // (a piece of a split proxy reports an event only when the event's point lies in its cell)
                    const uint32_t cellb = __float_as_uint(r3.w);
                    const bool in_e = (te >= t_lo) && (te < t_hi) && (!cellb || piece_owns(cellb, r0.w, o_g, d_g, te));
                    const bool in_x = (tx >= t_lo) && (tx < t_hi) && (!cellb || piece_owns(cellb, r0.w, o_g, d_g, tx));
                    if (in_e || in_x) {
                        // alpha does not depend on the hit distance (shaders/tracer.cuh:354-357):
                        // evaluated once, carried by the entry and the exit hit
                        const float alpha = fminf(0.99f, response_from(A, mu, o, d, o_g, d_g) * r1.w);
                        const uint32_t id = __float_as_uint(r2.w);
                        if (in_e) {
                            const uint64_t k = mk_key(te, id, 0);
                            if (k > last_key) kbuf_insert(kb, k, alpha);
                        }
                        if (in_x) {
                            const uint64_t k = mk_key(tx, id, 1);
                            if (k > last_key) kbuf_insert(kb, k, alpha);
                        }
                        if (kb.key[KK - 1] != kKeyInvalid) bound = key_t(kb.key[KK - 1]);
                    }
                }
            }
            if (sp == 0) break;
            cur = stk[(--sp) * kBlock];
        } else {
            const float4* __restrict__ q = a.nodes + (size_t)cur * 4;
            const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            if (COUNT) c.node_visits++;
            float n0, f0, n1, f1;
            box_interval(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, ri, n0, f0);
            box_interval(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, ri, n1, f1);
            const bool h0 = (n0 <= f0) && (f0 >= t_lo) && (n0 <= bound);
            const bool h1 = (n1 <= f1) && (f1 >= t_lo) && (n1 <= bound);
            const uint32_t c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y);
            if (h0 && h1) {
                const bool first0 = n0 <= n1;
                stk[(sp++) * kBlock] = first0 ? c1 : c0;
                cur = first0 ? c0 : c1;
            } else if (h0) {
                cur = c0;
            } else if (h1) {
                cur = c1;
            } else {
                if (sp == 0) break;
                cur = stk[(--sp) * kBlock];
            }
        }
    }
    return true;
}

// trace() — shaders/tracer.cuh:328-373
template <bool COUNT, int KK>
__device__ __forceinline__ bool trace_gaussians(const RenderArgs& a, uint32_t* __restrict__ stk, f3 o, f3 d,
                                                float t_min, float t_max, float& density, f3& radiance, Cnt& c,
                                                uint32_t limit = 0xFFFFFFFFu)
{
    float T = 1.0f - density;
    const float epsT = 1e-9f;
    float lastT = t_min;
    radiance = mk3(0.0f, 0.0f, 0.0f);
    if (COUNT) c.segments++;
    if (a.root_ref == kNoRoot) return true; // no hittable particle: density unchanged
    const f3 dn = normalize3(d);
    const rayinv ri = mk_rayinv(o, d);
    uint64_t last_key = mk_key(lastT + epsT, 0x7FFFFFFFu, 1);
    const float t_hi = t_max + epsT;
    const float minT = a.p.minTransmittance;
    KBuf<KK> kb;
    while (lastT <= t_max && T > minT) {
        if (!gps_round<COUNT, KK>(a, stk, o, d, ri, last_key, t_hi, kb, c, limit)) return false;
        if (COUNT) c.rounds++;
        if (kb.key[0] == kKeyInvalid) break;
#pragma unroll
        for (int i = 0; i < KK; i++) {
            if (kb.key[i] != kKeyInvalid && T > minT) {
                if (COUNT) c.hit_evals++;
                lastT = fmaxf(key_t(kb.key[i]), lastT);
                const float hitAlpha = kb.alpha[i];
                if (a.p.alpha_min < hitAlpha) {
                    const uint32_t id = key_id(kb.key[i]);
                    f3 L;
                    if (a.p.sh_degree_max == 0) {
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
        if (kb.key[KK - 1] == kKeyInvalid) break; // fewer than KK hits left: the next round would be empty
        last_key = kb.key[KK - 1];
    }
    density = 1.0f - T;
    return true;
}

// traceMesh on a per-lane LDS stack (grt_mesh.h)
template <bool COUNT>
__device__ __forceinline__ MeshHit mesh_closest(const RenderArgs& a, uint32_t* __restrict__ stk, f3 o, f3 d,
                                                float tmin, float tmax, Cnt& c)
{
    return mesh_closest_t<COUNT, kBlock>(a, stk, o, d, tmin, tmax, c.iters, c.node_visits);
}

// __raygen__raygeneration bounce loop (shaders/tracer.cu:58-106), resumable from a RayState
// seg_budget: iterations ONE Gaussian segment may take; a ray over it stops, *retry receives its state at the start of
// that iteration (nothing of the iteration is kept, counters included) and *gave_up is set: such a ray is one of the few
// with hundreds of events, and is finished wave-cooperatively (tile kernel, one ray per wave)
template <bool COUNT, int KK>
__device__ __forceinline__ f3 shade_ray(const RenderArgs& a, uint32_t* __restrict__ stk, RayState st, Cnt& c,
                                        uint32_t seg_budget = 0xFFFFFFFFu, RayState* retry = nullptr, bool* gave_up = nullptr)
{
    f3 curO = st.curO, curD = st.curD;
    f3 accumColor = st.accumColor, directLight = mk3(0, 0, 0);
    float accumAlpha = st.accumAlpha, blocking = st.blocking, density = st.density;
    uint32_t numBounces = st.numBounces, timeout = st.timeout;
    while (length3(curD) > 0.1f && numBounces < a.p.max_bounces) {
        const f3 ray_o = curO, ray_d = curD;
        const Cnt c0 = c;
        const uint32_t nb0 = numBounces;
        const float density0 = density;
        int state = MeshPass;
        const MeshHit mh = mesh_closest<COUNT>(a, stk, ray_o, ray_d, kTraceMeshTmin, kTraceMeshTmax, c);
        f3 normal;
        float seg_tmax;
        mesh_shade(a, mh, ray_o, ray_d, state, seg_tmax, normal, curO, curD, numBounces);
        // the single Gaussian segment of this iteration (one call site keeps the kernel small)
        f3 rad;
        const uint32_t limit = (seg_budget == 0xFFFFFFFFu) ? seg_budget : c.iters + seg_budget;
        if (!trace_gaussians<COUNT, KK>(a, stk, ray_o, ray_d, a.p.t_min, seg_tmax, density, rad, c, limit)) {
            c = c0;
            retry->curO = ray_o; retry->curD = ray_d; retry->accumColor = accumColor;
            retry->accumAlpha = accumAlpha; retry->blocking = blocking; retry->density = density0;
            retry->numBounces = nb0; retry->timeout = timeout;
            *gave_up = true;
            return accumColor;
        }
        const float alpha = density;
        if (state == Terminate) { // renderNormal, shaders/tracer.cuh:417-428
            accumColor = add3(accumColor, rad);
            accumAlpha += alpha;
            const f3 normalColor = mul3s(add3(normal, mk3(1.0f, 1.0f, 1.0f)), 0.5f);
            accumColor = add3(accumColor, mul3s(normalColor, 1.0f - alpha));
            accumAlpha += (1.0f - alpha);
            break;
        }
        if (state == LastGaussianPass) { // shaders/tracer.cu:68-82
            directLight = mul3s(rad, alpha);
            accumAlpha = clampf(accumAlpha + alpha, 0.0f, 1.0f);
        } else { // shaders/tracer.cu:84-98
            accumColor = add3(accumColor, mul3s(rad, 1.0f - accumAlpha));
            accumAlpha = clampf(accumAlpha + alpha, 0.0f, 1.0f);
            blocking = clampf(blocking + alpha, 0.0f, 1.0f);
        }
        accumColor = add3(accumColor, mul3s(directLight, 1.0f - blocking)); // shaders/tracer.cu:101
        timeout += 1;
        if (timeout > kTimeoutIterations) break;
    }
    return accumColor;
}

// workgroup -> screen block: consecutive workgroup ids go round-robin over the 8 XCDs; give each
// XCD a contiguous run of blocks so neighbouring screen blocks share an L2 (speed only)

__device__ __forceinline__ RayState fresh_ray(f3 o, f3 d)
{
    RayState st;
    st.curO = o; st.curD = d; st.accumColor = mk3(0, 0, 0);
    st.accumAlpha = 0.0f; st.blocking = 0.0f; st.density = 0.0f;
    st.numBounces = 0; st.timeout = 0;
    return st;
}

// ---- wavefront pipeline for mesh frames -------------------------------------------------------------------
// stage 1 (this kernel): camera ray + mesh closest hit + closest-hit shading for every pixel, one record per thread
//   prec[3*i+0] = (seg_tmax, flags, curO.x, curO.y)   flags = state | numBounces << 8 | have_ray << 16
//   prec[3*i+1] = (curO.z, curD.x, curD.y, curD.z)     prec[3*i+2] = (normal.xyz, 0)
// stage 2 (grt_render_tile.hip / grt_render_stream.hip, MESH = true): the coherent primary Gaussian segment
//   [t_min, seg_tmax] on the wave-per-tile kernel, first iteration of the compositing; the rays that go on are written
//   to the continuation queue, one 64-entry chunk per tile (lane l in slot l)
// stage 3, tile kernel only, bundle_rounds times: k_queue_mesh (the mesh hit of every queued ray, per lane; the mesh
//   tree is small) + the tile kernel with BUNDLE = true (one wave per chunk: the bounced rays of a tile are still a
//   bundle, their Gaussian segment is traced wave-cooperatively) -> the next queue
// stage 4 (k_bounce): whatever is still going finishes its bounce loop on the per-lane traversal
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_queue_mesh(const RenderArgs a)
{
    extern __shared__ uint32_t lds_stack[];
    uint32_t* stk = lds_stack + threadIdx.x;
    Cnt c;
    const uint32_t lane = threadIdx.x & 63u;
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; // queue entry
    if ((uint32_t)(i >> 6) >= *a.qcount_in) return;             // wave-uniform: whole chunks
    const float4 q0 = a.queue_in[i * 4], q1 = a.queue_in[i * 4 + 1], q3 = a.queue_in[i * 4 + 3];
    const bool have_ray = (__float_as_uint(q3.y) >> 31) != 0u;
    const f3 ray_o = mk3(q0.x, q0.y, q0.z), ray_d = mk3(q0.w, q1.x, q1.y);
    int state = MeshPass;
    float seg_tmax = a.p.t_max;
    f3 normal = mk3(0, 0, 0), curO = mk3(0, 0, 0), curD = mk3(0, 0, 0);
    uint32_t numBounces = __float_as_uint(q3.x);
    if (have_ray) {
        const MeshHit mh = mesh_closest<COUNT>(a, stk, ray_o, ray_d, kTraceMeshTmin, kTraceMeshTmax, c);
        mesh_shade(a, mh, ray_o, ray_d, state, seg_tmax, normal, curO, curD, numBounces);
    }
    const uint32_t flags = (uint32_t)state | (numBounces << 8) | ((have_ray ? 1u : 0u) << 16);
    a.prec[i * 3 + 0] = make_float4(seg_tmax, __uint_as_float(flags), curO.x, curO.y);
    a.prec[i * 3 + 1] = make_float4(curO.z, curD.x, curD.y, curD.z);
    a.prec[i * 3 + 2] = make_float4(normal.x, normal.y, normal.z, 0.0f);
    // Bundle verdicts (first round only): a wave is one chunk = the bounced rays of one 8x8 tile.  A tile that gave up as a bundle in an
    // earlier frame OF THIS VIEW (RenderArgs::bverdict) has its rays put on the early list of the one-ray-per-wave kernel here, and its chunk is
    // marked for the bundle kernel to skip: the two kernels then run side by side, and the bundle that would be thrown away is not run.
    if (a.bverdict) {
        const uint32_t chunk = (uint32_t)(i >> 6);
        const uint32_t u = a.qunit[chunk];
        const bool v = a.bverdict[u] == a.bverdict_epoch; // given under this very view: exact (verdicts of other views do not count)
        if (lane == 0) a.qskip[chunk] = v ? 1u : 0u;
        if (v) {
            const uint64_t vm = __ballot(have_ray);
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(a.hcount_a, (uint32_t)__popcll(vm));
            base = (uint32_t)__shfl((int)base, 0);
            if (have_ray) a.heavy_a[base + (uint32_t)__popcll(vm & ((1ull << lane) - 1ull))] = (uint32_t)i;
        }
    }
    if (COUNT) {
        uint32_t x = c.node_visits;
        for (int off = 32; off > 0; off >>= 1) x += (uint32_t)__shfl_xor((int)x, off);
        if (lane == 0 && x) {
            atomicAdd(&a.counters[4], (unsigned long long)x);
            atomicAdd(&a.counters[6], (unsigned long long)x);
        }
    }
}

template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_primary_mesh(const RenderArgs a)
{
    extern __shared__ uint32_t lds_stack[];
    uint32_t* stk = lds_stack + threadIdx.x;
    Cnt c;
    const uint32_t blk = blockIdx.x;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t lx = (wave & 1u) * 8u + (lane & 7u), ly = (wave >> 1) * 8u + (lane >> 3);
    uint32_t px, py;
    bool in_frame;
    if (a.mode == 0) {
        px = a.x0 + (blk % a.nbx) * 16u + lx;
        py = a.y0 + (blk / a.nbx) * 16u + ly;
        in_frame = (px < a.x1) && (py < a.y1);
    } else {
        const uint32_t per_tile = a.nbx * a.nby;
        const uint32_t j = blk / per_tile, sub = blk % per_tile;
        const uint32_t tile = a.first_tile + j * a.tile_stride;
        px = (tile % a.tiles_x) * a.tile_w + (sub % a.nbx) * 16u + lx;
        py = (tile / a.tiles_x) * a.tile_h + (sub / a.nbx) * 16u + ly;
        in_frame = (px < a.p.width) && (py < a.p.height);
    }
    const f3 nU = mk3(-a.p.U[0], -a.p.U[1], -a.p.U[2]), nV = mk3(-a.p.V[0], -a.p.V[1], -a.p.V[2]);
    const f3 W = mk3(a.p.W[0], a.p.W[1], a.p.W[2]);
    const f3 eye = mk3(a.p.eye[0], a.p.eye[1], a.p.eye[2]);
    f3 dir = mk3(0, 0, 0);
    bool have_ray = in_frame;
    if (in_frame) {
        if (!a.p.mode_fisheye) get_ray(px, py, nU, nV, W, a.p.width, a.p.height, dir);
        else have_ray = get_fisheye_ray(px, py, nU, nV, W, a.p.width, a.p.height, dir);
    }
    if (COUNT && have_ray) c.rays++;
    have_ray = have_ray && (length3(dir) > 0.1f) && (a.p.max_bounces > 0u); // loop guard, shaders/tracer.cu:59
    int state = MeshPass;
    float seg_tmax = a.p.t_max;
    f3 normal = mk3(0, 0, 0), curO = mk3(0, 0, 0), curD = mk3(0, 0, 0);
    uint32_t numBounces = 0;
    if (have_ray) {
        const MeshHit mh = mesh_closest<COUNT>(a, stk, eye, dir, kTraceMeshTmin, kTraceMeshTmax, c);
        mesh_shade(a, mh, eye, dir, state, seg_tmax, normal, curO, curD, numBounces);
    }
    const size_t i = ((size_t)blk * kBlock + threadIdx.x) * 3;
    const uint32_t flags = (uint32_t)state | (numBounces << 8) | ((have_ray ? 1u : 0u) << 16);
    a.prec[i + 0] = make_float4(seg_tmax, __uint_as_float(flags), curO.x, curO.y);
    a.prec[i + 1] = make_float4(curO.z, curD.x, curD.y, curD.z);
    a.prec[i + 2] = make_float4(normal.x, normal.y, normal.z, 0.0f);
    if (COUNT) {
        uint32_t v[2] = {c.rays, c.node_visits};
        for (int k = 0; k < 2; k++) {
            uint32_t x = v[k];
            for (int off = 32; off > 0; off >>= 1) x += (uint32_t)__shfl_xor((int)x, off);
            if (lane == 0 && x) {
                atomicAdd(&a.counters[k == 0 ? 0 : 4], (unsigned long long)x);
                if (k == 1) atomicAdd(&a.counters[6], (unsigned long long)x);
            }
        }
    }
}

// ---- stage 1, wave-cooperative (round 6): the 64 camera rays of an 8x8 tile are almost parallel and meet the same few triangles.
// k_primary_mesh walks the mesh tree once PER LANE (a dependent vector load per node, per-lane stacks in LDS: 246 us on the 1080p
// frame with the 32 k-face sphere, a fifth of that frame's Gaussian stage — waves whose rays graze the sphere's limb take up to ten
// times the mean).  Here the wave walks it ONCE: the node — both children's boxes in one 64-B record — comes by a scalar load, every
// lane tests the two boxes against ITS ray and its own current closest hit, a child is entered when any lane wants it (nearer child
// first by the first voting lane's order), the far one waits on ONE stack per wave; a leaf's triangles come by scalar loads too and
// every lane runs the same Moeller-Trumbore test with the same closest-hit rule as mesh_closest_t (grt_mesh.h).  A lane visits a
// superset of the nodes its own walk would visit, the closest hit and the tie rule (lowest face at equal t) do not depend on the
// order, and (t, u, v) come from the same arithmetic on the same triangle: the SAME record, bit for bit
// (tests/test_gpu_parity.py: every mesh test runs both kernels).  Reference: traceMesh + __closesthit__ / __miss__,
// shaders/tracer.cuh:266-287, shaders/tracer.cu:112-122,155-187.
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_primary_mesh_wave(const RenderArgs a)
{
    extern __shared__ uint32_t lds_stack[];
    Cnt c;
    const uint32_t blk = blockIdx.x;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t* stk = lds_stack + wave * a.mstack_depth; // ONE stack per wave (entries are wave-uniform)
    const uint32_t lx = (wave & 1u) * 8u + (lane & 7u), ly = (wave >> 1) * 8u + (lane >> 3);
    uint32_t px, py;
    bool in_frame;
    if (a.mode == 0) {
        px = a.x0 + (blk % a.nbx) * 16u + lx;
        py = a.y0 + (blk / a.nbx) * 16u + ly;
        in_frame = (px < a.x1) && (py < a.y1);
    } else {
        const uint32_t per_tile = a.nbx * a.nby;
        const uint32_t j = blk / per_tile, sub = blk % per_tile;
        const uint32_t tile = a.first_tile + j * a.tile_stride;
        px = (tile % a.tiles_x) * a.tile_w + (sub % a.nbx) * 16u + lx;
        py = (tile / a.tiles_x) * a.tile_h + (sub / a.nbx) * 16u + ly;
        in_frame = (px < a.p.width) && (py < a.p.height);
    }
    const f3 nU = mk3(-a.p.U[0], -a.p.U[1], -a.p.U[2]), nV = mk3(-a.p.V[0], -a.p.V[1], -a.p.V[2]);
    const f3 W = mk3(a.p.W[0], a.p.W[1], a.p.W[2]);
    const f3 eye = mk3(a.p.eye[0], a.p.eye[1], a.p.eye[2]);
    f3 dir = mk3(0, 0, 0);
    bool have_ray = in_frame;
    if (in_frame) {
        if (!a.p.mode_fisheye) get_ray(px, py, nU, nV, W, a.p.width, a.p.height, dir);
        else have_ray = get_fisheye_ray(px, py, nU, nV, W, a.p.width, a.p.height, dir);
    }
    if (COUNT && have_ray) c.rays++;
    have_ray = have_ray && (length3(dir) > 0.1f) && (a.p.max_bounces > 0u); // loop guard, shaders/tracer.cu:59
    const MeshHit best = mesh_closest_wave<COUNT>(a, stk, have_ray, eye, dir, kTraceMeshTmin, kTraceMeshTmax, c.node_visits);
    int state = MeshPass;
    float seg_tmax = a.p.t_max;
    f3 normal = mk3(0, 0, 0), curO = mk3(0, 0, 0), curD = mk3(0, 0, 0);
    uint32_t numBounces = 0;
    if (have_ray) mesh_shade(a, best, eye, dir, state, seg_tmax, normal, curO, curD, numBounces);
    const size_t i = ((size_t)blk * kBlock + threadIdx.x) * 3;
    const uint32_t flags = (uint32_t)state | (numBounces << 8) | ((have_ray ? 1u : 0u) << 16);
    a.prec[i + 0] = make_float4(seg_tmax, __uint_as_float(flags), curO.x, curO.y);
    a.prec[i + 1] = make_float4(curO.z, curD.x, curD.y, curD.z);
    a.prec[i + 2] = make_float4(normal.x, normal.y, normal.z, 0.0f);
    if (COUNT) {
        uint32_t v[2] = {c.rays, c.node_visits};
        for (int k = 0; k < 2; k++) {
            uint32_t x = v[k];
            for (int off = 32; off > 0; off >>= 1) x += (uint32_t)__shfl_xor((int)x, off);
            if (lane == 0 && x) {
                atomicAdd(&a.counters[k == 0 ? 0 : 4], (unsigned long long)x);
                if (k == 1) atomicAdd(&a.counters[6], (unsigned long long)x);
            }
        }
    }
}

// stage 4: finish the queued rays (queue record layout: grt_render_tile.hip / grt_render_stream_body.inc, MESH epilogue)
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_bounce(const RenderArgs a)
{
    extern __shared__ uint32_t lds_stack[];
    uint32_t* stk = lds_stack + threadIdx.x;
    Cnt c;
    const uint32_t lane = threadIdx.x & 63u;
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t n_ent = (size_t)*a.qcount_in * 64u;
    if (((size_t)blockIdx.x * kBlock + (threadIdx.x & ~63u)) >= n_ent) return; // wave-uniform
    float4 q3 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n_ent) q3 = a.queue_in[i * 4 + 3];
    if ((__float_as_uint(q3.y) >> 31) != 0u) { // the slot carries a ray
        const float4 q0 = a.queue_in[i * 4], q1 = a.queue_in[i * 4 + 1], q2 = a.queue_in[i * 4 + 2];
        RayState st;
        st.curO = mk3(q0.x, q0.y, q0.z);
        st.curD = mk3(q0.w, q1.x, q1.y);
        st.accumColor = mk3(q1.z, q1.w, q2.x);
        st.accumAlpha = q2.y; st.blocking = q2.z; st.density = q2.w;
        st.numBounces = __float_as_uint(q3.x); st.timeout = __float_as_uint(q3.y) & 0x7FFFFFFFu;
        const size_t out_idx = (size_t)__float_as_uint(q3.z) | ((size_t)__float_as_uint(q3.w) << 32);
        RayState rt;
        bool gave_up = false;
        const f3 col = shade_ray<COUNT, kBounceK>(a, stk, st, c, a.lane_budget, &rt, &gave_up);
        if (gave_up) { // to the retry queue (packed; read by the one-ray-per-wave launch that follows)
            float4* q = a.fqueue + (size_t)atomicAdd(a.fcount, 1u) * 4;
            q[0] = make_float4(rt.curO.x, rt.curO.y, rt.curO.z, rt.curD.x);
            q[1] = make_float4(rt.curD.y, rt.curD.z, rt.accumColor.x, rt.accumColor.y);
            q[2] = make_float4(rt.accumColor.z, rt.accumAlpha, rt.blocking, rt.density);
            q[3] = make_float4(__uint_as_float(rt.numBounces), __uint_as_float(rt.timeout | 0x80000000u), q3.z, q3.w);
        } else {
            if (a.outf) {
                a.outf[out_idx * 3] = col.x; a.outf[out_idx * 3 + 1] = col.y; a.outf[out_idx * 3 + 2] = col.z;
            }
            if (a.out8) {
                a.out8[out_idx * 3] = quantize8(col.x);
                a.out8[out_idx * 3 + 1] = quantize8(col.y);
                a.out8[out_idx * 3 + 2] = quantize8(col.z);
            }
        }
    }
    if (COUNT) {
        uint32_t v[6] = {0u, c.segments, c.hit_evals, c.rounds, c.node_visits, c.proxy_tests};
#pragma unroll
        for (int k = 1; k < 6; k++) {
            uint32_t x = v[k];
            for (int off = 32; off > 0; off >>= 1) x += (uint32_t)__shfl_xor((int)x, off);
            if (lane == 0 && x) atomicAdd(&a.counters[k], (unsigned long long)x);
            if (lane == 0 && x && k >= 4) atomicAdd(&a.counters[6], (unsigned long long)x);
        }
    }
}

template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_render(const RenderArgs a)
{
    extern __shared__ uint32_t lds_stack[];
    uint32_t* stk = lds_stack + threadIdx.x;
    Cnt c;
    const uint32_t blk = a.order ? a.order[blockIdx.x] : xcd_swizzle(blockIdx.x, a.n_blocks, a.swizzle_chunk);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t lx = (wave & 1u) * 8u + (lane & 7u), ly = (wave >> 1) * 8u + (lane >> 3);

    if (a.mode == 2) { // ray buffer
        const uint64_t i = (uint64_t)blk * kBlock + threadIdx.x;
        if (i < a.n_rays) {
            const float* r = a.rays + i * 6;
            if (COUNT) c.rays++;
            const f3 col = shade_ray<COUNT, K>(a, stk, fresh_ray(mk3(r[0], r[1], r[2]), mk3(r[3], r[4], r[5])), c);
            a.outf[i * 3] = col.x; a.outf[i * 3 + 1] = col.y; a.outf[i * 3 + 2] = col.z;
        }
    } else {
        uint32_t px, py;
        size_t out_idx;
        bool in_frame;
        if (a.mode == 0) { // window of the full frame
            px = a.x0 + (blk % a.nbx) * 16u + lx;
            py = a.y0 + (blk / a.nbx) * 16u + ly;
            in_frame = (px < a.x1) && (py < a.y1);
            out_idx = (size_t)py * a.p.width + px; // shaders/tracer.cuh:487
        } else { // compact tile list
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
        bool write = in_frame || (a.mode == 1);
        f3 col = mk3(0.0f, 0.0f, 0.0f);
        if (in_frame) {
            const f3 nU = mk3(-a.p.U[0], -a.p.U[1], -a.p.U[2]), nV = mk3(-a.p.V[0], -a.p.V[1], -a.p.V[2]);
            const f3 W = mk3(a.p.W[0], a.p.W[1], a.p.W[2]);
            f3 dir;
            bool have_ray = true;
            if (!a.p.mode_fisheye) get_ray(px, py, nU, nV, W, a.p.width, a.p.height, dir);
            else have_ray = get_fisheye_ray(px, py, nU, nV, W, a.p.width, a.p.height, dir);
            if (have_ray) {
                if (COUNT) c.rays++;
                col = shade_ray<COUNT, K>(a, stk, fresh_ray(mk3(a.p.eye[0], a.p.eye[1], a.p.eye[2]), dir), c);
            }
        }
        if (write) {
            if (a.outf) {
                a.outf[out_idx * 3] = col.x; a.outf[out_idx * 3 + 1] = col.y; a.outf[out_idx * 3 + 2] = col.z;
            }
            if (a.out8) { // writeOutputBuffer, shaders/tracer.cuh:484-496
                a.out8[out_idx * 3] = quantize8(col.x);
                a.out8[out_idx * 3 + 1] = quantize8(col.y);
                a.out8[out_idx * 3 + 2] = quantize8(col.z);
            }
        }
    }

    if (a.cost) { // scheduling feedback: the block's cost is its slowest lane
        uint32_t m = c.iters;
        for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
        if (lane == 0) atomicMax(&a.cost[blk], m);
    }
    if (COUNT) {
        uint32_t v[6] = {c.rays, c.segments, c.hit_evals, c.rounds, c.node_visits, c.proxy_tests};
#pragma unroll
        for (int k = 0; k < 6; k++) {
            uint32_t x = v[k];
            for (int off = 32; off > 0; off >>= 1) x += (uint32_t)__shfl_xor((int)x, off);
            if (lane == 0 && x) atomicAdd(&a.counters[k], (unsigned long long)x);
            if (lane == 0 && x && k >= 4) atomicAdd(&a.counters[6], (unsigned long long)x); // one 64-B fetch per lane visit
        }
    }
}

int launch_render(const RenderArgs& a, bool count, int kernel_variant, uint32_t stack_depth, bool tile_kernel,
                  hipStream_t stream, const LaunchAux* aux, std::string* err)
{
    if (a.n_blocks == 0) return GRT_OK;
    const size_t lds = (size_t)kBlock * sizeof(uint32_t) * (stack_depth ? stack_depth : 1);
    const bool wave_ok = (a.mroot == kNoRoot) && (a.mode != 2) && (stack_depth <= 120);
    const bool streaming = uses_stream_kernel(kernel_variant, a.mode, stack_depth); // same test as do_launch's
    // auto: the single-pass streaming kernel (measured faster than the round-based wave kernel from 10k to 3M
    // Gaussians; the two are bit-identical)
    if (streaming && a.mroot == kNoRoot)
        return tile_kernel ? launch_render_tile(a, count, false, 0, stream, err, aux) : launch_render_stream(a, count, false, stream, aux, err);
    // mesh frames: wavefront pipeline (primary segment on the streaming wave kernel, compaction, per-lane bounces)
    if (streaming) {
        if (!a.prec || !a.queue || !a.qcount) {
            if (err) *err = "wavefront pipeline: continuation buffers missing";
            return GRT_ERR_INVALID;
        }
        if (lds > 160 * 1024) {
            if (err) *err = "BVH height " + std::to_string(stack_depth) + " needs more than 160 KiB of LDS stack";
            return GRT_ERR_LIMIT;
        }
        if (!a.queue_alt) {
            if (err) *err = "wavefront pipeline: continuation buffers missing";
            return GRT_ERR_INVALID;
        }
        // (the kernels that walk only the MESH tree per lane — stage 1's per-lane form, k_queue_mesh — need LDS stacks as deep as THAT tree,
        //  not as the Gaussian tree whose height `stack_depth` also covers (1 M Gaussians: 30 levels = 30 KB per workgroup, five workgroups
        //  per CU; the 32 k-face sphere: 17): more of these latency-bound waves are resident)
        const size_t lds_mesh = a.mstack_depth ? std::min(lds, (size_t)kBlock * sizeof(uint32_t) * a.mstack_depth) : lds;
        // counters of the pipeline: see kWfCounters (grt_internal.h)
        hipError_t e = hipMemsetAsync(a.qcount, 0, sizeof(uint32_t) * kWfCounters, stream);
        auto fp = count ? k_primary_mesh<true> : k_primary_mesh<false>;
        auto fq = count ? k_queue_mesh<true> : k_queue_mesh<false>;
        auto fb = count ? k_bounce<true> : k_bounce<false>;
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(fp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(fq), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(fb), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            if (err) *err = std::string("wavefront setup: ") + hipGetErrorString(e);
            return GRT_ERR_HIP;
        }
        if (tile_kernel && a.mesh_primary_wave == 2u) {
            // stage 1 runs inside stage 2 (grt_render_tile.hip, MODE 0 with MESH): no launch here
        } else if (a.mesh_primary_wave && a.mstack_depth) { // one walk of the mesh tree per 8x8 tile instead of one per lane
            auto fw = count ? k_primary_mesh_wave<true> : k_primary_mesh_wave<false>;
            hipLaunchKernelGGL(fw, dim3(a.n_blocks), dim3(kBlock), sizeof(uint32_t) * 4u * a.mstack_depth, stream, a);
        } else {
            hipLaunchKernelGGL(fp, dim3(a.n_blocks), dim3(kBlock), lds_mesh, stream, a);
        }
        int rc = tile_kernel ? launch_render_tile(a, count, true, 0, stream, err) : launch_render_stream(a, count, true, stream, aux, err);
        if (rc != GRT_OK) return rc;
        RenderArgs b = a;
        b.order = nullptr;
        b.cost = nullptr;
        // the most chunks a queue can hold: one per wave of the primary stage — per 8x8 tile, or per entry of a launch order that
        // runs heavy tiles as parts (do_launch sizes the queues for it); the per-entry stages launch one thread per queue entry
        const uint32_t max_chunks = (tile_kernel && a.order && a.n_launch) ? a.n_launch : a.n_blocks * 4u;
        const uint32_t qblocks = (max_chunks * 64u + kBlock - 1u) / kBlock;
        b.n_launch = max_chunks;
        float4* q[2] = {a.queue, a.queue_alt};
        uint32_t in = 0, stage = 0;
        const uint32_t rounds = tile_kernel ? std::min<uint32_t>(a.bundle_rounds, (uint32_t)kMaxBundleRounds) : 0u;
        b.fcount = a.qcount + 2 * kMaxBundleRounds + 1;
        const bool predict = tile_kernel && a.bverdict && a.qunit && a.qskip && a.heavy_a && aux && aux->aux && aux->fork && aux->join;
        uint32_t* const qun[2] = {a.qunit, a.qunit ? a.qunit + a.qunit_cap : nullptr}; // the tile numbers of the chunks of q[0] / q[1]
        b.bverdict = nullptr; b.qunit = nullptr; b.qunit_out = nullptr; b.qskip = nullptr; b.heavy_a = nullptr; b.hcount_a = nullptr;
        for (uint32_t r = 0; r < rounds; r++) {
            b.queue_in = q[in]; b.qcount_in = a.qcount + stage;
            b.queue = q[in ^ 1u]; b.qcount = a.qcount + stage + 1u;
            b.hcount = a.qcount + kMaxBundleRounds + 1 + r;
            b.hnext = a.qcount + 2 * kMaxBundleRounds + 2 + r;
            // bundle verdicts, per round: the tiles whose rays of THIS round gave up as a bundle under this view
            if (predict) {
                b.bverdict = a.bverdict + (size_t)r * a.n_blocks * 4u; b.qunit = qun[in]; b.qunit_out = qun[in ^ 1u]; b.qskip = a.qskip; b.heavy_a = a.heavy_a;
                b.hcount_a = a.qcount + 3 * kMaxBundleRounds + 3 + 2 * r;
            }
            hipLaunchKernelGGL(fq, dim3(qblocks), dim3(kBlock), lds_mesh, stream, b);
            if (predict) {
                // the tiles known not to be bundles go one ray per wave BESIDE the bundle kernel: the bundles first (short waves, they take
                // the machine and leave it within ~0.1 ms), the resident one-ray-per-wave grid on the second stream fills in behind them
                if ((e = hipEventRecord(aux->fork, stream)) != hipSuccess || (e = hipStreamWaitEvent(aux->aux, aux->fork, 0)) != hipSuccess) {
                    if (err) *err = std::string("wavefront pipeline: fork to the second stream: ") + hipGetErrorString(e);
                    return GRT_ERR_HIP;
                }
            }
            rc = launch_render_tile(b, count, true, 1, stream, err);          // bundles; chunks over budget -> heavy list
            if (rc == GRT_OK && predict) {
                RenderArgs s1 = b;
                s1.heavy = a.heavy_a; s1.hcount = b.hcount_a; s1.hnext = a.qcount + 3 * kMaxBundleRounds + 4 + 2 * r;
                rc = launch_render_tile(s1, count, true, 2, aux->aux, err);   // the early list: one ray per wave, to their end
                if (rc == GRT_OK && ((e = hipEventRecord(aux->join, aux->aux)) != hipSuccess || (e = hipStreamWaitEvent(stream, aux->join, 0)) != hipSuccess)) {
                    if (err) *err = std::string("wavefront pipeline: join of the second stream: ") + hipGetErrorString(e);
                    return GRT_ERR_HIP;
                }
            }
            if (rc == GRT_OK) rc = launch_render_tile(b, count, true, 2, stream, err); // the rays of the chunks that gave up now: one per wave, to their end
            if (rc != GRT_OK) return rc;
            b.bverdict = nullptr; b.qunit = nullptr; b.qunit_out = nullptr; b.qskip = nullptr; b.heavy_a = nullptr; b.hcount_a = nullptr;
            in ^= 1u;
            stage++;
        }
        // whatever still bounces finishes per lane; a segment over the lane budget sends its ray to the retry queue, and
        // those rays are finished one per wave (tile kernel only: the other pipelines have no per-child BVH layout)
        b.queue_in = q[in]; b.qcount_in = a.qcount + stage;
        b.lane_budget = tile_kernel ? a.lane_budget : 0xFFFFFFFFu;
        hipLaunchKernelGGL(fb, dim3(qblocks), dim3(kBlock), lds, stream, b);
        if (tile_kernel) {
            RenderArgs s2 = b;
            s2.queue_in = a.fqueue; s2.heavy = nullptr; s2.hcount = b.fcount; // the retry queue itself is the list
            s2.hnext = a.qcount + 3 * kMaxBundleRounds + 2;
            s2.single_own_mesh = 1;                                           // its rays carry no mesh-hit record
            rc = launch_render_tile(s2, count, true, 2, stream, err);
            if (rc != GRT_OK) return rc;
        }
        e = hipGetLastError();
        if (e != hipSuccess) {
            if (err) *err = std::string("wavefront launch: ") + hipGetErrorString(e);
            return GRT_ERR_HIP;
        }
        return GRT_OK;
    }
    if (kernel_variant == 2 && wave_ok) return launch_render_wave(a, count, stream, err);
    if (lds > 160 * 1024) {
        if (err) *err = "BVH height " + std::to_string(stack_depth) + " needs more than 160 KiB of LDS stack";
        return GRT_ERR_LIMIT;
    }
    auto fn = count ? k_render<true> : k_render<false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) {
        if (err) *err = std::string("hipFuncSetAttribute: ") + hipGetErrorString(e);
        return GRT_ERR_HIP;
    }
    hipLaunchKernelGGL(fn, dim3(a.n_blocks), dim3(kBlock), lds, stream, a);
    e = hipGetLastError();
    if (e != hipSuccess) {
        if (err) *err = std::string("k_render launch: ") + hipGetErrorString(e);
        return GRT_ERR_HIP;
    }
    return GRT_OK;
}

}  // namespace grt
