// grt_internal.h — structures shared by the translation units of libgrt_hip.so (not installed).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <mutex>
#include <vector>

#include "../../include/grt.h"

namespace grt {

// Child reference inside a BVH node: bit 31 set => leaf RANGE of sorted primitives: bits 0-27 = first
// index, bits 28-30 = count-1 (bottom subtrees of <= leaf_max primitives are collapsed into one leaf);
// else index of an internal node.
constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kLeafIndexMask = 0x0FFFFFFFu;
constexpr uint32_t kLeafMaxPrims = 8u;
__host__ __device__ inline uint32_t leaf_first(uint32_t ref) { return ref & kLeafIndexMask; }
__host__ __device__ inline uint32_t leaf_count(uint32_t ref) { return ((ref >> 28) & 7u) + 1u; }
constexpr uint32_t kNoRoot = 0xFFFFFFFFu;
// children per node in the tile kernel's view of the tree (DevBvh::qnodes): 4, 8 or 16
constexpr uint32_t kTileWide = 8u; // (4-wide: C3 +6.5 %, 16-wide: +9-14 %, measured in rounds 2 and 4)

// LBVH in traversal layout.  One 64-B record (4 x float4) per INTERNAL node holding the boxes of
// its two children, so a node fetch decides both descents:
//   q0 = (lo0.x lo0.y lo0.z hi0.x)  q1 = (hi0.y hi0.z lo1.x lo1.y)  q2 = (lo1.z hi1.x hi1.y hi1.z)
//   q3 = (child0, child1, 0, 0) as raw bits
struct DevBvh {
    float4* nodes = nullptr;   // [(n_prims-1) * 4]
    float4* wnodes = nullptr;  // [(n_prims-1) * 8] 4-wide view of the same tree (two binary levels per record):
                               //   24 floats = 4 child boxes, each (lo.x lo.y | hi.x hi.y | lo.z hi.z),
                               //   W6 = 4 child refs, W7 pad; an unused child has ref kNoRoot
    float4* qnodes = nullptr;  // [(n_prims-1) * 2 * kTileWide] the tree laid out per CHILD for the tile kernel
                               //   (grt_render_tile.hip: one lane tests one child box), kTileWide children per record:
                               //   child c of node i at [i*2W + 2c] = (lo.xyz, ref bits), [+1] = (hi.xyz, 0); unused: ref kNoRoot
    float4* pbox = nullptr;    // [n_prims * 2] box of every sorted primitive, (lo.xyz,0)(hi.xyz,r), r = bounding radius about the box centre: what a leaf-range
                               //   child expands to in the tile kernel (built only for the Gaussian BVH)
    uint32_t* level = nullptr; // [n_prims-1] refit pass in which each node was finished (kept when asked for: refit_lbvh)
    uint32_t* order = nullptr; // [n_prims] sorted position -> input primitive index
    uint32_t n_prims = 0;      // valid primitives (leaves)
    uint32_t height = 0;       // levels of internal nodes (bounds the traversal stack)
    uint32_t root_ref = kNoRoot;
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    size_t cap_nodes = 0, cap_order = 0;
};

// Build an LBVH over n_in boxes (invalid primitives have lo.x > hi.x and are left out).
// Returns GRT_OK or an error code (message in *err).
// want_quad: also build the per-child layout (qnodes, pbox) the tile kernel traverses.
// keep_levels: keep the per-node refit order so that refit_lbvh can re-fit the boxes of the SAME hierarchy later.
// size_classes: Gaussian BVH only (GRT_OPT_SIZE_CLASSES of the context that builds).
int build_lbvh(const float4* d_lo, const float4* d_hi, uint32_t n_in, uint32_t leaf_max, bool want_quad, bool keep_levels,
               int size_classes, DevBvh* out, hipStream_t stream, std::string* err, bool widen_area_only = false,
               int rotation_sweeps = -1);
// (widen_area_only: the 8-wide records open the largest box first throughout — trees with pieces; grt_bvh.hip: k_qwiden.
//  rotation_sweeps: GRT_OPT_BVH_ROTATIONS of the context that builds; DevBvh::height is the height of the tree as traversed, rotations included)
// Re-fit every reachable node's boxes to new primitive boxes (same primitives, same order, same hierarchy): what a
// gizmo drag needs (reference: full GAS + IAS rebuild per frame, src/GaussianTracer.cpp:711-794).
int refit_lbvh(const float4* d_lo, const float4* d_hi, uint32_t n_in, DevBvh* bvh, hipStream_t stream, std::string* err);
void free_bvh(DevBvh* b);
// exclusive scan of n uint32 values on the device (hand-written, grt_bvh.hip); synchronises the stream
int device_exclusive_scan_u32(const uint32_t* d_in, uint32_t* d_out, uint32_t n, hipStream_t stream, std::string* err);

// Everything the render kernel reads, passed by value.
struct RenderArgs {
    grt_params p;
    // Gaussian scene
    const float4* rec;    // [n_prox*4] Morton-sorted proxy records (see grt_api.hip: k_gather_records)
    const float4* erec;   // per-eye part of the proxy test, same order; camera frames only: [n_prox] (k_eye_records) for the
                          // streaming kernel, [n_prox*4] (k_eye_records_wide) for the tile kernel
    const float4* nodes;
    const float4* wnodes; // 4-wide records (streaming kernel)
    const float4* qnodes; // 4-wide records, one 32-B slot per child (tile kernel)
    const float4* pbox;   // [n_prox*2] per-proxy boxes in sorted order (tile kernel)
    uint32_t root_ref;
    uint32_t n_prox;
    uint32_t has_pieces;  // the tree's leaves include pieces of split proxies (the record's last word is the cell descriptor)
    const float4* color0; // [n_particles] degree-0 radiance by ORIGINAL particle id
    const float* sh;      // [n_particles][16][3] by original id
    // mesh scene
    const float4* mnodes;
    const float4* tri;    // [n_faces*3] sorted: (v0, face id bits) (v1,0) (v2,0)
    uint32_t mroot;
    uint32_t n_faces;
    const uint32_t* faces;  // [nf][3]
    const float* vnormals;  // [nv][3]
    // outputs
    uint8_t* out8;
    float* outf;
    // work mapping: 0 = window, 1 = tiles, 2 = ray buffer
    uint32_t mode;
    uint32_t x0, y0, x1, y1;
    uint32_t nbx;           // 16x16 blocks per row (window) / per tile row (tiles)
    uint32_t nby;           // per tile column (tiles)
    uint32_t tile_w, tile_h, first_tile, tile_stride, n_tiles, tiles_x;
    const float* rays;
    uint64_t n_rays;
    uint32_t n_blocks;
    uint32_t swizzle_chunk; // see xcd_swizzle (grt_device.h); 0 = identity
    uint32_t n_units;      // scheduling units of order[] / cost[]: n_blocks (256-thread kernels) or 4 * n_blocks (streaming: one per 8x8 tile)
    const uint32_t* order;  // cost-sorted block order from the previous frame (heaviest first) or nullptr
    uint32_t n_launch;      // tile kernel: entries of order[] = workgroups to launch (units + the extra parts of split tiles + padding); 0 = n_units
    uint32_t* cost;         // [n_blocks] per-block cost of THIS frame (max wave iterations), zeroed before launch
    const uint32_t* n_heavy; // device count of leading blocks of `order` that run on the big-window kernel
    uint32_t heavy_role;     // 0 = every block, 1 = only ranks < *n_heavy, 2 = only ranks >= *n_heavy
    unsigned long long* counters; // kNumCounters x u64 or nullptr
    // tile kernel tuning (grt_render_tile.hip)
    uint32_t tile_ready_min; // lanes that must hold a final event before a compositing sweep starts
    float tile_band;         // particles within F * (1 + band) of the front are tested in one batch
    float tile_look;         // nodes within Fn * (1 + look) are expanded in one step
    float tile_band_abs;     // trees with pieces: absolute floor of band / look-ahead (world units: a multiple of the typical proxy size)
    uint32_t tile_reserve;   // free frontier slots below which leaf steps are forced
    uint32_t tile_prio_div;  // the first 1/div of the cost-sorted launch order runs at raised wave priority (0 = off)
    uint32_t quad_parts;     // the launch order's four-way parts run on the quad kernel (MODE 3) beside the camera-ray kernel, which skips them
    const uint32_t* qparts;  // ... those entries of the order, compacted (grt_bvh.hip: k_quad_list), heaviest first
    const uint32_t* qpart_count; // [0] = how many
    uint32_t quad_known;     // host's knowledge of that number + 1 (0 = unknown: the grid is the list's capacity)
    float4* ovf_pool;        // window overflow bags: [chunk][entry][lane] x 16 B, one chunk per tile that overflows
    uint32_t* ovf_next;      // next free chunk (zeroed before the launch)
    uint32_t ovf_chunks;     // chunks in the pool
    uint32_t ovf_entries;    // per-lane capacity of a bag actually used (<= kTileOvfEntries; smaller only in tests)
    uint32_t ovf_cls0;       // chunks (1 or 3) a whole tile starts in when its order entry carries no size class (grt_render_tile.hip kSub)
    // always-on failure signal: a wave that has to give up on a ray (watchdog, stack guard, two passes without progress)
    // ORs its reason into this device word, whatever the kernel variant; grt_sync / grt_get_counters report it
    uint32_t* err_word;
    uint32_t max_iters;      // watchdog of the tile kernel: steps one tile may take (test override: GRT_OPT_MAX_ITERS)
    // wavefront pipeline for mesh frames (grt_render.hip: k_primary_mesh / k_bounce, grt_render_stream.hip MESH=true)
    float4* prec;      // [n_blocks*256][3] mesh-hit records: of the pixels (stage 1), later of the queue entries (stage 3)
    float4* queue;     // [n_blocks*256][4] continuation rays this launch WRITES: 64-entry chunks, one per 8x8 tile that has
                       // a ray going on, lane l in slot l (bit 31 of the timeout word = the slot carries a ray)
    uint32_t* qcount;  // chunks written
    const float4* queue_in;   // the queue this launch READS (stage 3: k_queue_mesh, bundle kernel, k_bounce)
    const uint32_t* qcount_in;
    float4* queue_alt;        // the second queue (launch_render ping-pongs between `queue` and this one)
    uint32_t bundle_rounds;   // bounce iterations on the bundle kernel before k_bounce finishes the rest (tile kernel only)
    uint32_t bundle_budget;   // steps a chunk may take on the bundle kernel before its rays are traced one per wave
    uint32_t* heavy;          // [n_blocks*256] entries of queue_in whose chunk gave up (written by mode 1, read by mode 2;
    uint32_t* hcount;         //   nullptr for mode 2 = the entries 0 .. *hcount - 1 of queue_in themselves)
    uint32_t* hnext;          // mode 2: next entry of the heavy list to be drawn (zeroed with the other counters)
    float4* fqueue;           // [n_blocks*256][4] PACKED retry queue: rays whose segment went over the lane budget in k_bounce
                              // (state at the start of that iteration), finished by the last one-ray-per-wave launch
    uint32_t* fcount;
    uint32_t lane_budget;     // k_bounce: iterations one Gaussian segment may take before its ray goes to the retry queue
    uint32_t mesh_primary_wave; // stage 1 of mesh frames: 1 = one walk of the mesh tree per 8x8 tile (k_primary_mesh_wave), 0 = one per lane
    uint32_t mstack_depth;      // ... entries of a wave's stack: the mesh tree's height + 2
    uint32_t single_own_mesh; // mode 2: the rays come from the retry queue (no mesh-hit record: the wave traces the mesh)
    // bundle verdicts (mesh frames on the tile kernel, first bundle round; GRT_OPT_BUNDLE_PREDICT): a tile whose bounced rays gave up as
    // a bundle is remembered, and the next frames send its continuation rays one per wave AT ONCE — on a list of their own that the
    // one-ray-per-wave kernel works off BESIDE the bundle kernel (second stream) instead of behind it, and without the budget-long bundle
    // that is thrown away.  Same rays through the same two kernels as before, so the same pixels.
    uint32_t* bverdict;       // [bundle round][n_units] the number of the view under which the tile's rays of that round gave up as a bundle (0 = never)
    uint32_t* qunit;          // [chunks] the unit (8x8 tile) a chunk's rays belong to — of the queue the primary stage WRITES, of the queue a
                              // bundle round READS; `qunit_out`: of the queue a bundle round writes (the tile's number travels with its rays)
    uint32_t* qunit_out;
    uint32_t qunit_cap;       // chunks one queue's array of tile numbers holds (the second queue's array follows the first's)
    uint32_t* qskip;          // [chunks] 1 = the chunk's rays are on the early list: the bundle kernel leaves it alone (written by k_queue_mesh)
    uint32_t* heavy_a;        // the early list: entries of queue_in (k_queue_mesh writes it, a mode-2 launch reads it as its `heavy`)
    uint32_t* hcount_a;
    uint32_t bverdict_epoch;  // number of the VIEW (never 0; the host counts a new one whenever the frame's parameters change).  A verdict is the
                              // number of the view it was given under and counts only under that very view, where it is exact (the same
                              // bundle would give up again).  Verdicts of OTHER views were tried as guesses (eight frames each): a tile wrongly
                              // sent one ray per wave costs 64 waves where a wrong "bundle" costs one budget-long wave — after a camera
                              // move the home view ran at 10-11 ms for eight frames (profiles/r06_experiments_log.md 4): not used
    float single_look, single_band; // look-ahead / band of the one-ray-per-wave mode
};

// second stream + events used to run the big-window kernel (heavy blocks) beside the default one
struct LaunchAux {
    hipStream_t aux = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    uint32_t heavy_cap = 0; // grid of the big-window launch (0 = no split)
    bool force_big = false; // GRT_OPT_KERNEL = 4: every block on the big-window kernel (testing)
};
int launch_render(const RenderArgs& a, bool count, int kernel_variant, uint32_t stack_depth, bool tile_kernel,
                  hipStream_t stream, const LaunchAux* aux, std::string* err);
int launch_render_wave(const RenderArgs& a, bool count, hipStream_t stream, std::string* err);
int launch_render_stream(const RenderArgs& a, bool count, bool mesh, hipStream_t stream, const LaunchAux* aux,
                         std::string* err);
int launch_render_tile(const RenderArgs& a, bool count, bool mesh, int mode, hipStream_t stream, std::string* err, const LaunchAux* aux = nullptr);
int launch_render_tile_quad(const RenderArgs& a, bool count, hipStream_t stream, std::string* err); // mode 3 (its own TU)
int launch_render_tile_single(const RenderArgs& a, bool count, hipStream_t stream, std::string* err); // mode 2 (its own TU)
constexpr int kNumCounters = 8;
// bits of RenderArgs::err_word
constexpr uint32_t kErrWatchdog = 1u, kErrStack = 2u, kErrStall = 4u;
constexpr uint32_t kCostStackBit = 0x40000000u, kCostStallBit = 0x20000000u; // give-up reasons in a tile's cost word
// A heavy 8x8 tile may be launched as 2 waves (4 rows each) or 4 waves (4x4 pixels each): an entry of the launch order is
// unit | part << 28 | code << 30 (code 1 = two parts, 2 = four; 0xFFFFFFFF = padding: the wave exits), and the cost word
// such a tile leaves carries its code in bits 27-28 beside the largest step count of its parts — the feedback kernels scale
// that back to the whole tile's (cost_eff: x 9/8 for halves, x 10/8 for quarters — measured: a quarter of the heaviest tile
// of the 1 M scene takes 0.79 of the whole tile's time, a half all of it), or a tile would be split on every other frame only.
// (A larger factor inflates the split tiles' costs frame over frame until their parts no longer fit the launch.)
// A WHOLE tile's entry (code 0) uses the part field for the size class of its overflow bags: the chunks of the pool it starts in (1, 2,
// 3; 0 = no cost word yet), from the two lowest bits of its cost word (grt_render_tile.hip kBagKeep1 / kBagKeep2, grt_bvh.hip bag_class).
constexpr uint32_t kOrderUnitMask = 0x0FFFFFFFu, kOrderPad = 0xFFFFFFFFu;
// code 3 = a four-way part that runs on the QUAD kernel (grt_render_tile.hip MODE 3): k_quad_list (grt_bvh.hip) re-codes the first
// kQuadListCap four-way entries of an order and lists them for that kernel, whose grid is the list's capacity; what does not fit
// stays code 2, a part wave of the camera-ray kernel.  (A part's unit is < 2^28, so code 3 | part 3 | unit never equals kOrderPad.)
constexpr uint32_t kQuadListCap = 4096u;
constexpr uint32_t kTileResidentWaves = 256u * 16u; // MI355X: 256 CUs x 16 waves of the camera-ray kernel (128 VGPRs, < 10 KB of LDS)
constexpr uint32_t kCostPartShift = 27u, kCostStepsMask = 0x07FFFFFFu;
__host__ __device__ inline uint32_t cost_eff(uint32_t c)
{
    const uint32_t code = (c >> kCostPartShift) & 3u, steps = c & kCostStepsMask;
    return code ? (uint32_t)(((uint64_t)steps * (8u + code)) >> 3) : steps;
}
constexpr uint32_t kTileMaxItersDefault = 1u << 21; // a heavy C3 tile takes ~2000 steps
constexpr uint32_t kTileStack = 288u;               // depth-first overflow stack of the tile kernel (entries)
// The stack receives the batch that overflowed (<= 64) plus up to kTileWide - 1 siblings per 8-wide level below it
// (three binary levels per wide level): the launcher sends taller trees to the streaming kernel.
inline bool tile_stack_fits(uint32_t height) { return ((height + 2u) / 3u) * (kTileWide - 1u) + 64u <= kTileStack; }
constexpr int kMaxBundleRounds = 4; // GRT_OPT_BUNDLE_ROUNDS <= this
// counters of the wavefront pipeline (one uint32 each, zeroed per frame): [0 .. R] chunks written by stage 2 and by bundle
// round r, [R+1 .. 2R] rays on the heavy list of round r, [2R+1] entries of the retry queue, [2R+2 .. 3R+1] the draw
// counter of round r's one-ray-per-wave launch, [3R+2] the draw counter of the last one
//, [3R+3+2r] rays on round r's EARLY heavy list (bundle verdicts: RenderArgs::bverdict), [3R+4+2r] its draw counter
constexpr int kWfCounters = 5 * kMaxBundleRounds + 3;
constexpr uint32_t kTileOvfEntries = 96u; // per-lane capacity of a window-overflow bag
constexpr uint32_t kTileOvfSub = 32u;     // ... handed out this many entries at a time: a chunk of the pool = 32 entries x 64 lanes
constexpr size_t kTileOvfChunkBytes = (size_t)kTileOvfSub * 64 * 16; // 32 KiB; a tile holds up to kTileOvfEntries / kTileOvfSub of them
constexpr uint32_t kTileOvfChunksPerTile = kTileOvfEntries / kTileOvfSub;
// GRT_OPT_KERNEL values: 0 auto (tile kernel where it applies, else streaming), 1 per-lane, 2 round-based wave,
// 3 streaming, 4 big-window streaming (testing), 5 tile
constexpr int GRT_KERNEL_MAX = 5;
// true when the launch runs on a wave-per-tile kernel (streaming or tile kernel; alone, or as stage 2 of the mesh
// wavefront pipeline): its scheduling units are 8x8 tiles (4 per 16x16 block).  ONE predicate for do_launch (sizes
// order[] / cost[]) and launch_render (picks the kernel), so the two can never disagree about the unit of order[].
inline bool uses_stream_kernel(int variant, uint32_t mode, uint32_t stack_depth)
{
    return variant != 1 && variant != 2 && mode != 2 && stack_depth <= 120u;
}
// the tile kernel expands leaf ranges of <= 4 proxies, four lanes per range
// (n_prims < 2^26: the tile kernel addresses the 64-B records by a 32-bit byte offset, scalar loads with an SGPR offset)
constexpr uint32_t kTileMaxPrims = 1u << 26;
constexpr int kTileLeafMax = 4; // lanes per leaf range in a leaf step of the tile kernel = the largest leaf it takes (8: measured, mixed: r04 log item 22)
inline bool uses_tile_kernel(int variant, uint32_t mode, uint32_t stack_depth, int built_leaf_max, uint32_t n_prims)
{
    return uses_stream_kernel(variant, mode, stack_depth) && (variant == 0 || variant == 5) && built_leaf_max <= kTileLeafMax &&
           tile_stack_fits(stack_depth) && n_prims < kTileMaxPrims;
}
// per-tile cost map dilated by `radius` tiles (full-frame / window launches of the wave-per-tile kernels)
int dilate_unit_costs(const uint32_t* d_cost, uint32_t* d_out, uint32_t nbx, uint32_t nby, int radius, hipStream_t stream,
                      std::string* err);
// launch order with the heaviest tiles as 2 / 4 parts (tile kernel): n + extra_cap entries, padded with kOrderPad
// (d_scratch: order_scratch_bytes() of device memory, zeroed once; launches of multi_min units and more are ordered by several workgroups)
int order_units_with_parts(const uint32_t* d_cost_order, const uint32_t* d_cost_raw, uint32_t* d_order, uint32_t n, uint32_t extra_cap,
                           uint32_t pct2, uint32_t pct4, uint32_t pct_load, uint32_t resident_waves, uint32_t* d_zero, uint32_t* d_scratch,
                           uint32_t multi_min, uint32_t bag_classes, hipStream_t stream, std::string* err);
uint32_t order_scratch_bytes();
// the first kQuadListCap four-way part entries of a launch order (n_entries of them incl. padding), in order, listed in d_list and re-coded
// 2 -> 3 in the order itself; d_count[0] = how many
int quad_part_list(uint32_t* d_order, uint32_t n_entries, uint32_t* d_list, uint32_t* d_count, uint32_t cap, hipStream_t stream, std::string* err);
// launch order = units by cost class, heaviest first; also the number of heavy units when d_n_heavy != nullptr
// (d_zero != nullptr: that array of n cost words is zeroed once the order is made — the costs are consumed)
int order_units_by_cost(const uint32_t* d_cost, uint32_t* d_order, uint32_t n, uint32_t heavy_cap, uint32_t thr_x2,
                        uint32_t* d_n_heavy, uint32_t* d_zero, hipStream_t stream, std::string* err);

}  // namespace grt

// A context owns a SCENE (Gaussians, BVHs, records, meshes) and the per-frame state of ONE frame slot (eye records,
// scheduling feedback, overflow pool, wavefront queues, counters, events).  A VIEW (grt_create_view) is a context with
// frame-slot state of its own that renders its parent's scene: D frames in flight share one scene replica.
struct grt_ctx {
    grt_ctx* parent = nullptr; // view: the context whose scene this one renders
    int n_views = 0;           // live views of this context
    std::vector<grt_ctx*> views; // ... and which (a scene's frame slots look at each other's frame-end events: are frames in flight?)
    std::mutex views_mu;         // guards `views` / `n_views` and the siblings' frame-end events while they are looked at: views are contexts,
                                 // and grt.h lets distinct contexts be driven from distinct threads
    bool zombie = false;       // destroyed while views were alive: freed with the last of them
    uint64_t seen_epoch = 0;   // scene_epoch of the scene at this slot's last launch
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    int opt_counters = 0;
    int opt_kernel = 0;
    int opt_leaf_max = 4;
    int opt_swizzle = 2;
    int opt_tile_ready = 24, opt_tile_band = 64, opt_tile_look = 64, opt_tile_reserve = -1 /* auto: 16, trees with pieces 24 */, opt_tile_prio = 0; // band / look in 1/1024
    int opt_size_classes = 1;
    int opt_bvh_rotations = -1; // GRT_OPT_BVH_ROTATIONS
    int opt_band_abs = 512;       // GRT_OPT_TILE_BAND_ABS: that floor in 1/64 of the geometric-mean proxy diagonal
    float gm_diag = 0.f;          // geometric mean of the proxies' box diagonals (grt_build_bvh)
    int opt_split = -1;           // GRT_OPT_SPLIT: piece length of the spatial splits in quarters of the typical proxy diagonal (0 = off; < 0 = by the scene: 6 .. 16)
    int split_used = 0;           // ... the length the last build used
    int opt_split_vol_pct = 400;  // GRT_OPT_SPLIT_VOL_PCT (rounds 3-5: 50 — "only when the cells' boxes hold under half the proxy's box"; swept in round 6: 50 / 65 / 80 /
                                  // 100 / 200 / 400 % at sigma 1.0: 3.83 / 3.51 / 3.21 / 2.97 / 2.98 / 2.97 ms; sigma 1.6: 13.8 / . / . / 9.2 / 8.6 / 8.6)
    uint32_t n_hittable = 0;      // particles with opacity > alpha_min (BVH primitives = these, or their pieces)
    bool has_pieces = false;      // the current Gaussian BVH holds pieces of split proxies (the PIECES kernel instantiations)
    float4* d_ovf = nullptr;      // tile kernel: pool of window-overflow bags
    uint32_t* d_ovf_next = nullptr;
    uint32_t ovf_chunks = 0;
    int opt_ovf_chunks = 0;       // 0 = sized from demand; > 0: exactly this many chunks (testing); < 0: no pool at all
    int opt_ovf_entries = 0;      // 0 = kTileOvfEntries; testing: a smaller per-lane bag
    int opt_max_iters = 0;        // 0 = kTileMaxItersDefault; testing: a tiny watchdog
    uint32_t* h_ovf_used = nullptr; // pinned: chunks the last finished frame asked for (read back behind every frame)
    hipEvent_t ev_ovf = nullptr;
    bool ovf_pending = false;
    uint32_t ovf_demand = 0;      // the demand the pool follows: median of the last eight readings for the current launch geometry
    uint32_t ovf_hist[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // ... those readings (a ring: ovf_hist_n counts all of them)
    uint32_t ovf_hist_n = 0;
    bool ovf_stale = false;       // the reading on its way was asked for under another launch geometry
    bool ovf_short = false;       // an allocation of the size wanted failed
    uint32_t ovf_idle_run = 0;    // consecutive frames that found their stream idle (the application waits for its frames)
    bool ovf_sized = false;       // the pool in hand was made from a known demand (not the three chunks per tile of a first frame)
    bool order_classes = false;   // the entries of the launch order in d_order carry size classes (made by order_units_with_parts)
    uint32_t ovf_demand_max = 0;  // ... and the largest of those readings
    std::vector<std::pair<float4*, hipEvent_t>> ovf_old; // pools replaced while frames that may use them were in flight
    uint32_t ovf_units = 0;       // tiles of the launch the pool was last sized for
    uint32_t ovf_hint = 0, ovf_hint_units = 0; // scene context: the demand its frame slots have seen (a new slot starts from it)
    int opt_tile_parts2_pct = 0, opt_tile_parts4_pct = 60, opt_tile_parts_load_pct = 75; // GRT_OPT_TILE_PARTS2_PCT / _PARTS4_PCT / _PARTS_LOAD_PCT
    int opt_cold_parts_pct = 40;  // GRT_OPT_COLD_PARTS_PCT (with GRT_OPT_COLD_ESTIMATE = 2): four-way threshold of a cold frame, % of the heaviest estimate
    int opt_static_sharp = 1;     // GRT_OPT_STATIC_SHARP: a view that stood still for two frames is ordered by its tiles' own costs, not the dilated map
    int opt_order_multi_min = 16384; // GRT_OPT_ORDER_MULTI_MIN: launches of this many tiles and more are ordered by several workgroups
    int opt_mesh_parts = 1;       // GRT_OPT_MESH_PARTS: heavy tiles of a MESH frame's primary stage run as part waves too
    int opt_ovf_classes = 1;      // GRT_OPT_OVF_CLASSES: tiles take one to three chunks of the overflow pool by how deep their bags got (0: three)
    int opt_quad_parts = 1;       // GRT_OPT_QUAD_PARTS: four-way parts of camera-ray frames (no meshes, no pieces) run on the quad kernel
    bool parts_ok = false;        // this launch may run heavy tiles as parts (tile kernel, camera rays, no meshes)
    uint32_t order_launch = 0;    // entries of d_order when it holds parts (units + extra + padding); 0 = one entry per unit
    bool launch_order_matched = false; // the current launch's order had been made for this very frame (scene, camera, options)
    bool order_settled = false;        // ... and so had the frame d_order was made from: the order may be kept for repeats of it
    uint32_t* d_err = nullptr;    // sticky device error word (RenderArgs::err_word)
    uint32_t* h_err = nullptr;    // pinned copy of it, refreshed behind every frame on the frame's stream (read after ev_tail)
    hipStream_t tail_stream = nullptr; // stream the post-frame work (next order, zeroing) was queued on
    hipEvent_t ev_tail = nullptr;
    bool tail_pending = false;
    int built_leaf_max = 4; // leaf_max of the current Gaussian BVH (the tile kernel expands ranges of <= 4)
    // uploaded attributes (original order)
    uint64_t n = 0;
    float *d_pos = nullptr, *d_scale = nullptr, *d_quat = nullptr, *d_opacity = nullptr, *d_sh = nullptr;
    float4* d_color0 = nullptr;
    std::vector<float> h_opacity;
    // Gaussian BVH + records
    float alpha_min = 0.01f;
    grt::DevBvh gbvh;
    float4* d_rec = nullptr;
    float4* d_erec = nullptr;   // per-eye records of the streaming kernel (grt_api.hip: k_eye_records)
    float4* d_erec_wide = nullptr; // 64-B eye records of the tile kernel (k_eye_records_wide)
    size_t cap_erec_wide = 0, cap_erec = 0;
    bool erec_is_wide = false;
    float erec_eye[3] = {0, 0, 0};
    bool erec_valid = false;
    size_t cap_rec = 0;
    bool built = false;
    float build_ms = 0.f;
    float mesh_update_ms = 0.f; // device time of the last grt_set_meshes / grt_update_meshes
    // meshes
    grt::DevBvh mbvh;
    float4* d_tri = nullptr;
    uint32_t* d_faces = nullptr;
    float* d_vnormals = nullptr;
    uint32_t n_faces = 0, n_verts = 0;
    std::vector<uint32_t> mesh_nv, mesh_nf; // per mesh, as given to the last grt_set_meshes
    uint64_t faces_hash = 0;
    // instrumentation
    unsigned long long* d_counters = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool have_timing = false;
    // frame-to-frame scheduling feedback (grt_api.hip: do_launch)
    int opt_feedback = 1;
    uint32_t *d_cost = nullptr, *d_order = nullptr, *d_cost_dil = nullptr, *d_ord_scratch = nullptr;
    uint32_t *d_qparts = nullptr, *d_qpcount = nullptr; // the order's four-way parts as a list for the quad kernel (allocated with d_order)
    bool qparts_valid = false;                         // d_qparts was made from the order in d_order
    uint64_t qlist_epoch = 0, tail_qepoch = ~0ull;     // lists made so far; the one whose length the last frame tail copied to the host
    uint64_t qknown_epoch = ~0ull;                     // the list whose length the host has read ...
    uint32_t qknown_count = 0;                         // ... and that length
    int opt_cost_radius = 4; // tiles; 0 = off
    int opt_cold_estimate = 2; // order a frame without previous-frame costs by projected particle counts
    uint32_t cost_cap = 0;
    bool cost_valid = false;
    bool order_ready = false; // d_order already holds the order for the next frame with this geometry (do_launch, post-frame)
    bool order_split = false;
    bool order_valid = false; // d_order is a permutation of the units of cost_sig's launch geometry
    grt_params order_params{};   // the frame d_order was made from ...
    uint64_t order_epoch = 0;    // ... and the scene it showed (scene_epoch: bumped by every upload / build / mesh call)
    uint64_t scene_epoch = 1;
    bool cost_zeroed = false, ovf_zeroed = false; // d_cost / d_ovf_next were zeroed behind the last frame
    // wavefront buffers (allocated on first mesh frame)
    float4 *d_prec = nullptr, *d_queue = nullptr; // d_queue: two queues (ping-pong between the stages)
    uint32_t *d_bverdict = nullptr, *d_qunit = nullptr, *d_qskip = nullptr, *d_heavy_a = nullptr; // bundle verdicts (RenderArgs::bverdict)
    uint32_t bv_cap = 0;          // units d_bverdict holds
    uint64_t bv_sig[6] = {0, 0, 0, 0, 0, 0}; // launch geometry the verdicts belong to
    uint64_t bv_epoch = ~0ull;    // scene epoch they belong to
    uint32_t bv_view = 1;         // number of the current view (RenderArgs::bverdict_epoch)
    grt_params bv_params;         // view of the last mesh frame that used or gave verdicts
    bool bv_params_valid = false;
    uint32_t* d_qcount = nullptr;                 // one chunk counter per stage (kMaxBundleRounds + 1)
    int opt_bundle_rounds = 2; // bounce iterations traced by the wave-per-bundle kernel before the per-lane kernel finishes
    int opt_bundle_budget = 1400; // (round 6, with bundle verdicts: C4 2.55 ms at 896, 2.48 at 1100, 2.42 at 1300, 2.40 at 1400-1500, 2.46 at 1700; a glass sphere 4.8 / 4.4 / 4.3 / 4.4 at 896 / 1300 / 1400 / 1500 — the bundles run BESIDE the one-ray-per-wave kernel now, so a longer bundle costs nothing until it outlasts that kernel; round 3 without verdicts: 896, profiles/tools/tune_c4.py)
    int opt_lane_budget = 128;
    int opt_bundle_predict = 1; // GRT_OPT_BUNDLE_PREDICT
    int opt_mesh_primary_wave = 2; // GRT_OPT_MESH_PRIMARY_WAVE (C4: 2.40 / 2.30 / 2.24 ms at 0 / 1 / 2)
    int opt_single_look = 256, opt_single_band = 256; // 1/1024
    uint32_t* d_heavy = nullptr;
    float4* d_fqueue = nullptr;
    size_t wf_cap = 0;
    uint64_t cost_sig[6] = {0, 0, 0, 0, 0, 0};
    uint32_t* d_n_heavy = nullptr;
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int opt_heavy_thr_x2 = 4;   // heavy = cost > thr_x2/2 x median
    int opt_heavy_cap_div = 8;  // at most n_blocks / cap_div heavy blocks
    // big-window kernel for the heaviest blocks: 0 off, 1 on, 2 auto = on for small launches (<= 3072 blocks: a
    // multi-GPU rank's share of a 1080p frame), where the heaviest tile and not throughput bounds the frame
    int opt_heavy_split = 2;
};
