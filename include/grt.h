/*
 * grt.h — C ABI of libgrt_hip.so, the MI355X-native replacement for the OptiX pipeline behind
 * GaussianTracer::render() (reference: Ray-Studio2/gaussian-ray-tracing).
 *
 * Plain C: pointers, sizes, PODs.  No HIP, torch, OptiX or C++ types in any signature
 * (a hipStream_t crosses as void*; device pointers cross as void* / typed pointers).
 * Every entry point returns GRT_OK (0) or a negative grt_status; grt_last_error(ctx) holds the
 * text (the reference throws std::runtime_error with call text, src/Exception.h:19-80 — the C++
 * facade gaussian-ray-tracing_amd/host/GaussianTracer.cpp converts codes back to exceptions).
 *
 * What each entry point replaces in the reference:
 *   grt_create / grt_destroy      createContext..createSBT + dtor          src/GaussianTracer.cpp:85-295, 54-70
 *   grt_create_view               (new) a second frame slot on the same scene: what D frames in flight need, where the
 *                                 reference has one stream and one frame at a time (src/GaussianTracer.cpp:504,537)
 *   grt_upload_gaussians          particle upload in initializeParams      src/GaussianTracer.cpp:491-502
 *   grt_build_bvh                 createGaussianParticlesBVH/createGAS/    src/GaussianTracer.cpp:297-317,
 *                                 buildAccelationStructure (OptiX, closed)   319-399, 422-473
 *   grt_set_meshes                createGAS+createIAS for primitives,      src/GaussianTracer.cpp:578-709
 *                                 sendGeometryAttributesToDevice
 *   grt_update_meshes             updateInstanceTransforms (refit, no rebuild) src/GaussianTracer.cpp:711-794
 *   grt_render                    render(): param upload + optixLaunch of  src/GaussianTracer.cpp:508-538,
 *                                 raygen/anyhit/closesthit/miss              shaders/tracer.cu:17-187
 *   grt_render_tiles              (new) screen-tile sharding for N GPUs    SURVEY.md §8(e)
 *   grt_assemble_tiles            (new) rank 0's un-permute of the gathered tiles   SURVEY.md §8(e)
 *   grt_render_rays               (new) ray-buffer input for parity tests  SURVEY.md §7 hard part 1
 *   grt_sync                      CUDA_SYNC_CHECK()                        src/GaussianTracer.cpp:537
 *   grt_host_*                    host-side pieces the facade shares with ctypes users:
 *                                 GaussianData::parse (src/GaussianData.cpp:25-132), Camera::UVWFrame
 *                                 (src/Camera.cpp:3-13), grt_host_primitive_* / grt_host_obj_* = Primitives
 *                                 (src/geometry/Primitives.cpp:6-216)
 *
 * Ownership: the library owns every device allocation it makes; the caller owns output buffers;
 * host input arrays are borrowed for the duration of the call only.  A context is bound to one
 * device and is not re-entrant; distinct contexts may be driven from distinct threads/processes.
 */
#ifndef GRT_H
#define GRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRT_API __attribute__((visibility("default")))

typedef struct grt_ctx grt_ctx;

typedef enum {
    GRT_OK = 0,
    GRT_ERR_INVALID = -1,   /* bad argument / call order */
    GRT_ERR_HIP = -2,       /* a HIP runtime call failed */
    GRT_ERR_NO_DEVICE = -3, /* no usable GPU: the product path has no CPU fallback */
    GRT_ERR_IO = -4,        /* file could not be read / parsed */
    GRT_ERR_LIMIT = -5      /* scene exceeds a built-in limit (e.g. BVH height) */
} grt_status;

/* enum MeshType — src/Parameters.h:78-83 */
enum { GRT_MIRROR = 0, GRT_NORMAL = 1, GRT_GLASS = 2 };

/* Activated Gaussians, one host array per attribute (reference AoS GaussianParticle,
 * src/GaussianData.h:12-20).  quat is (w,x,y,z), already normalised; scale and opacity are
 * already exp()/sigmoid()-activated exactly as src/GaussianData.cpp:97-131 does on the host. */
typedef struct {
    const float* pos;     /* [n][3]  */
    const float* scale;   /* [n][3]  */
    const float* quat;    /* [n][4]  */
    const float* opacity; /* [n]     */
    const float* sh;      /* [n][16][3] : sh[k] = (f_rest_{k-1}, f_rest_{14+k}, f_rest_{29+k}), sh[0] = f_dc */
} grt_gaussians;

/* Triangle mesh already placed in world space by the caller (the facade applies
 * Primitive::transform): verts[nv][3]; normals[nv][3] pre-multiplied by mat3(transform)
 * (src/GaussianTracer.cpp:659-662); faces[nf][3]. */
typedef struct {
    const float* verts;
    const float* normals;
    uint32_t nv;
    const uint32_t* faces;
    uint32_t nf;
} grt_mesh;

/* struct Params — src/Parameters.h:42-74, minus the OptiX handles / device pointers the library
 * now owns (handle, d_particles, mesh_handle, d_meshes, traceState, output_buffer). */
typedef struct {
    uint32_t width, height;
    uint32_t sh_degree_max;
    float eye[3], U[3], V[3], W[3];
    float t_min, t_max, minTransmittance, alpha_min;
    int32_t mode_fisheye;
    int32_t type;         /* GRT_MIRROR / GRT_NORMAL / GRT_GLASS */
    uint32_t max_bounces; /* reference constant MAX_BOUNCES = 32, shaders/tracer.cuh:13 */
} grt_params;

typedef struct {
    uint64_t rays;        /* rays spawned (pixels; fisheye: r <= 1 only) */
    uint64_t segments;    /* Gaussian trace() calls (primary + secondary segments) */
    uint64_t hit_evals;   /* k-buffer entries consumed with T > minT (entry and exit both count) */
    uint64_t rounds;      /* k-buffer traversal rounds (traceGPs equivalents) */
    uint64_t node_visits; /* BVH node box tests by live rays (Gaussian BVH + mesh BVH) */
    uint64_t proxy_tests; /* exact icosahedron-slab tests executed */
    uint64_t rec_fetches; /* BVH node / proxy record bytes fetched by the wave-cooperative kernels, in 16-B units, at
                             the granularity they are loaded (one scalar load per wave): streaming kernel — a 4-wide
                             node = 8, a proxy record + its eye record = 5; round-based wave kernel — a 64-B record = 4 */
    uint64_t stall_exits; /* rays a wave-per-tile kernel (tile or streaming) gave up on with transmittance left: two passes in
                             a row composited nothing, or the step watchdog / stack guard fired (must be 0: a non-zero value
                             means a pixel is missing hits).  Counted with GRT_OPT_COUNTERS; the same events set the sticky
                             error word in EVERY kernel variant: grt_sync / grt_get_counters then return GRT_ERR_LIMIT */
} grt_counters;

typedef struct {
    uint64_t n_particles;   /* uploaded */
    uint64_t n_proxies;     /* hittable particles (opacity > alpha_min) */
    uint32_t n_nodes;       /* internal nodes of the Gaussian LBVH */
    uint32_t height;        /* LBVH height (levels of internal nodes) */
    uint32_t mesh_faces;
    uint32_t mesh_height;
    float build_ms;         /* device time of the last grt_build_bvh */
    float mesh_update_ms;   /* device time of the last grt_set_meshes (build) / grt_update_meshes (refit) */
    float scene_lo[3], scene_hi[3];
    uint64_t n_primitives;  /* leaves of the Gaussian LBVH: the hittable particles, large anisotropic ones as several pieces
                               (GRT_OPT_SPLIT) */
} grt_bvh_info;

typedef struct {
    uint64_t scene_bytes;         /* device memory of the scene this context renders (shared by a context and its views) */
    uint64_t slot_bytes;          /* device memory of this frame slot: eye records, overflow pool, feedback, wavefront queues */
    uint64_t overflow_pool_bytes; /* of which: the tile kernel's pool of window-overflow bags */
    uint32_t overflow_chunks;     /* chunks (32 KiB: 32 entries x 64 rays; a tile takes up to three) in that pool */
    uint32_t overflow_demand;     /* the demand the pool follows: median of what the last eight frames read back asked for */
} grt_memory_info;

enum { GRT_OPT_COUNTERS = 1 /* 1: use the instrumented kernel and fill grt_counters */,
       GRT_OPT_KERNEL = 2   /* 0 = auto: camera-ray frames on the tile kernel (grt_render_tile.hip: BVH culling per child box
                               against the tile frustum; also stage 2 of the wavefront pipeline of mesh frames) — or on the
                               streaming kernel when the BVH was built with GRT_OPT_LEAF_MAX > 4; ray buffers on the per-lane kernel.
                               1 = per-lane kernel everywhere, 2 = round-based wave kernel for camera rays without meshes
                               (per-lane otherwise), 3 = streaming kernel, 4 = its 32-slot variant everywhere (testing),
                               5 = same as 0 */,
       GRT_OPT_LEAF_MAX = 3 /* max primitives per BVH leaf, 1..8 (default 4); applies to the next build */,
       GRT_OPT_SWIZZLE = 4  /* XCD-aware launch order: runs of value 16x16 screen blocks (4 x value 8x8 tiles of the
                               streaming kernel) go to one XCD, i.e. one L2 (0 = identity; default 2) */,
       GRT_OPT_FEEDBACK = 5 /* 1 (default): launch the scheduling units (8x8 tiles for the streaming kernel, 16x16 blocks
                               for the others) heaviest-first using the previous frame's per-unit cost; in launches of
                               <= 3072 blocks (a multi-GPU rank's share of a frame) additionally run the heaviest tiles on
                               the 32-slot big-window kernel on a second stream (shorter critical path).
                               3: heaviest-first only.  5: big-window split always.  0: off */,
       GRT_OPT_HEAVY_THRESHOLD_X2 = 6 /* a unit is heavy when its cost exceeds value/2 x the median cost (default 4) */,
       GRT_OPT_HEAVY_CAP_DIV = 7      /* at most 1/value of the units go to the big-window kernel (default 8) */,
       /* tile kernel (GRT_OPT_KERNEL = 5) tuning; pixels never depend on these */
       GRT_OPT_TILE_READY_MIN = 8     /* lanes that must hold a final event for a compositing sweep to start or go on (fewer when few lanes still want anything), 1..64 (24) */,
       GRT_OPT_TILE_BAND = 9          /* particles within value/1024 of the front distance are tested as one batch (64) */,
       GRT_OPT_TILE_LOOKAHEAD = 10    /* nodes within value/1024 of the nearest node's distance are expanded together (64) */,
       GRT_OPT_TILE_RESERVE = 11      /* with fewer than value free frontier slots the nearest leaf ranges are tested first (-1 = default: 16, trees with pieces 24) */,
       GRT_OPT_TILE_PRIO_DIV = 12     /* the heaviest 1/value of the tiles (by last frame's cost) run at raised wave priority; 0 = off */,
       GRT_OPT_COST_RADIUS = 13       /* scheduling feedback under a moving camera: a tile's cost is the largest of last frame's costs
                                         within value tiles of it (default 4; 0 = the tile's own cost) */,
       GRT_OPT_SIZE_CLASSES = 14      /* 1 (default): proxies much larger than average get subtrees of their own in the Gaussian LBVH
                                         (size class in the top Morton bits); 0: plain Morton order.  Per context; next build */,
       GRT_OPT_COLD_ESTIMATE = 15     /* 1: a frame with no previous-frame costs (first frame, new size) launches its tiles in the
                                         order of the number of particle centres projecting into them; 2 (default): and the tiles whose
                                         estimate exceeds GRT_OPT_COLD_PARTS_PCT % of the largest (and the load condition of the part
                                         waves) run as four part waves already in that frame; 0: screen order */,
       GRT_OPT_BUNDLE_ROUNDS = 16     /* mesh frames on the tile kernel: how many bounce iterations trace their Gaussian segment wave-
                                         cooperatively (the bounced rays of an 8x8 tile as one bundle) before the per-lane kernel
                                         finishes whatever still bounces; 0..4, default 2.  Same image for every value */,
       GRT_OPT_BUNDLE_BUDGET = 17     /* work (steps + particles fetched + 2 x exact tests) a bundle may take before it is given up and its
                                         rays are traced one per wave (a bundle whose rays have spread too far to share work); default
                                         896, doubled for GRT_GLASS.  Same image for every value */,
       GRT_OPT_SINGLE_LOOKAHEAD = 18  /* one-ray-per-wave mode: as GRT_OPT_TILE_LOOKAHEAD (default 256 = 25 %) */,
       GRT_OPT_SINGLE_BAND = 19       /* one-ray-per-wave mode: as GRT_OPT_TILE_BAND (default 256 = 25 %) */,
       GRT_OPT_SPLIT = 24             /* spatial splits: a proxy much longer than the typical one whose world box is mostly empty (a needle or
                                         sheet that is not axis-aligned) enters the LBVH as up to 512 pieces, each with the box of its cell;
                                         value = piece length in quarters of the geometric-mean proxy diagonal (8 = 2 x; 0 = off).  Default -1:
                                         the length follows the scene, by the primitives per proxy that cutting at 8 would give: under 1.02 no
                                         pieces at all (the benchmark scenes), under 1.25 -> 6 (mildly anisotropic proxies, a few times longer
                                         than thick: a trained scene), under 1.5 -> 8, under 1.7 -> 10, under 2.2 -> 12, else 16 (scene-sized
                                         needles and sheets, where every piece re-tests its particle).  Pure acceleration structure: same hits,
                                         same pixels.  Per context; next build */,
       GRT_OPT_TILE_BAND_ABS = 25     /* trees with pieces: absolute floor of the tile kernel's leaf band and node look-ahead, in 1/64 of the
                                         geometric-mean proxy diagonal (default 512 = 8 x; 0 = relative bands only).  Scheduling only */,
       /* testing knobs (frames never change; speed and the failure signal do) */
       GRT_OPT_OVF_CHUNKS = 21        /* tile kernel's pool of window-overflow bags: 0 (default) = sized from the demand of the
                                         frames before (1.25 x the median of eight); n > 0: exactly n chunks of 32 KiB; < 0: no pool (every overflow costs another pass) */,
       GRT_OPT_OVF_ENTRIES = 22       /* per-lane capacity of a bag actually used, 1..96 (0 = default 96) */,
       GRT_OPT_MAX_ITERS = 23         /* step watchdog of the tile kernel (0 = default 2^21): a tile over it gives up on its rays
                                         and sets the sticky error word */,
       GRT_OPT_LANE_BUDGET = 20       /* whatever still bounces after the bundle rounds finishes on the per-lane traversal; a Gaussian
                                         segment over this many iterations there sends its ray to the one-ray-per-wave mode, which
                                         finishes it (default 128).  Same image for every value */,
       GRT_OPT_TILE_PARTS4_PCT = 27   /* tile kernel, camera rays (mesh frames: their primary stage, GRT_OPT_MESH_PARTS): an 8x8 tile whose cost in the previous frame exceeded
                                         value % of the heaviest tile's — and GRT_OPT_TILE_PARTS_LOAD_PCT % of the launch's total cost per
                                         resident wave — is launched as FOUR waves of 4x4 pixels (a quarter of the rays each, a narrower
                                         frustum).  A frame takes at least its longest tile; a frame bound by its total work (1080p on
                                         one GPU) splits nothing.  Default 60; 0 = never.  Pixels never depend on it */,
       GRT_OPT_TILE_PARTS2_PCT = 26   /* ... above value % of the heaviest (and below the four-way threshold): TWO waves of 4x8 pixels.
                                         Default 0 = never (half a heavy tile takes as long as the whole) */,
       GRT_OPT_TILE_PARTS_LOAD_PCT = 28, /* see GRT_OPT_TILE_PARTS4_PCT (default 75; 0 = no such condition) */
       GRT_OPT_MESH_PARTS = 29          /* 1 (default): the part waves also split the heavy tiles of a MESH frame's primary stage
                                           (each part queues its own chunk of continuation rays); 0: camera-ray frames without
                                           meshes only, as before round 4's last change */,
       GRT_OPT_ORDER_MULTI_MIN = 30     /* (testing) launches of value tiles and more have their launch order made by several workgroups in
                                           four short kernels instead of one workgroup (default 16384: from 1080p on; behind every frame of
                                           a moving camera: 54 -> ~25 us at 1080p, 249 -> ~30 us at 4K).  Same order either way */,
       GRT_OPT_STATIC_SHARP = 31        /* 1 (default): once a view has stood still for two frames its launch order is made from the tiles' own
                                           costs instead of the map dilated by GRT_OPT_COST_RADIUS (which is for a camera that moves); 0: always
                                           dilated, as before */,
       GRT_OPT_COLD_PARTS_PCT = 32,     /* see GRT_OPT_COLD_ESTIMATE (default 40) */
       GRT_OPT_QUAD_PARTS = 33          /* 1 (default): the four-way parts of camera-ray frames without meshes or pieces run on the QUAD kernel
                                           (one 4x4 quadrant per wave, lanes = rays x slots: four survivors of a leaf step are slab-tested
                                           at once, one per slot, a ray's pending events are the pool of its four windows), launched beside
                                           the camera-ray kernel — in launches that their parts bound (up to 12 288 tiles: a small frame, a
                                           rank's share of a frame), with a four-way threshold scaled down to half for launches of one tile per resident
                                           wave and fewer; 2: whatever the size (testing); > 2: and at most that many parts (testing); 0: part
                                           waves of the camera-ray kernel with 16 of 64 lanes in use (round 4).  Same pixels either way */,
       GRT_OPT_OVF_CLASSES = 34         /* 1 (default): a tile STARTS in one, two or three 32-entry chunks of the overflow pool by how deep
                                           its bags got in the frame before (one when nothing is known of it) and moves to three fresh
                                           ones when it outgrows them (the frame slot's memory: 1.30 -> 0.63 GB on the 1080p benchmark
                                           frame under a standing camera); 0: three for every tile that overflows (round 4).  Same
                                           pixels either way */,
       GRT_OPT_BVH_ROTATIONS = 35       /* applies to the next build of the Gaussian BVH (refused on a view): bottom-up sweeps of tree rotations
                                           behind the LBVH build — what the reference asks OptiX for with PREFER_FAST_TRACE,
                                           src/GaussianTracer.cpp:360.  -1 (default): one sweep for trees that hold pieces of split proxies,
                                           none otherwise; 0: none; n (<= 8): n sweeps on any tree.  Levels and the reported height
                                           (grt_bvh_info) are re-derived behind every sweep.  Culling structure only: same pixels */,
       GRT_OPT_BUNDLE_PREDICT = 36      /* mesh frames on the tile kernel.  1 (default): an 8x8 tile whose bounced rays gave up as a bundle
                                           (GRT_OPT_BUNDLE_BUDGET) is remembered FOR THE VIEW it happened under; while that view stands, its
                                           continuation rays go one per wave at once, on a list that the one-ray-per-wave kernel works off
                                           BESIDE the bundle kernel (second stream) instead of behind it, and the bundle that would be thrown
                                           away is not run.  A frame with other parameters (a camera that moves), another scene, frame
                                           geometry or budget uses no verdicts: every tile is tried as a bundle, as in round 5.  0: always
                                           so.  Same rays through the same two kernels: same pixels */,
       GRT_OPT_MESH_PRIMARY_WAVE = 37   /* mesh frames, stage 1 (camera ray -> closest mesh hit -> closest-hit shading: traceMesh of
                                           shaders/tracer.cuh:266-287, shaders/tracer.cu:112-122,155-187).  0: every lane walks the mesh
                                           tree alone, in a kernel of its own in front of the Gaussian stage (rounds 1-5); 1: the 64 rays
                                           of an 8x8 tile walk it TOGETHER (nodes and triangles by scalar loads, a child is entered when
                                           any lane wants it, one stack per wave), still a kernel of its own; 2 (default): that walk runs
                                           at the head of the tile kernel's primary stage — no launch, no 48-B record per pixel (other
                                           pipelines, and mesh trees too deep for the tile kernel's stack: as 1).  The same hit records bit for
                                           bit */,
       GRT_OPT_SPLIT_VOL_PCT = 38       /* spatial splits: a proxy longer than the piece length is cut when the boxes of its cells together hold
                                           less than value % of its own box's volume (default 400: practically always; rounds 3-5: 50).  Per context;
                                           next build.  Same pixels */ };

/* ---- context ---- */
GRT_API int grt_create(grt_ctx** out, int device);
/* A view: a frame slot of its own (stream, eye records, scheduling feedback, overflow pool, queues, counters, error
 * word) that renders `scene`'s Gaussians, BVHs and meshes.  Scene calls (grt_upload_gaussians, grt_build_bvh,
 * grt_set_meshes, grt_update_meshes, GRT_OPT_LEAF_MAX / GRT_OPT_SIZE_CLASSES) are refused on a view.  Destroying a
 * scene with live views is deferred until the last view is destroyed. */
GRT_API int grt_create_view(grt_ctx* scene, grt_ctx** out);
GRT_API void grt_destroy(grt_ctx* ctx);
GRT_API const char* grt_last_error(const grt_ctx* ctx); /* ctx may be NULL: last error of grt_create */
GRT_API int grt_set_option(grt_ctx* ctx, int option, int value);

/* ---- scene ---- */
GRT_API int grt_upload_gaussians(grt_ctx* ctx, const grt_gaussians* host, uint64_t n);
GRT_API int grt_build_bvh(grt_ctx* ctx, float alpha_min);
GRT_API int grt_set_meshes(grt_ctx* ctx, const grt_mesh* meshes, uint32_t n_meshes);
/* The same meshes, moved: new positions / normals for the topology of the last grt_set_meshes (same nv, nf per
 * mesh).  The mesh LBVH keeps its hierarchy and re-fits its boxes (reference: updateInstanceTransforms rebuilds GAS +
 * IAS on every gizmo frame and leaks the old ones, src/GaussianTracer.cpp:711-794).  GRT_ERR_INVALID when counts differ. */
GRT_API int grt_update_meshes(grt_ctx* ctx, const grt_mesh* meshes, uint32_t n_meshes);
GRT_API int grt_get_bvh_info(const grt_ctx* ctx, grt_bvh_info* out);
GRT_API int grt_get_memory_info(const grt_ctx* ctx, grt_memory_info* out);
/* (testing) depth of the Gaussian LBVH walked on the host over a copy of its node records, in levels of internal nodes: must not
 * exceed grt_bvh_info::height, which sizes the kernels' traversal stacks (synchronises the device; ~0.1 s per million nodes). */
GRT_API int grt_debug_bvh_depth(grt_ctx* ctx, uint32_t* out_depth);

/* ---- render (all asynchronous on `stream`, a hipStream_t; NULL = the context's own stream) ----
 * d_rgb8 : device uchar3 frame, row-major y*width+x (shaders/tracer.cuh:484-496), may be NULL
 * d_rgbf : device float3 frame holding accumColor before clamp/quantise, may be NULL
 * Window [x0,x1) x [y0,y1) restricts which pixels are traced and written (0,0,width,height = all). */
GRT_API int grt_render(grt_ctx* ctx, const grt_params* p, uint8_t* d_rgb8, float* d_rgbf, uint32_t x0, uint32_t y0,
                       uint32_t x1, uint32_t y1, void* stream);
/* Tiles are numbered row-major over the ceil(width/tile_w) x ceil(height/tile_h) grid.  Renders
 * tiles first_tile + j*tile_stride (j = 0..n_tiles-1) into COMPACT buffers [j][tile_h][tile_w][3];
 * pixels of a border tile that fall outside the frame are written as 0. */
GRT_API int grt_render_tiles(grt_ctx* ctx, const grt_params* p, uint8_t* d_rgb8, float* d_rgbf, uint32_t tile_w,
                             uint32_t tile_h, uint32_t first_tile, uint32_t tile_stride, uint32_t n_tiles,
                             void* stream);
/* Rank 0 of an N-rank frame ("rank 0 un-permutes tiles with a trivial copy kernel"): d_gathered holds the
 * ranks' compact buffers back to back, [world][max_cnt][tile_h][tile_w][3] (tile t of the grid = tile t / world of rank
 * t % world, as grt_render_tiles with first_tile = rank, tile_stride = world writes them); d_rgb8 receives the frame. */
GRT_API int grt_assemble_tiles(grt_ctx* ctx, const uint8_t* d_gathered, uint32_t world, uint32_t max_cnt, uint32_t tile_w,
                               uint32_t tile_h, uint32_t width, uint32_t height, uint8_t* d_rgb8, void* stream);
/* d_rays[n][6] = origin, direction (device); d_rgbf[n][3] */
GRT_API int grt_render_rays(grt_ctx* ctx, const grt_params* p, const float* d_rays, uint64_t n, float* d_rgbf,
                            void* stream);
/* Waits for the context's stream and the last frame launched through this context (whatever stream it went to), then
 * reads the sticky device error word: GRT_ERR_LIMIT (text in grt_last_error, word cleared) when a wave had to give up on
 * live rays since the last check — the reference throws on traversal trouble (src/Exception.h:31-80). */
GRT_API int grt_sync(grt_ctx* ctx);
GRT_API int grt_get_counters(grt_ctx* ctx, grt_counters* out); /* syncs; counters of the last render; error word as grt_sync */
/* device time (ms, HIP events on the launch stream) of the last render's kernel; syncs */
GRT_API int grt_last_kernel_ms(grt_ctx* ctx, float* ms);

/* ---- host helpers (no GPU needed) ---- */
/* Raw 3DGS PLY columns -> activated attributes (src/GaussianData.cpp:97-131).  f_rest is [n][45]. */
GRT_API int grt_host_activate(uint64_t n, const float* pos, const float* f_dc, const float* f_rest,
                              const float* opacity_logit, const float* log_scale, const float* rot, float* out_pos,
                              float* out_scale, float* out_quat, float* out_opacity, float* out_sh);
/* Camera::UVWFrame (src/Camera.cpp:3-13) */
GRT_API void grt_host_uvw_frame(const float eye[3], const float lookat[3], const float up[3], float fovy_deg,
                                float aspect, float U[3], float V[3], float W[3]);
/* Deterministic synthetic 3DGS scene: fills raw PLY columns. */
GRT_API int grt_host_synth_scene(uint64_t seed, uint64_t n, float* pos, float* f_dc, float* f_rest,
                                 float* opacity_logit, float* log_scale, float* rot);
/* 3DGS PLY (binary little-endian or ascii; float properties looked up by name as
 * src/GaussianData.cpp:27-92 does).  Two-call pattern: n_out only, then fill. */
GRT_API int grt_host_ply_count(const char* path, uint64_t* n_out);
GRT_API int grt_host_ply_read(const char* path, uint64_t n, float* pos, float* f_dc, float* f_rest,
                              float* opacity_logit, float* log_scale, float* rot);
GRT_API int grt_host_ply_write(const char* path, uint64_t n, const float* pos, const float* f_dc, const float* f_rest,
                               const float* opacity_logit, const float* log_scale, const float* rot);
/* Procedural primitives of the reference (src/geometry/Primitives.cpp:6-140) at the origin: verts[nv][3],
 * normals[nv][3], faces[nf][3].  Two-call pattern: counts, then fill. */
enum { GRT_PRIM_PLANE = 0, GRT_PRIM_SPHERE = 1 };
GRT_API int grt_host_primitive_counts(int kind, uint32_t* nv, uint32_t* nf);
GRT_API int grt_host_primitive_fill(int kind, float* verts, float* normals, uint32_t* faces);
/* OBJ -> un-indexed triangle soup (one vertex per face corner, nf = nv / 3) with the reference's Y flip of positions
 * and normals (src/geometry/Primitives.cpp:142-202).  faces is [nv] = 0..nv-1.  grt_host_obj_write writes
 * "f a//a b//b c//c" with %.9g coordinates (fp32 round-trips). */
GRT_API int grt_host_obj_count(const char* path, uint32_t* nv, uint32_t* nf);
GRT_API int grt_host_obj_read(const char* path, uint32_t nv, float* verts, float* normals, uint32_t* faces);
GRT_API int grt_host_obj_write(const char* path, uint32_t nv, const float* verts, const float* normals, uint32_t nf,
                               const uint32_t* faces);
GRT_API const char* grt_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* GRT_H */
