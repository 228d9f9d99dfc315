/*
 * grt_oracle.h — CPU ORACLE for the Gaussian ray-tracing hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (libgrt_hip.so) never
 * links, imports or calls anything in oracle/.
 *
 * It is a plain-C restatement of the reference's algorithm (file:line cited at each
 * function in grt_oracle.c; all citations are into Ray-Studio2/gaussian-ray-tracing):
 *   shaders/tracer.cu:17-187, shaders/tracer.cuh:68-73,115-496, src/Parameters.h:10-23,
 *   src/GaussianData.cpp:97-131, src/GaussianTracer.cpp:297-317,475-486,653-670,
 *   src/Camera.cpp:3-13, src/geometry/Icosahedron.h:13-37, src/vector_math.h:146,560-606,
 *   third_party/glm/gtc/quaternion.inl:47-72, third_party/glm/detail/type_mat3x3.inl:468-474.
 *
 * Parity pin status: the reference ships NO tests, fixtures or golden images.
 * The oracle is pinned by (a) oracle/_ref — the parts of the reference that compile here from
 * their own sources (Camera.cpp, Icosahedron.h, vector_math.h, glm) — bit-exact, (b) the
 * probe values recorded in SURVEY.md §8(c) from the reference's own device functions, and
 * (c) the analytic known-answer tests of SURVEY.md §4.2.  The OptiX traversal itself is a
 * closed driver component: that part of parity is UNPINNED and follows the semantic
 * decisions (i)-(ix) of SURVEY.md §8(c).
 */
#ifndef GRT_ORACLE_H
#define GRT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* activated particle (reference GaussianParticle, GaussianData.h:12-20); quat stored w,x,y,z */
typedef struct {
    float pos[3];
    float scale[3];
    float quat[4]; /* w, x, y, z (already normalised) */
    float opacity;
    float sh[16][3];
} grto_particle;

typedef struct {
    uint32_t width, height;
    uint32_t sh_degree_max;
    float eye[3], U[3], V[3], W[3];
    float t_min, t_max, min_transmittance, alpha_min;
    int32_t mode_fisheye;
    int32_t type;        /* MeshType: 0 MIRROR, 1 NORMAL, 2 GLASS (Parameters.h:78-83) */
    uint32_t max_bounces; /* reference constant MAX_BOUNCES = 32 (tracer.cuh:13) */
} grto_params;

typedef struct {
    uint64_t rays;            /* primary rays spawned */
    uint64_t segments;        /* Gaussian-trace segments (primary + secondary) */
    uint64_t hit_evals;       /* k-buffer entries consumed with T > minT (entry and exit both count) */
    uint64_t rounds;          /* traceGPs calls */
    uint64_t node_visits;     /* BVH nodes visited (oracle's own BVH) */
    uint64_t proxy_tests;     /* exact slab tests executed */
} grto_counters;

typedef struct grto_scene grto_scene;

/* ---- function-level restatements (each pinned by tests/test_oracle_*.py) ---- */
void  grto_activate(const float pos[3], const float f_dc[3], const float f_rest[45], float opacity_logit,
                    const float log_scale[3], const float rot[4], grto_particle* out);
void  grto_mat3_cast(const float quat_wxyz[4], float R_colmajor[9]);
void  grto_inv_cov(const grto_particle* p, float A_rowmajor[9]);
float grto_proxy_scale(float opacity, float alpha_min);
float grto_compute_response(const grto_particle* p, const float o[3], const float d[3]);
void  grto_compute_radiance(const grto_particle* p, const float d[3], uint32_t deg, float rgb[3]);
void  grto_get_ray(uint32_t ix, uint32_t iy, const float U[3], const float V[3], const float W[3],
                   const float eye[3], uint32_t width, uint32_t height, float o[3], float d[3]);
int   grto_get_fisheye_ray(uint32_t ix, uint32_t iy, const float U[3], const float V[3], const float W[3],
                           const float eye[3], uint32_t width, uint32_t height, float o[3], float d[3]);
void  grto_uvw_frame(const float eye[3], const float lookat[3], const float up[3], float fovy_deg, float aspect,
                     float U[3], float V[3], float W[3]);
uint8_t grto_quantize(float x);
void  grto_reflect(const float i[3], const float n[3], float out[3]);
/* returns 1 if refracted (t_hit gets +1e-5), 0 if total internal reflection (bounces++) */
int   grto_refract(const float ray_d[3], const float normal[3], float etai_over_etat, float out[3]);
/* exact proxy test; returns 1 on hit and writes entry/exit ray parameters */
int   grto_proxy_hit(const grto_particle* p, float alpha_min, const float o[3], const float d[3],
                     float* t_entry, float* t_exit);
void  grto_icosahedron(float verts[12][3], uint32_t idx[60]);
void  grto_slab_normals(float n[10][3]);
/* ray/triangle (closest-hit arithmetic shared with the HIP kernel); returns 1 on hit */
int   grto_tri_hit(const float v0[3], const float v1[3], const float v2[3], const float o[3], const float d[3],
                   float* t, float* u, float* v);

/* ---- scene + frame level ---- */
grto_scene* grto_scene_create(const grto_particle* particles, uint64_t n, float alpha_min);
void        grto_scene_destroy(grto_scene* s);
/* world-space triangles: verts[nv][3], normals[nv][3] (already multiplied by mat3(transform),
 * GaussianTracer.cpp:659-662), faces[nf][3]; mesh_of_face[nf] kept only for tie order */
void        grto_scene_set_mesh(grto_scene* s, const float* verts, const float* normals, uint32_t nv,
                                const uint32_t* faces, uint32_t nf);
void        grto_scene_use_bvh(grto_scene* s, int use_bvh); /* 0 = brute force over all proxies */
/* checking mode for SURVEY 8(c) decision (v): every proxy test intersects the 20 triangles of the particle's instanced icosahedron
 * (world-space vertices [n][12][3] by original particle id + 60 indices, supplied by the caller: the tests take them from
 * oracle/_ref) instead of the ten slabs; NULL = slabs again */
void        grto_scene_set_proxy_triangles(grto_scene* s, const float* verts, const uint32_t idx[60]);

/* k nearest hits of one traversal (traceGPs + __anyhit__, tracer.cuh:289-326, tracer.cu:124-153) */
uint32_t grto_trace_gps(const grto_scene* s, const float o[3], const float d[3], float tmin, float tmax,
                        uint32_t ids[7], float ts[7]);
/* trace() (tracer.cuh:328-373): in/out density; writes radiance of this call */
void grto_trace(const grto_scene* s, const grto_params* prm, const float o[3], const float d[3], float t_min,
                float t_max, float* density_io, float radiance[3], grto_counters* c);
/* full raygen state machine for one pixel (tracer.cu:17-110) -> pre-clamp accumColor */
void grto_render_pixel(const grto_scene* s, const grto_params* prm, uint32_t ix, uint32_t iy, float rgb[3],
                       grto_counters* c);
/* window [x0,x1) x [y0,y1); out_u8/out_f32 are FULL-FRAME row-major buffers (either may be NULL) */
void grto_render(const grto_scene* s, const grto_params* prm, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                 uint8_t* out_u8, float* out_f32, grto_counters* c, int n_threads);
/* same, but for caller-supplied rays (ray-buffer mode): rays[n][6] = o,d ; out_f32[n][3] */
void grto_render_rays(const grto_scene* s, const grto_params* prm, const float* rays, uint64_t n, float* out_f32,
                      grto_counters* c, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
