/*
 * grt_oracle.c — CPU ORACLE (test infrastructure only; see grt_oracle.h header comment).
 *
 * Plain-C, fp32, IEEE (+,-,*,/,sqrt,fma) restatement of the reference's Gaussian ray tracer.
 * Build with -ffp-contract=off: every multiply/add below rounds exactly where the reference's
 * source expression rounds; explicit fmaf() is used only inside the proxy test, whose
 * arithmetic the reference leaves to the (closed) OptiX triangle intersector.
 * All file:line citations are into Ray-Studio2/gaussian-ray-tracing.
 */
#include "grt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* vector helpers: src/vector_math.h:146 (clamp), :572-575 (dot), :578-581 (cross),           */
/* :584-587 (length), :590-594 (normalize), :603-606 (reflect)                                */
/* ------------------------------------------------------------------------------------------ */
typedef struct { float x, y, z; } f3;

static inline f3 mk3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
static inline void st3(float* p, f3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
static inline f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 mul3s(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
static inline f3 neg3(f3 a) { return mk3(-a.x, -a.y, -a.z); }
static inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline f3 cross3(f3 a, f3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float length3(f3 v) { return sqrtf(dot3(v, v)); }
static inline f3 normalize3(f3 v) {
    float invLen = 1.0f / sqrtf(dot3(v, v));
    return mul3s(v, invLen);
}
static inline float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }
static inline f3 reflect3(f3 i, f3 n) { /* i - 2.0f * n * dot(n, i) : ((2*n)*dot) */
    f3 n2 = mul3s(n, 2.0f);
    return sub3(i, mul3s(n2, dot3(n, i)));
}

void grto_reflect(const float i[3], const float n[3], float out[3]) { st3(out, reflect3(ld3(i), ld3(n))); }

/* SH constants: src/Parameters.h:10-23 */
#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f
#define SH_C2_0 1.0925484305920792f
#define SH_C2_1 -1.0925484305920792f
#define SH_C2_2 0.31539156525252005f
#define SH_C2_3 -1.0925484305920792f
#define SH_C2_4 0.5462742152960396f
#define SH_C3_0 -0.5900435899266435f
#define SH_C3_1 2.890611442640554f
#define SH_C3_2 -0.4570457994644658f
#define SH_C3_3 0.3731763325901154f
#define SH_C3_4 -0.4570457994644658f
#define SH_C3_5 1.445305721320277f
#define SH_C3_6 -0.5900435899266435f

/* tracer.cuh:9-14 */
#define TRACE_MESH_TMIN 1e-5f
#define TRACE_MESH_TMAX 1e5f
#define MAX_HITS_PER_TRACE 7
#define TIMEOUT_ITERATIONS 1000u
#define REFRACTION_EPS_SHIFT 1e-5f

/* ------------------------------------------------------------------------------------------ */
/* a1: activations — src/GaussianData.cpp:97-128                                              */
/* ------------------------------------------------------------------------------------------ */
void grto_activate(const float pos[3], const float f_dc[3], const float f_rest[45], float opacity_logit,
                   const float log_scale[3], const float rot[4], grto_particle* out)
{
    out->pos[0] = pos[0]; out->pos[1] = pos[1]; out->pos[2] = pos[2];
    out->scale[0] = expf(log_scale[0]);                      /* :101-103 */
    out->scale[1] = expf(log_scale[1]);
    out->scale[2] = expf(log_scale[2]);
    const float norm = sqrtf(rot[0] * rot[0] + rot[1] * rot[1] + rot[2] * rot[2] + rot[3] * rot[3]); /* :104-107 */
    out->quat[0] = rot[0] / norm;  /* glm::quat(w,x,y,z) ctor, :108-111: rot_0 is w */
    out->quat[1] = rot[1] / norm;
    out->quat[2] = rot[2] / norm;
    out->quat[3] = rot[3] / norm;
    out->opacity = 1.0f / (1.0f + expf(-opacity_logit));     /* :112 */
    out->sh[0][0] = f_dc[0]; out->sh[0][1] = f_dc[1]; out->sh[0][2] = f_dc[2]; /* :113 */
    for (int k = 1; k < 16; k++) {                           /* :114-128: sh[k] = (f_rest[k-1], f_rest[14+k], f_rest[29+k]) */
        out->sh[k][0] = f_rest[k - 1];
        out->sh[k][1] = f_rest[14 + k];
        out->sh[k][2] = f_rest[29 + k];
    }
}

/* glm::mat3_cast — third_party/glm/gtc/quaternion.inl:47-72.  Output column-major Rg[c*3+r]. */
void grto_mat3_cast(const float q[4], float Rg[9])
{
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    const float qxx = x * x, qyy = y * y, qzz = z * z;
    const float qxz = x * z, qxy = x * y, qyz = y * z;
    const float qwx = w * x, qwy = w * y, qwz = w * z;
    Rg[0] = 1.0f - 2.0f * (qyy + qzz);
    Rg[1] = 2.0f * (qxy + qwz);
    Rg[2] = 2.0f * (qxz - qwy);
    Rg[3] = 2.0f * (qxy - qwz);
    Rg[4] = 1.0f - 2.0f * (qxx + qzz);
    Rg[5] = 2.0f * (qyz + qwx);
    Rg[6] = 2.0f * (qxz + qwy);
    Rg[7] = 2.0f * (qyz - qwx);
    Rg[8] = 1.0f - 2.0f * (qxx + qyy);
}

/* invCov = inv_s * transpose(R) — shaders/tracer.cuh:193-201.  With inv_s diagonal the glm
 * 3x3 product (type_mat3x3.inl:486-520) reduces to one exact multiply per element (the other
 * two addends are exact zeros):  A[r][c] = (1/scale_r) * R_math[c][r] = (1/scale_r) * Rg[r*3+c].
 * Output row-major A[r*3+c] so that (A v)_r = (A[r][0]*v.x + A[r][1]*v.y) + A[r][2]*v.z
 * (glm mat3*vec3, type_mat3x3.inl:468-474). */
void grto_inv_cov(const grto_particle* p, float A[9])
{
    float Rg[9];
    grto_mat3_cast(p->quat, Rg);
    for (int r = 0; r < 3; r++) {
        const float inv = 1.0f / p->scale[r];
        for (int c = 0; c < 3; c++) A[r * 3 + c] = inv * Rg[r * 3 + c];
    }
}

static inline f3 matvec(const float A[9], f3 v)
{
    return mk3(A[0] * v.x + A[1] * v.y + A[2] * v.z,
               A[3] * v.x + A[4] * v.y + A[5] * v.z,
               A[6] * v.x + A[7] * v.y + A[8] * v.z);
}

/* a3: proxy half-width — src/GaussianTracer.cpp:306 */
float grto_proxy_scale(float opacity, float alpha_min) { return sqrtf(2.0f * logf(opacity / alpha_min)); }

/* a13: computeResponse — shaders/tracer.cuh:187-214 */
static inline float response_from(const float A[9], f3 mu, f3 o, f3 d, f3 o_g, f3 d_g)
{
    const float d_val = -dot3(o_g, d_g) / fmaxf(1e-6f, dot3(d_g, d_g));
    const f3 pos = add3(o, mul3s(d, d_val));
    const f3 p_g = matvec(A, sub3(mu, pos));
    return expf(-0.5f * dot3(p_g, p_g));
}

float grto_compute_response(const grto_particle* p, const float o_[3], const float d_[3])
{
    float A[9];
    grto_inv_cov(p, A);
    const f3 mu = ld3(p->pos), o = ld3(o_), d = ld3(d_);
    const f3 o_g = matvec(A, sub3(o, mu));
    const f3 d_g = matvec(A, d);
    return response_from(A, mu, o, d, o_g, d_g);
}

/* a14: SHToRadiance + computeRadiance — shaders/tracer.cuh:216-264 (d is normalize(ray_d), :359) */
static inline f3 sh_to_radiance(const float sh[16][3], f3 d, uint32_t deg)
{
#define SHV(i) ld3(sh[i])
    f3 L = add3(mk3(0.5f, 0.5f, 0.5f), mul3s(SHV(0), SH_C0));
    if (deg == 0) return L;
    const float x = d.x, y = d.y, z = d.z;
    {
        f3 t = add3(mul3s(SHV(1), -y), mul3s(SHV(2), z));
        t = sub3(t, mul3s(SHV(3), x));
        L = add3(L, mul3s(t, SH_C1));
    }
    if (deg == 1) return L;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z;
    {
        /* "2. * zz - xx - yy" is evaluated in double and the product with SH_C2_2 rounded once
         * to float when it meets the float3 (tracer.cuh:243) */
        const float c22 = (float)((double)SH_C2_2 * (2. * (double)zz - (double)xx - (double)yy));
        f3 s = mul3s(SHV(4), SH_C2_0 * xy);
        s = add3(s, mul3s(SHV(5), SH_C2_1 * yz));
        s = add3(s, mul3s(SHV(6), c22));
        s = add3(s, mul3s(SHV(7), SH_C2_3 * xz));
        s = add3(s, mul3s(SHV(8), SH_C2_4 * (xx - yy)));
        L = add3(L, s);
    }
    if (deg == 2) return L;
    {
        f3 s = mul3s(SHV(9), (SH_C3_0 * y) * (3.0f * xx - yy));
        s = add3(s, mul3s(SHV(10), (SH_C3_1 * xy) * z));
        s = add3(s, mul3s(SHV(11), (SH_C3_2 * y) * (4.0f * zz - xx - yy)));
        s = add3(s, mul3s(SHV(12), (SH_C3_3 * z) * (2.0f * zz - 3.0f * xx - 3.0f * yy)));
        s = add3(s, mul3s(SHV(13), (SH_C3_4 * x) * (4.0f * zz - xx - yy)));
        s = add3(s, mul3s(SHV(14), (SH_C3_5 * z) * (xx - yy)));
        s = add3(s, mul3s(SHV(15), (SH_C3_6 * x) * (xx - 3.0f * yy)));
        L = add3(L, s);
    }
    return L;
#undef SHV
}

void grto_compute_radiance(const grto_particle* p, const float d[3], uint32_t deg, float rgb[3])
{
    f3 L = sh_to_radiance(p->sh, ld3(d), deg);
    rgb[0] = fmaxf(L.x, 0.0f); rgb[1] = fmaxf(L.y, 0.0f); rgb[2] = fmaxf(L.z, 0.0f); /* :263 */
}

/* a6: getRay — shaders/tracer.cuh:115-134; called with -U, -V, W (tracer.cu:35-45) by render_pixel */
void grto_get_ray(uint32_t ix, uint32_t iy, const float U[3], const float V[3], const float W[3],
                  const float eye[3], uint32_t width, uint32_t height, float o[3], float d[3])
{
    const float dx = 2.0f * (((float)ix + 0.5f) / (float)(int)width) - 1.0f;
    const float dy = 2.0f * (((float)iy + 0.5f) / (float)(int)height) - 1.0f;
    f3 dir = add3(add3(mul3s(ld3(U), dx), mul3s(ld3(V), dy)), ld3(W));
    st3(o, ld3(eye));
    st3(d, normalize3(dir));
}

/* a7: getFishEyeRay — shaders/tracer.cuh:136-165.  r > 1: the reference returns without writing
 * the ray (UB); decision (vii): no ray, black pixel -> return 0. */
int grto_get_fisheye_ray(uint32_t ix, uint32_t iy, const float U[3], const float V[3], const float W[3],
                         const float eye[3], uint32_t width, uint32_t height, float o[3], float d[3])
{
    const float dx = 2.0f * (((float)ix + 0.5f) / (float)(int)width) - 1.0f;
    const float dy = 2.0f * (((float)iy + 0.5f) / (float)(int)height) - 1.0f;
    const float r = sqrtf(dx * dx + dy * dy);
    if (r > 1.0f) return 0;
    const float f = 1.0f / sqrtf(2.0f);
    const float theta = 2.0f * asinf(r / (2.0f * f));
    const float phi = atan2f(dy, dx);
    const f3 dir = mk3(sinf(theta) * cosf(phi), sinf(theta) * sinf(phi), cosf(theta));
    f3 w = add3(add3(mul3s(ld3(U), dir.x), mul3s(ld3(V), dir.y)), mul3s(ld3(W), dir.z));
    st3(o, ld3(eye));
    st3(d, normalize3(w));
    return 1;
}

/* a5: Camera::UVWFrame — src/Camera.cpp:3-13 */
void grto_uvw_frame(const float eye[3], const float lookat[3], const float up[3], float fovy_deg, float aspect,
                    float U[3], float V[3], float W[3])
{
    f3 w = sub3(ld3(lookat), ld3(eye));
    float wlen = length3(w);
    f3 u = normalize3(cross3(w, ld3(up)));
    f3 v = normalize3(cross3(u, w));
    float vlen = wlen * tanf(0.5f * fovy_deg * 3.14159265358979323846f / 180.0f);
    v = mul3s(v, vlen);
    float ulen = vlen * aspect;
    u = mul3s(u, ulen);
    st3(U, u); st3(V, v); st3(W, w);
}

/* a15: quantizeUnsigned8Bits — shaders/tracer.cuh:68-73 */
uint8_t grto_quantize(float x)
{
    x = clampf(x, 0.0f, 1.0f);
    unsigned int q = (unsigned int)(x * 256.0f);
    return (uint8_t)(q < 255u ? q : 255u);
}

/* a10: refract — shaders/tracer.cuh:432-464 (renderGlass :466-482 supplies n2/n1 = 1.5f/1.0003f) */
int grto_refract(const float ray_d_[3], const float normal_[3], float etai_over_etat, float out[3])
{
    f3 ray_d = ld3(ray_d_), normal = ld3(normal_);
    float ri;
    if (dot3(ray_d, normal) < 0.0f) {
        ri = 1.0f / etai_over_etat;
    } else {
        ri = etai_over_etat;
        normal = neg3(normal);
    }
    float cos_theta = fminf(dot3(neg3(ray_d), normal), 1.0f);
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    int cannot_refract = ri * sin_theta > 1.0f;
    if (cannot_refract) {
        f3 rn = dot3(ray_d, normal) < 0.0f ? normal : neg3(normal);
        st3(out, reflect3(ray_d, rn));
        return 0;
    }
    f3 r_out_perp = mul3s(add3(ray_d, mul3s(normal, cos_theta)), ri);
    f3 r_out_parallel = mul3s(normal, -sqrtf(fabsf(1.0f - dot3(r_out_perp, r_out_perp))));
    st3(out, add3(r_out_perp, r_out_parallel));
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* a2: icosahedron proxy — src/geometry/Icosahedron.h:13-37                                    */
/* ------------------------------------------------------------------------------------------ */
void grto_icosahedron(float v[12][3], uint32_t idx[60])
{
    const float rr = (3 + sqrtf(5.0f)) / (2 * sqrtf(3.0f));
    const float ss = 1.0f / rr;
    const float tt = (1.0f + sqrtf(5.0f)) / (2.0f * rr);
    const float V[12][3] = {{-ss, tt, 0}, {ss, tt, 0}, {-ss, -tt, 0}, {ss, -tt, 0}, {0, -ss, tt}, {0, ss, tt},
                            {0, -ss, -tt}, {0, ss, -tt}, {tt, 0, -ss}, {tt, 0, ss}, {-tt, 0, -ss}, {-tt, 0, ss}};
    static const uint32_t I[60] = {0, 11, 5, 0, 5, 1, 0, 1, 7, 0, 7, 10, 0, 10, 11, 1, 5, 9, 5, 11, 4, 11, 10, 2,
                                   10, 7, 6, 7, 1, 8, 3, 9, 4, 3, 4, 2, 3, 2, 6, 3, 6, 8, 3, 8, 9, 4, 9, 5,
                                   2, 4, 11, 6, 2, 10, 8, 6, 7, 9, 8, 1};
    memcpy(v, V, sizeof(V));
    memcpy(idx, I, sizeof(I));
}

/* The 20 faces are 10 antipodal pairs at plane distance exactly 1 (circumscribes the unit
 * sphere): 6 normals of the form (0,P,+-Q) and cyclic shifts, 4 of the form (1,+-1,+-1)/sqrt(3),
 * P = phi/sqrt(3), Q = 1/(phi*sqrt(3)).  tests/test_oracle_units.py checks this table against
 * the reference mesh (oracle/_ref when built, grto_icosahedron otherwise). */
#define ICO_P 0.9341723322868347f
#define ICO_Q 0.35682210326194763f
#define ICO_K 0.5773502588272095f
#define ICO_SQRT3 1.7320508075688772f

void grto_slab_normals(float n[10][3])
{
    const float N[10][3] = {{0, ICO_P, -ICO_Q}, {0, ICO_P, ICO_Q}, {ICO_Q, 0, -ICO_P}, {ICO_Q, 0, ICO_P},
                            {ICO_K, ICO_K, -ICO_K}, {ICO_K, -ICO_K, -ICO_K}, {ICO_K, -ICO_K, ICO_K},
                            {ICO_K, ICO_K, ICO_K}, {ICO_P, -ICO_Q, 0}, {ICO_P, ICO_Q, 0}};
    memcpy(n, N, sizeof(N));
}

/* projections of v on the 10 slab normals; slabs 4..7 are left un-normalised (x+-y+-z) and are
 * compared against s*sqrt(3) instead of s.  Arithmetic is OURS (the reference delegates it to
 * OptiX) and is replicated bit-for-bit by the HIP kernel (csrc/grt_device.h: slab_project). */
static inline void slab_project(f3 v, float a[10])
{
    const float py = ICO_P * v.y, qz = ICO_Q * v.z;
    const float qx = ICO_Q * v.x, pz = ICO_P * v.z;
    const float px = ICO_P * v.x, qy = ICO_Q * v.y;
    const float xpy = v.x + v.y, xmy = v.x - v.y;
    a[0] = py - qz; a[1] = py + qz;
    a[2] = qx - pz; a[3] = qx + pz;
    a[4] = xpy - v.z; a[5] = xmy - v.z; a[6] = xmy + v.z; a[7] = xpy + v.z;
    a[8] = px - qy; a[9] = px + qy;
}

/* exact proxy test in Gaussian space: |n_i . (o_g + t d_g)| <= s for the 10 slabs (decision (v)).
 * Entry = max of the near parameters, exit = min of the far ones; the arg-max/arg-min is found
 * by cross-multiplication (no division), then ONE IEEE division each.  hit iff entry <= exit. */
static inline int proxy_slabs(f3 o_g, f3 d_g, float s, float* t_entry, float* t_exit)
{
    float a[10], b[10];
    slab_project(o_g, a);
    slab_project(d_g, b);
    const float s3 = s * ICO_SQRT3;
    float nn = 0.0f, nd = 0.0f, fn = 0.0f, fd = 0.0f;
    for (int i = 0; i < 10; i++) {
        const float h = (i >= 4 && i <= 7) ? s3 : s;
        /* a with the sign of b folded in, by the sign BIT of b (b = -0 reads as negative: that slab is parallel to the ray and
         * its two planes land at -+1e30 x something on either reading, so it bounds nothing; what matters is that the GPU
         * kernels do the very same thing, grt_device.h: proxy_slabs_pre) */
        const float ap = signbit(b[i]) ? -a[i] : a[i];
        const float bp = fmaxf(fabsf(b[i]), 1e-30f);
        const float cn = -(ap + h); /* near = cn / bp */
        const float cf = h - ap;    /* far  = cf / bp */
        if (i == 0) {
            nn = cn; nd = bp; fn = cf; fd = bp;
        } else {
            if (cn * nd > nn * bp) { nn = cn; nd = bp; }
            if (cf * fd < fn * bp) { fn = cf; fd = bp; }
        }
    }
    const float te = nn / nd, tx = fn / fd;
    *t_entry = te;
    *t_exit = tx;
    return te <= tx;
}

int grto_proxy_hit(const grto_particle* p, float alpha_min, const float o_[3], const float d_[3], float* t_entry,
                   float* t_exit)
{
    const float s = grto_proxy_scale(p->opacity, alpha_min);
    if (!(s > 0.0f)) return 0; /* decision (vi): opacity <= alpha_min => NaN/0 transform, unhittable */
    float A[9];
    grto_inv_cov(p, A);
    const f3 mu = ld3(p->pos), o = ld3(o_), d = ld3(d_);
    return proxy_slabs(matvec(A, sub3(o, mu)), matvec(A, d), s, t_entry, t_exit);
}

/* ray/triangle, Moeller-Trumbore, no culling (mesh GAS flags NONE, GaussianTracer.cpp:355-360);
 * barycentrics as OptiX reports them: hit = (1-u-v) v0 + u v1 + v v2. */
int grto_tri_hit(const float v0_[3], const float v1_[3], const float v2_[3], const float o_[3], const float d_[3],
                 float* t, float* u, float* v)
{
    const f3 v0 = ld3(v0_), e1 = sub3(ld3(v1_), v0), e2 = sub3(ld3(v2_), v0);
    const f3 o = ld3(o_), d = ld3(d_);
    const f3 p = cross3(d, e2);
    const float det = dot3(e1, p);
    if (det == 0.0f) return 0;
    const float inv = 1.0f / det;
    const f3 tv = sub3(o, v0);
    const float uu = dot3(tv, p) * inv;
    if (!(uu >= 0.0f && uu <= 1.0f)) return 0;
    const f3 q = cross3(tv, e1);
    const float vv = dot3(d, q) * inv;
    if (!(vv >= 0.0f && uu + vv <= 1.0f)) return 0;
    *t = dot3(e2, q) * inv;
    *u = uu;
    *v = vv;
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* scene: derived per-particle proxies + a conservative BVH (culling only, never decides hits) */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    float mu[3];
    float A[9];
    float s;
    float opacity;
    uint32_t id;
} proxy_t;

typedef struct {
    float lo[3], hi[3];
    uint32_t left, right; /* internal: child node indices */
    uint32_t first, count; /* leaf: range in prim order (count > 0) */
} node_t;

typedef struct {
    node_t* nodes;
    uint32_t n_nodes;
    uint32_t* prim; /* prim order */
} bvh_t;

struct grto_scene {
    uint64_t n, m;
    grto_particle* parts;
    proxy_t* prox;
    float alpha_min;
    int use_bvh;
    bvh_t gbvh;
    /* mesh (world space) */
    float* mv; float* mn; uint32_t nv;
    uint32_t* mf; uint32_t nf;
    bvh_t mbvh;
    /* brute-force proxy mode (grto_scene_set_proxy_triangles): world-space vertices of every particle's instanced
     * icosahedron [n][12][3], by ORIGINAL particle id, and the 20 faces' indices — supplied by the caller */
    float* ptv; uint32_t pti[60]; int tri_mode;
};

typedef struct { const float* lo; const float* hi; float* cen; uint32_t* prim; node_t* nodes; uint32_t n_nodes; } build_t;

static uint32_t build_rec(build_t* b, uint32_t first, uint32_t count)
{
    uint32_t me = b->n_nodes++;
    node_t* nd = &b->nodes[me];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = first; i < first + count; i++) {
        uint32_t p = b->prim[i];
        for (int k = 0; k < 3; k++) {
            lo[k] = fminf(lo[k], b->lo[p * 3 + k]);
            hi[k] = fmaxf(hi[k], b->hi[p * 3 + k]);
            clo[k] = fminf(clo[k], b->cen[p * 3 + k]);
            chi[k] = fmaxf(chi[k], b->cen[p * 3 + k]);
        }
    }
    memcpy(nd->lo, lo, sizeof(lo));
    memcpy(nd->hi, hi, sizeof(hi));
    if (count <= 4) {
        nd->first = first; nd->count = count; nd->left = nd->right = 0;
        return me;
    }
    int ax = 0;
    if (chi[1] - clo[1] > chi[ax] - clo[ax]) ax = 1;
    if (chi[2] - clo[2] > chi[ax] - clo[ax]) ax = 2;
    float mid = 0.5f * (clo[ax] + chi[ax]);
    uint32_t i = first, j = first + count;
    while (i < j) {
        if (b->cen[b->prim[i] * 3 + ax] < mid) i++;
        else { j--; uint32_t t = b->prim[i]; b->prim[i] = b->prim[j]; b->prim[j] = t; }
    }
    uint32_t nl = i - first;
    if (nl == 0 || nl == count) nl = count / 2; /* degenerate: split by index */
    uint32_t l = build_rec(b, first, nl);
    uint32_t r = build_rec(b, first + nl, count - nl);
    nd = &b->nodes[me];
    nd->left = l; nd->right = r; nd->first = 0; nd->count = 0;
    return me;
}

static void bvh_build(bvh_t* out, const float* lo, const float* hi, uint32_t n)
{
    memset(out, 0, sizeof(*out));
    if (n == 0) return;
    build_t b;
    b.lo = lo; b.hi = hi;
    b.cen = (float*)malloc(sizeof(float) * 3 * n);
    b.prim = (uint32_t*)malloc(sizeof(uint32_t) * n);
    b.nodes = (node_t*)malloc(sizeof(node_t) * (2 * (size_t)n + 1));
    b.n_nodes = 0;
    for (uint32_t i = 0; i < n; i++) {
        b.prim[i] = i;
        for (int k = 0; k < 3; k++) b.cen[i * 3 + k] = 0.5f * (lo[i * 3 + k] + hi[i * 3 + k]);
    }
    build_rec(&b, 0, n);
    free(b.cen);
    out->nodes = b.nodes; out->n_nodes = b.n_nodes; out->prim = b.prim;
}

static void bvh_free(bvh_t* b) { free(b->nodes); free(b->prim); memset(b, 0, sizeof(*b)); }

static inline float inflate_eps(float lo, float hi)
{
    return 1e-5f * (1.0f + fmaxf(fabsf(lo), fabsf(hi)));
}

grto_scene* grto_scene_create(const grto_particle* particles, uint64_t n, float alpha_min)
{
    grto_scene* s = (grto_scene*)calloc(1, sizeof(*s));
    s->n = n;
    s->alpha_min = alpha_min;
    s->parts = (grto_particle*)malloc(sizeof(grto_particle) * (n ? n : 1));
    memcpy(s->parts, particles, sizeof(grto_particle) * n);
    s->prox = (proxy_t*)malloc(sizeof(proxy_t) * (n ? n : 1));
    float* lo = (float*)malloc(sizeof(float) * 3 * (n ? n : 1));
    float* hi = (float*)malloc(sizeof(float) * 3 * (n ? n : 1));
    float iv[12][3];
    uint32_t ii[60];
    grto_icosahedron(iv, ii);
    uint64_t m = 0;
    for (uint64_t i = 0; i < n; i++) {
        const grto_particle* p = &particles[i];
        const float sc = grto_proxy_scale(p->opacity, alpha_min); /* GaussianTracer.cpp:306 */
        if (!(sc > 0.0f)) continue;                                /* decision (vi) */
        proxy_t* q = &s->prox[m];
        memcpy(q->mu, p->pos, sizeof(q->mu));
        grto_inv_cov(p, q->A);
        q->s = sc;
        q->opacity = p->opacity;
        q->id = (uint32_t)i;
        /* world AABB of the 12 vertices of M = T * (R * diag(scale*s)) (GaussianTracer.cpp:307-311) */
        float Rg[9];
        grto_mat3_cast(p->quat, Rg);
        float l[3] = {INFINITY, INFINITY, INFINITY}, h[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int v = 0; v < 12; v++) {
            const float lx = (p->scale[0] * sc) * iv[v][0], ly = (p->scale[1] * sc) * iv[v][1],
                        lz = (p->scale[2] * sc) * iv[v][2];
            for (int r = 0; r < 3; r++) {
                /* R_math[r][c] = Rg[c*3+r] */
                const float w = (Rg[0 * 3 + r] * lx + Rg[1 * 3 + r] * ly) + Rg[2 * 3 + r] * lz + p->pos[r];
                l[r] = fminf(l[r], w);
                h[r] = fmaxf(h[r], w);
            }
        }
        for (int r = 0; r < 3; r++) {
            const float e = inflate_eps(l[r], h[r]);
            lo[m * 3 + r] = l[r] - e;
            hi[m * 3 + r] = h[r] + e;
        }
        m++;
    }
    s->m = m;
    bvh_build(&s->gbvh, lo, hi, (uint32_t)m);
    free(lo); free(hi);
    s->use_bvh = 1;
    return s;
}

void grto_scene_use_bvh(grto_scene* s, int use_bvh) { s->use_bvh = use_bvh; }

void grto_scene_set_mesh(grto_scene* s, const float* verts, const float* normals, uint32_t nv, const uint32_t* faces,
                         uint32_t nf)
{
    free(s->mv); free(s->mn); free(s->mf);
    bvh_free(&s->mbvh);
    s->mv = s->mn = NULL; s->mf = NULL; s->nv = s->nf = 0;
    if (nf == 0) return;
    s->mv = (float*)malloc(sizeof(float) * 3 * nv);
    s->mn = (float*)malloc(sizeof(float) * 3 * nv);
    s->mf = (uint32_t*)malloc(sizeof(uint32_t) * 3 * nf);
    memcpy(s->mv, verts, sizeof(float) * 3 * nv);
    memcpy(s->mn, normals, sizeof(float) * 3 * nv);
    memcpy(s->mf, faces, sizeof(uint32_t) * 3 * nf);
    s->nv = nv; s->nf = nf;
    float* lo = (float*)malloc(sizeof(float) * 3 * nf);
    float* hi = (float*)malloc(sizeof(float) * 3 * nf);
    for (uint32_t f = 0; f < nf; f++) {
        for (int k = 0; k < 3; k++) {
            float a = verts[faces[f * 3 + 0] * 3 + k], b = verts[faces[f * 3 + 1] * 3 + k],
                  c = verts[faces[f * 3 + 2] * 3 + k];
            float l = fminf(a, fminf(b, c)), h = fmaxf(a, fmaxf(b, c));
            float e = inflate_eps(l, h);
            lo[f * 3 + k] = l - e;
            hi[f * 3 + k] = h + e;
        }
    }
    bvh_build(&s->mbvh, lo, hi, nf);
    free(lo); free(hi);
}

/* Decision (v) of SURVEY 8(c) replaces the ray / triangle tests OptiX runs against the instanced 20-triangle icosahedron
 * (src/geometry/Icosahedron.h:13-37; instance transform T * (R * S), src/GaussianTracer.cpp:304-311; one GAS, one instance per
 * particle, :401-420) by ten slab tests in Gaussian space.  This mode puts the triangles back, for CHECKING that decision on whole
 * frames: the caller hands over the instanced vertices — the tests take them from oracle/_ref (ref_instance_vertex x
 * ref_icosahedron: the reference's own glm chain and mesh, compiled from its sources) — and every proxy test intersects the 20
 * triangles (Moeller-Trumbore in double precision, no culling: tracer.cuh:306) instead of the slabs; everything behind the
 * intersection (interval, k-buffer, response, integration) is unchanged.  verts == NULL switches back. */
void grto_scene_set_proxy_triangles(grto_scene* s, const float* verts, const uint32_t idx[60])
{
    free(s->ptv);
    s->ptv = NULL;
    s->tri_mode = 0;
    if (!verts) return;
    s->ptv = (float*)malloc(sizeof(float) * 36 * (s->n ? s->n : 1));
    memcpy(s->ptv, verts, sizeof(float) * 36 * s->n);
    memcpy(s->pti, idx, sizeof(uint32_t) * 60);
    s->tri_mode = 1;
}

void grto_scene_destroy(grto_scene* s)
{
    if (!s) return;
    free(s->ptv);
    free(s->parts); free(s->prox);
    bvh_free(&s->gbvh);
    free(s->mv); free(s->mn); free(s->mf);
    bvh_free(&s->mbvh);
    free(s);
}

/* conservative ray/box interval */
typedef struct { f3 o, inv; } rayinv_t;
static inline rayinv_t mk_rayinv(f3 o, f3 d)
{
    rayinv_t r;
    r.o = o;
    r.inv.x = 1.0f / (fabsf(d.x) < 1e-30f ? copysignf(1e-30f, d.x) : d.x);
    r.inv.y = 1.0f / (fabsf(d.y) < 1e-30f ? copysignf(1e-30f, d.y) : d.y);
    r.inv.z = 1.0f / (fabsf(d.z) < 1e-30f ? copysignf(1e-30f, d.z) : d.z);
    return r;
}
static inline int box_interval(const node_t* n, const rayinv_t* r, float* tn, float* tf)
{
    float t0 = (n->lo[0] - r->o.x) * r->inv.x, t1 = (n->hi[0] - r->o.x) * r->inv.x;
    float a = fminf(t0, t1), b = fmaxf(t0, t1);
    t0 = (n->lo[1] - r->o.y) * r->inv.y; t1 = (n->hi[1] - r->o.y) * r->inv.y;
    a = fmaxf(a, fminf(t0, t1)); b = fminf(b, fmaxf(t0, t1));
    t0 = (n->lo[2] - r->o.z) * r->inv.z; t1 = (n->hi[2] - r->o.z) * r->inv.z;
    a = fmaxf(a, fminf(t0, t1)); b = fminf(b, fmaxf(t0, t1));
    *tn = a; *tf = b;
    return a <= b;
}

/* ------------------------------------------------------------------------------------------ */
/* a11: traceGPs + __anyhit__anyhit — tracer.cuh:289-326, tracer.cu:124-153                    */
/* One traversal returning the k = 7 nearest proxy hits inside the open interval.  Hits are     */
/* totally ordered by the key (t, particle id, entry<exit) — decision (iv).                      */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint64_t key; float alpha; } khit_t;

static inline uint64_t mk_key(float t, uint32_t id, uint32_t is_exit)
{
    uint32_t tb;
    memcpy(&tb, &t, 4);
    return ((uint64_t)tb << 32) | ((uint64_t)id << 1) | (uint64_t)is_exit;
}
static inline float key_t(uint64_t k) { uint32_t tb = (uint32_t)(k >> 32); float t; memcpy(&t, &tb, 4); return t; }
static inline uint32_t key_id(uint64_t k) { return (uint32_t)(k & 0xFFFFFFFFu) >> 1; }

#define KEY_INVALID 0xFFFFFFFFFFFFFFFFull

/* insertion identical in effect to the reference's 7 compare-and-swap steps (tracer.cu:124-146) */
static inline void kbuf_insert(khit_t buf[MAX_HITS_PER_TRACE], uint64_t key, float alpha)
{
    if (key >= buf[MAX_HITS_PER_TRACE - 1].key) return;
    khit_t h = {key, alpha};
    for (int i = 0; i < MAX_HITS_PER_TRACE; i++) {
        if (h.key < buf[i].key) { khit_t t = buf[i]; buf[i] = h; h = t; }
    }
}

typedef struct {
    const grto_scene* s;
    f3 o, d;
    uint64_t last_key; /* exclusive lower bound on the key */
    float t_hi;        /* exclusive upper bound on t */
    khit_t* buf;
    grto_counters* c;
} gps_ctx;

/* ray (full line) against one triangle in double precision; no face culling (tracer.cuh:306) */
static inline int tri_hit_f64(const float* a, const float* b, const float* c, f3 o, f3 d, double* t)
{
    const double e1[3] = {(double)b[0] - a[0], (double)b[1] - a[1], (double)b[2] - a[2]};
    const double e2[3] = {(double)c[0] - a[0], (double)c[1] - a[1], (double)c[2] - a[2]};
    const double dd[3] = {d.x, d.y, d.z}, tv[3] = {(double)o.x - a[0], (double)o.y - a[1], (double)o.z - a[2]};
    const double p[3] = {dd[1] * e2[2] - dd[2] * e2[1], dd[2] * e2[0] - dd[0] * e2[2], dd[0] * e2[1] - dd[1] * e2[0]};
    const double det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
    if (det == 0.0) return 0;
    const double inv = 1.0 / det;
    const double u = (tv[0] * p[0] + tv[1] * p[1] + tv[2] * p[2]) * inv;
    if (!(u >= 0.0 && u <= 1.0)) return 0;
    const double q[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
    const double v = (dd[0] * q[0] + dd[1] * q[1] + dd[2] * q[2]) * inv;
    if (!(v >= 0.0 && u + v <= 1.0)) return 0;
    *t = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) * inv;
    return 1;
}

/* the instanced icosahedron of particle `id`: nearest and farthest crossing of the ray's line (a convex body: decision (iv),
 * at most one entry and one exit, however many triangles share the touched edge) */
static inline int proxy_triangles(const grto_scene* s, uint32_t id, f3 o, f3 d, float* t_entry, float* t_exit)
{
    const float* v = &s->ptv[(size_t)id * 36];
    double tmin = INFINITY, tmax = -INFINITY;
    int n = 0;
    for (int f = 0; f < 20; f++) {
        double t;
        if (!tri_hit_f64(&v[s->pti[f * 3] * 3], &v[s->pti[f * 3 + 1] * 3], &v[s->pti[f * 3 + 2] * 3], o, d, &t)) continue;
        tmin = fmin(tmin, t); tmax = fmax(tmax, t);
        n++;
    }
    if (n == 0) return 0;
    *t_entry = (float)tmin;
    *t_exit = (float)tmax;
    return 1;
}

static inline void gps_test_proxy(gps_ctx* g, const proxy_t* q)
{
    if (g->c) g->c->proxy_tests++;
    const f3 mu = ld3(q->mu);
    const f3 o_g = matvec(q->A, sub3(g->o, mu));
    const f3 d_g = matvec(q->A, g->d);
    float te, tx;
    if (g->s->tri_mode) { if (!proxy_triangles(g->s, q->id, g->o, g->d, &te, &tx)) return; }
    else if (!proxy_slabs(o_g, d_g, q->s, &te, &tx)) return;
    const float t_lo = key_t(g->last_key);
    int in_e = (te >= t_lo) && (te < g->t_hi);
    int in_x = (tx >= t_lo) && (tx < g->t_hi);
    if (!in_e && !in_x) return;
    /* alpha is independent of the hit distance (tracer.cuh:354-357) so it is evaluated once
     * per particle and carried by both the entry and the exit hit */
    float alpha = response_from(q->A, mu, g->o, g->d, o_g, d_g);
    alpha = fminf(0.99f, alpha * q->opacity);
    if (in_e) { uint64_t k = mk_key(te, q->id, 0); if (k > g->last_key) kbuf_insert(g->buf, k, alpha); }
    if (in_x) { uint64_t k = mk_key(tx, q->id, 1); if (k > g->last_key) kbuf_insert(g->buf, k, alpha); }
}

static void gps_traverse(gps_ctx* g)
{
    const grto_scene* s = g->s;
    for (int i = 0; i < MAX_HITS_PER_TRACE; i++) { g->buf[i].key = KEY_INVALID; g->buf[i].alpha = 0.0f; }
    if (s->m == 0) return;
    if (!s->use_bvh) {
        for (uint64_t i = 0; i < s->m; i++) gps_test_proxy(g, &s->prox[i]);
        return;
    }
    const rayinv_t ri = mk_rayinv(g->o, g->d);
    const float t_lo = key_t(g->last_key);
    uint32_t stack[128];
    int sp = 0;
    stack[sp++] = 0;
    while (sp) {
        const node_t* n = &s->gbvh.nodes[stack[--sp]];
        if (g->c) g->c->node_visits++;
        float tn, tf;
        if (!box_interval(n, &ri, &tn, &tf)) continue;
        float t_k = key_t(g->buf[MAX_HITS_PER_TRACE - 1].key); /* NaN-bits when invalid => compare false */
        if (tf < t_lo || tn >= g->t_hi) continue;
        if (g->buf[MAX_HITS_PER_TRACE - 1].key != KEY_INVALID && tn > t_k) continue;
        if (n->count) {
            for (uint32_t i = 0; i < n->count; i++) gps_test_proxy(g, &s->prox[s->gbvh.prim[n->first + i]]);
        } else {
            /* near child last on the stack (popped first) */
            const node_t* l = &s->gbvh.nodes[n->left];
            const node_t* r = &s->gbvh.nodes[n->right];
            float ln, lf, rn, rf;
            int hl = box_interval(l, &ri, &ln, &lf), hr = box_interval(r, &ri, &rn, &rf);
            if (hl && hr) {
                if (ln <= rn) { stack[sp++] = n->right; stack[sp++] = n->left; }
                else { stack[sp++] = n->left; stack[sp++] = n->right; }
            } else if (hl) stack[sp++] = n->left;
            else if (hr) stack[sp++] = n->right;
        }
    }
}

uint32_t grto_trace_gps(const grto_scene* s, const float o[3], const float d[3], float tmin, float tmax,
                        uint32_t ids[7], float ts[7])
{
    khit_t buf[MAX_HITS_PER_TRACE];
    gps_ctx g = {s, ld3(o), ld3(d), mk_key(tmin, 0x7FFFFFFFu, 1), tmax, buf, NULL};
    gps_traverse(&g);
    uint32_t n = 0;
    for (int i = 0; i < MAX_HITS_PER_TRACE; i++) {
        if (buf[i].key == KEY_INVALID) { ids[i] = 0xFFFFFFFFu; ts[i] = 1e20f; } /* tracer.cuh:63-64,296-297 */
        else { ids[i] = key_id(buf[i].key); ts[i] = key_t(buf[i].key); n++; }
    }
    return n;
}

/* ------------------------------------------------------------------------------------------ */
/* a12: trace() — shaders/tracer.cuh:328-373                                                   */
/* ------------------------------------------------------------------------------------------ */
void grto_trace(const grto_scene* s, const grto_params* prm, const float o_[3], const float d_[3], float t_min,
                float t_max, float* density_io, float radiance_out[3], grto_counters* c)
{
    const f3 o = ld3(o_), d = ld3(d_);
    float T = 1.0f - *density_io;                  /* :334 */
    const float epsT = 1e-9f;                      /* :335 */
    float lastT = t_min;                           /* :336 */
    f3 radiance = mk3(0.0f, 0.0f, 0.0f);           /* :338 */
    const f3 dn = normalize3(d);                   /* normalize(ray_d) passed to computeRadiance, :359,362 */
    khit_t buf[MAX_HITS_PER_TRACE];
    /* first round: t > lastT + epsT ; later rounds: key > key of the 7th hit (== "t > lastT" up to
     * exact-t ties, decision (iv); lastT + 1e-9f == lastT for every t >= 2^-5) */
    uint64_t last_key = mk_key(lastT + epsT, 0x7FFFFFFFu, 1);
    const float t_hi = t_max + epsT;               /* :342 */
    if (c) c->segments++;
    while (lastT <= t_max && T > prm->min_transmittance) { /* :341 */
        gps_ctx g = {s, o, d, last_key, t_hi, buf, c};
        gps_traverse(&g);
        if (c) c->rounds++;
        if (buf[0].key == KEY_INVALID) break;      /* :344-346 */
        int n = 0;
        for (int i = 0; i < MAX_HITS_PER_TRACE; i++) { /* :349-368 */
            if (buf[i].key == KEY_INVALID) break;
            n++;
            if (T > prm->min_transmittance) {
                if (c) c->hit_evals++;
                lastT = fmaxf(key_t(buf[i].key), lastT);
                const float hitAlpha = buf[i].alpha; /* fminf(0.99, response*opacity), :356-357 */
                if (prm->alpha_min < hitAlpha) {     /* :361 */
                    const grto_particle* p = &s->parts[key_id(buf[i].key)];
                    f3 L = sh_to_radiance(p->sh, dn, prm->sh_degree_max);
                    L = mk3(fmaxf(L.x, 0.0f), fmaxf(L.y, 0.0f), fmaxf(L.z, 0.0f));
                    /* radiance += rayTransmittance * hitRadiance * hitAlpha : ((T*L)*alpha), :364 */
                    radiance = add3(radiance, mul3s(mul3s(L, T), hitAlpha));
                    T *= (1.0f - hitAlpha);          /* :365 */
                }
            }
        }
        if (n < MAX_HITS_PER_TRACE) break; /* the reference's next traceGPs would return nothing */
        last_key = buf[MAX_HITS_PER_TRACE - 1].key;
    }
    st3(radiance_out, radiance);                   /* :371 */
    *density_io = 1.0f - T;                        /* :372 */
}

/* ------------------------------------------------------------------------------------------ */
/* a9: traceMesh + closest hit + barycentric normal — tracer.cuh:266-287,167-185               */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int hit; float t, u, v; uint32_t face; } mesh_hit_t;

static inline void mesh_test_face(const grto_scene* s, uint32_t f, const float o[3], const float d[3], float tmin,
                                  mesh_hit_t* best, float* tmax)
{
    float t, u, v;
    const uint32_t* fc = &s->mf[f * 3];
    if (!grto_tri_hit(&s->mv[fc[0] * 3], &s->mv[fc[1] * 3], &s->mv[fc[2] * 3], o, d, &t, &u, &v)) return;
    if (!(t > tmin && t < *tmax)) {
        /* exact tie with the current best: lowest face index wins (decision (iv) analogue) */
        if (!(best->hit && t == best->t && f < best->face)) return;
    }
    best->hit = 1; best->t = t; best->u = u; best->v = v; best->face = f;
    *tmax = t;
}

static mesh_hit_t mesh_closest(const grto_scene* s, f3 o_, f3 d_, float tmin, float tmax)
{
    mesh_hit_t best = {0, 0, 0, 0, 0};
    if (s->nf == 0) return best; /* mesh_handle == 0 => miss (decision (ix)) */
    float o[3], d[3];
    st3(o, o_); st3(d, d_);
    if (!s->use_bvh) {
        for (uint32_t f = 0; f < s->nf; f++) mesh_test_face(s, f, o, d, tmin, &best, &tmax);
        return best;
    }
    const rayinv_t ri = mk_rayinv(o_, d_);
    uint32_t stack[128];
    int sp = 0;
    stack[sp++] = 0;
    while (sp) {
        const node_t* n = &s->mbvh.nodes[stack[--sp]];
        float tn, tf;
        if (!box_interval(n, &ri, &tn, &tf)) continue;
        if (tf < tmin || tn > tmax) continue;
        if (n->count) {
            for (uint32_t i = 0; i < n->count; i++) mesh_test_face(s, s->mbvh.prim[n->first + i], o, d, tmin, &best, &tmax);
        } else {
            stack[sp++] = n->right;
            stack[sp++] = n->left;
        }
    }
    return best;
}

static inline f3 bary_normal(const grto_scene* s, const mesh_hit_t* h)
{
    const uint32_t* fc = &s->mf[h->face * 3];
    const f3 n0 = ld3(&s->mn[fc[0] * 3]), n1 = ld3(&s->mn[fc[1] * 3]), n2 = ld3(&s->mn[fc[2] * 3]);
    const float w0 = 1.0f - h->u - h->v, w1 = h->u, w2 = h->v; /* tracer.cuh:179-181 */
    return normalize3(add3(add3(mul3s(n0, w0), mul3s(n1, w1)), mul3s(n2, w2)));
}

/* ------------------------------------------------------------------------------------------ */
/* a8: __raygen__raygeneration + __closesthit__ + __miss__ — tracer.cu:17-187    */
/* ------------------------------------------------------------------------------------------ */
enum { TraceLastGaussianPass = 0, TraceGaussianPass = 1, TraceMeshPass = 2, TraceTerminate = 3 };
enum { MIRROR = 0, NORMAL = 1, GLASS = 2 };

static void shade_ray(const grto_scene* s, const grto_params* prm, f3 curO, f3 curD, float rgb[3], grto_counters* c)
{
    f3 accumColor = mk3(0, 0, 0), directLight = mk3(0, 0, 0);
    float accumAlpha = 0.0f, blocking = 0.0f, t_hit_payload = 0.0f, density = 0.0f;
    unsigned int numBounces = 0, timeout = 0;
    while (length3(curD) > 0.1f && numBounces < prm->max_bounces) {   /* tracer.cu:59 */
        const f3 ray_o = curO, ray_d = curD;
        float ro[3], rd[3];
        st3(ro, ray_o); st3(rd, ray_d);
        int state = TraceMeshPass;                                    /* traceMesh, tracer.cuh:268 */
        mesh_hit_t mh = mesh_closest(s, ray_o, ray_d, TRACE_MESH_TMIN, TRACE_MESH_TMAX);
        if (mh.hit) {                                                 /* __closesthit__, tracer.cu:155-187 */
            float t_hit = mh.t;
            const f3 normal = bary_normal(s, &mh);
            f3 newDir = mk3(0, 0, 0);
            state = TraceGaussianPass;
            if (prm->type == MIRROR) {                                /* renderMirror, tracer.cuh:396-404 */
                newDir = reflect3(ray_d, normal);
                numBounces += 1;
            } else if (prm->type == NORMAL) {                         /* renderNormal, tracer.cuh:406-429 */
                float rad[3];
                grto_trace(s, prm, ro, rd, prm->t_min, t_hit, &density, rad, c);
                const float alpha = density;
                accumColor = add3(accumColor, ld3(rad));
                accumAlpha += alpha;
                const f3 normalColor = mul3s(add3(normal, mk3(1.0f, 1.0f, 1.0f)), 0.5f); /* (n+1)/2 */
                accumColor = add3(accumColor, mul3s(normalColor, 1.0f - alpha));
                accumAlpha += (1.0f - alpha);
                state = TraceTerminate;
            } else if (prm->type == GLASS) {                          /* renderGlass, tracer.cuh:466-482 */
                float nd[3], nrm[3], out[3];
                st3(nd, ray_d); st3(nrm, normal);
                const float n1 = 1.0003f, n2 = 1.5f;
                if (grto_refract(nd, nrm, n2 / n1, out)) t_hit += REFRACTION_EPS_SHIFT;
                else numBounces += 1;
                newDir = ld3(out);
            }
            t_hit_payload = t_hit;
            curO = add3(ray_o, mul3s(ray_d, t_hit));
            curD = newDir;
        } else {                                                      /* __miss__, tracer.cu:112-122 */
            curO = mk3(0, 0, 0);
            curD = mk3(0, 0, 0);
            state = TraceLastGaussianPass;
        }
        if (state == TraceTerminate) break;                           /* tracer.cu:65 */
        float rad[3];
        if (state == TraceLastGaussianPass) {                         /* tracer.cu:68-82 */
            grto_trace(s, prm, ro, rd, prm->t_min, prm->t_max, &density, rad, c);
            const float alpha = density;
            directLight = mul3s(ld3(rad), alpha);
            accumAlpha = clampf(accumAlpha + alpha, 0.0f, 1.0f);
        } else {                                                      /* tracer.cu:84-98 */
            grto_trace(s, prm, ro, rd, prm->t_min, t_hit_payload, &density, rad, c);
            const float alpha = density;
            accumColor = add3(accumColor, mul3s(ld3(rad), 1.0f - accumAlpha));
            accumAlpha = clampf(accumAlpha + alpha, 0.0f, 1.0f);
            blocking = clampf(blocking + alpha, 0.0f, 1.0f);
        }
        accumColor = add3(accumColor, mul3s(directLight, 1.0f - blocking)); /* tracer.cu:101 */
        timeout += 1;
        if (timeout > TIMEOUT_ITERATIONS) break;
    }
    st3(rgb, accumColor);
}

void grto_render_pixel(const grto_scene* s, const grto_params* prm, uint32_t ix, uint32_t iy, float rgb[3],
                       grto_counters* c)
{
    float o[3], d[3];
    float nU[3] = {-prm->U[0], -prm->U[1], -prm->U[2]}, nV[3] = {-prm->V[0], -prm->V[1], -prm->V[2]};
    if (!prm->mode_fisheye) {
        grto_get_ray(ix, iy, nU, nV, prm->W, prm->eye, prm->width, prm->height, o, d);
    } else if (!grto_get_fisheye_ray(ix, iy, nU, nV, prm->W, prm->eye, prm->width, prm->height, o, d)) {
        rgb[0] = rgb[1] = rgb[2] = 0.0f; /* decision (vii) + pre-clear GaussianTracer.cpp:510-513 */
        return;
    }
    if (c) c->rays++;
    shade_ray(s, prm, ld3(o), ld3(d), rgb, c);
}

static inline void add_counters(grto_counters* a, const grto_counters* b)
{
    a->rays += b->rays; a->segments += b->segments; a->hit_evals += b->hit_evals; a->rounds += b->rounds;
    a->node_visits += b->node_visits; a->proxy_tests += b->proxy_tests;
}

void grto_render(const grto_scene* s, const grto_params* prm, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                 uint8_t* out_u8, float* out_f32, grto_counters* c, int n_threads)
{
    const uint32_t TW = 8, TH = 8;
    const uint32_t tx = (x1 - x0 + TW - 1) / TW, ty = (y1 - y0 + TH - 1) / TH;
    const int64_t ntiles = (int64_t)tx * ty;
    grto_counters total;
    memset(&total, 0, sizeof(total));
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#pragma omp parallel num_threads(n_threads)
#endif
    {
        grto_counters loc;
        memset(&loc, 0, sizeof(loc));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
        for (int64_t t = 0; t < ntiles; t++) {
            const uint32_t bx = x0 + (uint32_t)(t % tx) * TW, by = y0 + (uint32_t)(t / tx) * TH;
            for (uint32_t y = by; y < by + TH && y < y1; y++)
                for (uint32_t x = bx; x < bx + TW && x < x1; x++) {
                    float rgb[3];
                    grto_render_pixel(s, prm, x, y, rgb, &loc);
                    const size_t pi = (size_t)y * prm->width + x; /* tracer.cuh:487 */
                    if (out_f32) { out_f32[pi * 3] = rgb[0]; out_f32[pi * 3 + 1] = rgb[1]; out_f32[pi * 3 + 2] = rgb[2]; }
                    if (out_u8) {
                        out_u8[pi * 3] = grto_quantize(rgb[0]);
                        out_u8[pi * 3 + 1] = grto_quantize(rgb[1]);
                        out_u8[pi * 3 + 2] = grto_quantize(rgb[2]);
                    }
                }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        add_counters(&total, &loc);
    }
    if (c) *c = total;
}

void grto_render_rays(const grto_scene* s, const grto_params* prm, const float* rays, uint64_t n, float* out_f32,
                      grto_counters* c, int n_threads)
{
    grto_counters total;
    memset(&total, 0, sizeof(total));
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#pragma omp parallel num_threads(n_threads)
#endif
    {
        grto_counters loc;
        memset(&loc, 0, sizeof(loc));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 64)
#endif
        for (int64_t i = 0; i < (int64_t)n; i++) {
            loc.rays++;
            shade_ray(s, prm, ld3(&rays[i * 6]), ld3(&rays[i * 6 + 3]), &out_f32[i * 3], &loc);
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        add_counters(&total, &loc);
    }
    if (c) *c = total;
}
