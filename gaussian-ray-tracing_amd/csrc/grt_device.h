// grt_device.h — device-side math of the Gaussian ray tracer (gfx950 only).
//
// Compiled with -ffp-contract=off: every multiply/add rounds where the reference's source
// expression rounds (shaders/tracer.cuh, src/vector_math.h, glm), so the per-ray hit ORDER is
// reproducible; explicit __builtin_fmaf is used only in the conservative box tests, whose result
// never decides a hit.  File:line citations are into Ray-Studio2/gaussian-ray-tracing.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace grt {

struct f3 { float x, y, z; };

__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 mul3s(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 neg3(f3 a) { return mk3(-a.x, -a.y, -a.z); }
// src/vector_math.h:572-575
__device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// src/vector_math.h:578-581
__device__ __forceinline__ f3 cross3(f3 a, f3 b)
{
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// src/vector_math.h:584-594
__device__ __forceinline__ float length3(f3 v) { return sqrtf(dot3(v, v)); }
__device__ __forceinline__ f3 normalize3(f3 v)
{
    float invLen = 1.0f / sqrtf(dot3(v, v));
    return mul3s(v, invLen);
}
// src/vector_math.h:146
__device__ __forceinline__ float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }
// src/vector_math.h:603-606 : i - 2.0f * n * dot(n, i)
__device__ __forceinline__ f3 reflect3(f3 i, f3 n)
{
    f3 n2 = mul3s(n, 2.0f);
    return sub3(i, mul3s(n2, dot3(n, i)));
}

// ---- constants: src/Parameters.h:10-23, shaders/tracer.cuh:9-14 ----
#define GRT_SH_C0 0.28209479177387814f
#define GRT_SH_C1 0.4886025119029199f
#define GRT_SH_C2_0 1.0925484305920792f
#define GRT_SH_C2_1 -1.0925484305920792f
#define GRT_SH_C2_2 0.31539156525252005f
#define GRT_SH_C2_3 -1.0925484305920792f
#define GRT_SH_C2_4 0.5462742152960396f
#define GRT_SH_C3_0 -0.5900435899266435f
#define GRT_SH_C3_1 2.890611442640554f
#define GRT_SH_C3_2 -0.4570457994644658f
#define GRT_SH_C3_3 0.3731763325901154f
#define GRT_SH_C3_4 -0.4570457994644658f
#define GRT_SH_C3_5 1.445305721320277f
#define GRT_SH_C3_6 -0.5900435899266435f

constexpr float kTraceMeshTmin = 1e-5f;
constexpr float kTraceMeshTmax = 1e5f;
constexpr float kRefractionEpsShift = 1e-5f;
constexpr uint32_t kTimeoutIterations = 1000u;

// icosahedron face-normal pairs (src/geometry/Icosahedron.h:13-37): (0,P,+-Q) + cyclic shifts,
// (1,+-1,+-1)/sqrt(3); P = phi/sqrt(3), Q = 1/(phi sqrt(3))
constexpr float kIcoP = 0.9341723322868347f;
constexpr float kIcoQ = 0.35682210326194763f;
constexpr float kSqrt3 = 1.7320508075688772f;

// glm mat3 * vec3 with row-major math matrix (third_party/glm/detail/type_mat3x3.inl:468-474)
struct m33 { float a[9]; };
__device__ __forceinline__ f3 matvec(const m33& A, f3 v)
{
    return mk3(A.a[0] * v.x + A.a[1] * v.y + A.a[2] * v.z,
               A.a[3] * v.x + A.a[4] * v.y + A.a[5] * v.z,
               A.a[6] * v.x + A.a[7] * v.y + A.a[8] * v.z);
}

// glm::mat3_cast (third_party/glm/gtc/quaternion.inl:47-72) then invCov = inv_s * transpose(R)
// (shaders/tracer.cuh:193-201): A[r][c] = (1/scale_r) * Rg[r*3+c]  (Rg column-major)
__device__ __forceinline__ void mat3_cast(float w, float x, float y, float z, float Rg[9])
{
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

// projections on the 10 slab normals; 4..7 un-normalised (compared against s*sqrt(3))
__device__ __forceinline__ void slab_project(f3 v, float a[10])
{
    const float py = kIcoP * v.y, qz = kIcoQ * v.z;
    const float qx = kIcoQ * v.x, pz = kIcoP * v.z;
    const float px = kIcoP * v.x, qy = kIcoQ * v.y;
    const float xpy = v.x + v.y, xmy = v.x - v.y;
    a[0] = py - qz; a[1] = py + qz;
    a[2] = qx - pz; a[3] = qx + pz;
    a[4] = xpy - v.z; a[5] = xmy - v.z; a[6] = xmy + v.z; a[7] = xpy + v.z;
    a[8] = px - qy; a[9] = px + qy;
}

// exact proxy test; same operation sequence as oracle/grt_oracle.c:proxy_slabs
// a[] = slab_project(o_g), formed by the caller (per lane, or once per eye and particle: k_eye_records)
__device__ __forceinline__ bool proxy_slabs_pre(const float a[10], f3 d_g, float s, float& t_entry, float& t_exit)
{
    float b[10];
    slab_project(d_g, b);
    const float s3 = s * kSqrt3;
    float nn = 0.0f, nd = 0.0f, fn = 0.0f, fd = 0.0f;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const float h = (i >= 4 && i <= 7) ? s3 : s;
        // a with the sign of b folded in: the sign BIT of b (so b = -0 counts as negative; its slab then sits at -+1e30 times
        // something on either reading and bounds nothing), two VALU operations instead of a compare, a move and a select
        const float ap = __uint_as_float(__float_as_uint(a[i]) ^ (__float_as_uint(b[i]) & 0x80000000u));
        const float bp = fmaxf(fabsf(b[i]), 1e-30f);
        const float cn = -(ap + h);
        const float cf = h - ap;
        if (i == 0) {
            nn = cn; nd = bp; fn = cf; fd = bp;
        } else {
            if (cn * nd > nn * bp) { nn = cn; nd = bp; }
            if (cf * fd < fn * bp) { fn = cf; fd = bp; }
        }
    }
    t_entry = nn / nd;
    t_exit = fn / fd;
    return t_entry <= t_exit;
}
__device__ __forceinline__ bool proxy_slabs(f3 o_g, f3 d_g, float s, float& t_entry, float& t_exit)
{
    float a[10];
    slab_project(o_g, a);
    return proxy_slabs_pre(a, d_g, s, t_entry, t_exit);
}

// Conservative pre-test (culling only, never decides a hit): can the ray touch the sphere that circumscribes
// the proxy icosahedron in Gaussian space (circumradius / inradius = 1.2584086)?  The 4e-6 slack covers the
// fp32 rounding of the three dot products; when |o_g| is so large that the slack exceeds R^2 the test simply
// passes everything.
__device__ __forceinline__ float proxy_sphere_cc(f3 o_g, float s)
{
    const float R = 1.2585f * s;
    return dot3(o_g, o_g) - R * R;
}
__device__ __forceinline__ bool proxy_sphere_maybe_pre(f3 o_g, float cc, f3 d_g)
{
    const float b = dot3(o_g, d_g), aa = dot3(d_g, d_g);
    return (cc <= 0.0f) || (b * b * (1.0f + 4e-6f) >= aa * cc);
}
__device__ __forceinline__ bool proxy_sphere_maybe(f3 o_g, f3 d_g, float s)
{
    return proxy_sphere_maybe_pre(o_g, proxy_sphere_cc(o_g, s), d_g);
}

// ---- pieces of a split proxy (grt_api.hip: k_piece_boxes) ----
// The proxy-local box [-tt s, tt s]^3 of Gaussian space (tt = 1.0705: the icosahedron's extent along its principal axes)
// is cut into p0 x p1 x p2 cells; a piece's descriptor holds its cell (k_r) and the grid (p_r - 1), 5 bits each, bit 31 set.
constexpr float kIcoTTdev = 1.0704663f;
__host__ __device__ __forceinline__ uint32_t piece_desc(const uint32_t k[3], const uint32_t p[3])
{
    return 0x80000000u | k[0] | ((p[0] - 1u) << 5) | (k[1] << 10) | ((p[1] - 1u) << 15) | (k[2] << 20) | ((p[2] - 1u) << 25);
}
// Does the event at distance t of this ray (Gaussian-space origin o_g, direction d_g) belong to the piece?  Its point's
// cell index along every axis must be the piece's; the outermost cells reach to infinity, so every point has exactly one
// owner whatever the rounding at the proxy's surface.  (Scheduling only: which piece reports a hit never changes the hit.)
__device__ __forceinline__ bool piece_owns(uint32_t desc, float s, f3 o_g, f3 d_g, float t)
{
    const float inv = 0.5f / (kIcoTTdev * s);
    const float y[3] = {o_g.x + t * d_g.x, o_g.y + t * d_g.y, o_g.z + t * d_g.z};
    bool own = true;
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const uint32_t k = (desc >> (10 * r)) & 31u, pm1 = (desc >> (10 * r + 5)) & 31u;
        const float fp = (float)(pm1 + 1u);
        const float c = floorf((y[r] * inv + 0.5f) * fp); // cell index before clamping (NaN compares false below: no owner)
        const float ck = fminf(fmaxf(c, 0.0f), (float)pm1);
        own = own && (ck == (float)k);
    }
    return own;
}

// expf for an argument <= 0: the device library's own sequence (x log2(e) in two floats, v_exp_f32 of the fraction, v_ldexp_f32)
// without its two range checks — the overflow one cannot fire, and below -87 the scaling underflows to the same 0 the
// check would return.  Bit-identical to expf on (-87, 0]; four VALU instructions fewer per particle inserted.
__device__ __forceinline__ float exp_nonpos(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float ph = x * 0x1.715476p+0f;
    float pl = __builtin_fmaf(x, 0x1.715476p+0f, -ph);
    pl = __builtin_fmaf(x, 0x1.4ae0bep-26f, pl);
    const float e = __builtin_rintf(ph);
    const float a = (ph - e) + pl;
    return __builtin_ldexpf(__builtin_amdgcn_exp2f(a), (int)e);
#else
    return expf(x);
#endif
}

// computeResponse — shaders/tracer.cuh:187-214, given o_g = A(o-mu), d_g = A d already formed
__device__ __forceinline__ float response_from(const m33& A, f3 mu, f3 o, f3 d, f3 o_g, f3 d_g)
{
    const float d_val = -dot3(o_g, d_g) / fmaxf(1e-6f, dot3(d_g, d_g));
    const f3 pos = add3(o, mul3s(d, d_val));
    const f3 p_g = matvec(A, sub3(mu, pos));
    return exp_nonpos(-0.5f * dot3(p_g, p_g));
}

// SHToRadiance + computeRadiance — shaders/tracer.cuh:216-264.  sh points at 16 float3 (48 floats).
__device__ __forceinline__ f3 sh_radiance(const float* __restrict__ sh, f3 d, uint32_t deg)
{
#define GRT_SHV(i) mk3(sh[(i) * 3], sh[(i) * 3 + 1], sh[(i) * 3 + 2])
    f3 L = add3(mk3(0.5f, 0.5f, 0.5f), mul3s(GRT_SHV(0), GRT_SH_C0));
    if (deg >= 1) {
        const float x = d.x, y = d.y, z = d.z;
        {
            f3 t = add3(mul3s(GRT_SHV(1), -y), mul3s(GRT_SHV(2), z));
            t = sub3(t, mul3s(GRT_SHV(3), x));
            L = add3(L, mul3s(t, GRT_SH_C1));
        }
        if (deg >= 2) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z;
            {
                // "2. * zz - xx - yy" is a double expression in the reference (tracer.cuh:243)
                const float c22 = (float)((double)GRT_SH_C2_2 * (2. * (double)zz - (double)xx - (double)yy));
                f3 s = mul3s(GRT_SHV(4), GRT_SH_C2_0 * xy);
                s = add3(s, mul3s(GRT_SHV(5), GRT_SH_C2_1 * yz));
                s = add3(s, mul3s(GRT_SHV(6), c22));
                s = add3(s, mul3s(GRT_SHV(7), GRT_SH_C2_3 * xz));
                s = add3(s, mul3s(GRT_SHV(8), GRT_SH_C2_4 * (xx - yy)));
                L = add3(L, s);
            }
            if (deg >= 3) {
                f3 s = mul3s(GRT_SHV(9), (GRT_SH_C3_0 * y) * (3.0f * xx - yy));
                s = add3(s, mul3s(GRT_SHV(10), (GRT_SH_C3_1 * xy) * z));
                s = add3(s, mul3s(GRT_SHV(11), (GRT_SH_C3_2 * y) * (4.0f * zz - xx - yy)));
                s = add3(s, mul3s(GRT_SHV(12), (GRT_SH_C3_3 * z) * (2.0f * zz - 3.0f * xx - 3.0f * yy)));
                s = add3(s, mul3s(GRT_SHV(13), (GRT_SH_C3_4 * x) * (4.0f * zz - xx - yy)));
                s = add3(s, mul3s(GRT_SHV(14), (GRT_SH_C3_5 * z) * (xx - yy)));
                s = add3(s, mul3s(GRT_SHV(15), (GRT_SH_C3_6 * x) * (xx - 3.0f * yy)));
                L = add3(L, s);
            }
        }
    }
#undef GRT_SHV
    return mk3(fmaxf(L.x, 0.0f), fmaxf(L.y, 0.0f), fmaxf(L.z, 0.0f));
}

// getRay — shaders/tracer.cuh:115-134 (U, V arrive negated: shaders/tracer.cu:35-45)
__device__ __forceinline__ void get_ray(uint32_t ix, uint32_t iy, f3 U, f3 V, f3 W, uint32_t width, uint32_t height,
                                        f3& dir)
{
    const float dx = 2.0f * (((float)ix + 0.5f) / (float)(int)width) - 1.0f;
    const float dy = 2.0f * (((float)iy + 0.5f) / (float)(int)height) - 1.0f;
    dir = normalize3(add3(add3(mul3s(U, dx), mul3s(V, dy)), W));
}

// getFishEyeRay — shaders/tracer.cuh:136-165; r > 1 => no ray (decision vii)
__device__ __forceinline__ bool get_fisheye_ray(uint32_t ix, uint32_t iy, f3 U, f3 V, f3 W, uint32_t width,
                                                uint32_t height, f3& dir)
{
    const float dx = 2.0f * (((float)ix + 0.5f) / (float)(int)width) - 1.0f;
    const float dy = 2.0f * (((float)iy + 0.5f) / (float)(int)height) - 1.0f;
    const float r = sqrtf(dx * dx + dy * dy);
    if (r > 1.0f) return false;
    const float f = 1.0f / sqrtf(2.0f);
    const float theta = 2.0f * asinf(r / (2.0f * f));
    const float phi = atan2f(dy, dx);
    const f3 d = mk3(sinf(theta) * cosf(phi), sinf(theta) * sinf(phi), cosf(theta));
    dir = normalize3(add3(add3(mul3s(U, d.x), mul3s(V, d.y)), mul3s(W, d.z)));
    return true;
}

// quantizeUnsigned8Bits — shaders/tracer.cuh:68-73
__device__ __forceinline__ uint8_t quantize8(float x)
{
    x = clampf(x, 0.0f, 1.0f);
    unsigned int q = (unsigned int)(x * 256.0f);
    return (uint8_t)(q < 255u ? q : 255u);
}

// refract — shaders/tracer.cuh:432-464; returns true when refracted (caller shifts t_hit by 1e-5)
__device__ __forceinline__ bool refract_dir(f3 ray_d, f3 normal, float etai_over_etat, f3& out)
{
    float ri;
    if (dot3(ray_d, normal) < 0.0f) {
        ri = 1.0f / etai_over_etat;
    } else {
        ri = etai_over_etat;
        normal = neg3(normal);
    }
    float cos_theta = fminf(dot3(neg3(ray_d), normal), 1.0f);
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    if (ri * sin_theta > 1.0f) {
        f3 rn = dot3(ray_d, normal) < 0.0f ? normal : neg3(normal);
        out = reflect3(ray_d, rn);
        return false;
    }
    f3 r_out_perp = mul3s(add3(ray_d, mul3s(normal, cos_theta)), ri);
    f3 r_out_parallel = mul3s(normal, -sqrtf(fabsf(1.0f - dot3(r_out_perp, r_out_perp))));
    out = add3(r_out_perp, r_out_parallel);
    return true;
}

// Moeller-Trumbore, no culling; same operation sequence as oracle/grt_oracle.c:grto_tri_hit
__device__ __forceinline__ bool tri_hit(f3 v0, f3 v1, f3 v2, f3 o, f3 d, float& t, float& u, float& v)
{
    const f3 e1 = sub3(v1, v0), e2 = sub3(v2, v0);
    const f3 p = cross3(d, e2);
    const float det = dot3(e1, p);
    if (det == 0.0f) return false;
    const float inv = 1.0f / det;
    const f3 tv = sub3(o, v0);
    const float uu = dot3(tv, p) * inv;
    if (!(uu >= 0.0f && uu <= 1.0f)) return false;
    const f3 q = cross3(tv, e1);
    const float vv = dot3(d, q) * inv;
    if (!(vv >= 0.0f && uu + vv <= 1.0f)) return false;
    t = dot3(e2, q) * inv;
    u = uu;
    v = vv;
    return true;
}

// ---- conservative ray/box (culling only: FMA + reciprocal are fine here) ----
struct rayinv { f3 inv, oinv; };  // oinv = -o * inv
__device__ __forceinline__ rayinv mk_rayinv(f3 o, f3 d)
{
    rayinv r;
    r.inv.x = 1.0f / (fabsf(d.x) < 1e-30f ? copysignf(1e-30f, d.x) : d.x);
    r.inv.y = 1.0f / (fabsf(d.y) < 1e-30f ? copysignf(1e-30f, d.y) : d.y);
    r.inv.z = 1.0f / (fabsf(d.z) < 1e-30f ? copysignf(1e-30f, d.z) : d.z);
    r.oinv = mk3(-o.x * r.inv.x, -o.y * r.inv.y, -o.z * r.inv.z);
    return r;
}
__device__ __forceinline__ void box_interval(float lx, float ly, float lz, float hx, float hy, float hz,
                                             const rayinv& r, float& tn, float& tf)
{
    const float x0 = __builtin_fmaf(lx, r.inv.x, r.oinv.x), x1 = __builtin_fmaf(hx, r.inv.x, r.oinv.x);
    const float y0 = __builtin_fmaf(ly, r.inv.y, r.oinv.y), y1 = __builtin_fmaf(hy, r.inv.y, r.oinv.y);
    const float z0 = __builtin_fmaf(lz, r.inv.z, r.oinv.z), z1 = __builtin_fmaf(hz, r.inv.z, r.oinv.z);
    tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
    tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
}

// Euclidean distance from point o to an axis-aligned box (0 inside)
__device__ __forceinline__ float box_dist(float lx, float ly, float lz, float hx, float hy, float hz, f3 o)
{
    const float dx = fmaxf(fmaxf(lx - o.x, o.x - hx), 0.0f);
    const float dy = fmaxf(fmaxf(ly - o.y, o.y - hy), 0.0f);
    const float dz = fmaxf(fmaxf(lz - o.z, o.z - hz), 0.0f);
    return sqrtf(__builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, dz * dz)));
}

// Workgroup -> screen-block map.  Consecutive workgroup ids are dealt round-robin to the 8 XCDs (speed only,
// never correctness).  With chunk = C, XCD x works on runs of C consecutive screen blocks: blocks
// [g*8C + x*C, g*8C + (x+1)*C) for g = 0,1,...  Small C keeps per-XCD L2 locality (neighbouring tiles touch the
// same BVH nodes / proxies) while interleaving XCDs finely enough to balance the spatially varying ray cost.
__device__ __forceinline__ uint32_t xcd_swizzle(uint32_t b, uint32_t nb, uint32_t chunk)
{
    if (chunk == 0u) return b;
    const uint32_t group = 8u * chunk;
    const uint32_t full = (nb / group) * group; // blocks beyond the last full group keep their id
    if (b >= full) return b;
    const uint32_t g = b / group, r = b % group;
    const uint32_t xcd = r & 7u, k = r >> 3;      // k-th block this XCD receives inside the group
    return g * group + xcd * chunk + k;
}

// 64-bit hit key: (t bits, particle id, entry<exit) — decision (iv) total order
__device__ __forceinline__ uint64_t mk_key(float t, uint32_t id, uint32_t is_exit)
{
    return ((uint64_t)__float_as_uint(t) << 32) | (uint64_t)((id << 1) | is_exit);
}
__device__ __forceinline__ float key_t(uint64_t k) { return __uint_as_float((uint32_t)(k >> 32)); }
__device__ __forceinline__ uint32_t key_id(uint64_t k) { return ((uint32_t)k) >> 1; }
constexpr uint64_t kKeyInvalid = 0xFFFFFFFFFFFFFFFFull;

}  // namespace grt
