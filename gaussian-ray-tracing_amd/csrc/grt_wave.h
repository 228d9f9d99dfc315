// grt_wave.h — wave64-level building blocks shared by the wave-cooperative render kernels (gfx950 only):
// scalar (constant-address-space) record fetches, DPP min reductions, lane-mask votes, slot-key packing.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "grt_device.h"

namespace grt {
namespace {

constexpr uint64_t kCellMask = 31ull; // payload-cell bits of a slot key
__device__ __forceinline__ uint64_t mk_skey(float t, uint32_t id, uint32_t is_exit)
{
    return ((uint64_t)__float_as_uint(t) << 32) | (uint64_t)((id << 6) | (is_exit << 5));
}
__device__ __forceinline__ uint32_t skey_id(uint64_t k) { return ((uint32_t)k) >> 6; }

struct Cnt {
    uint32_t rays = 0, segments = 0, hit_evals = 0, rounds = 0, node_visits = 0, proxy_tests = 0, fetches = 0, stall_exits = 0;
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

// one 64-B record by ONE scalar load at base + a 32-bit byte offset (s_load_dwordx16 sdst, sbase, soffset)
typedef float v16f __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void sload64(const float4* base, uint32_t byte_off, float4& q0, float4& q1, float4& q2, float4& q3)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) char* cptr1;
    typedef const __attribute__((address_space(4))) v16f* cptr16;
    const v16f v = *(cptr16)((cptr1)(uintptr_t)base + byte_off);
    q0 = make_float4(v.s0, v.s1, v.s2, v.s3); q1 = make_float4(v.s4, v.s5, v.s6, v.s7);
    q2 = make_float4(v.s8, v.s9, v.sa, v.sb); q3 = make_float4(v.sc, v.sd, v.se, v.sf);
#else
    const float4* p = (const float4*)((const char*)base + byte_off);
    q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = p[3];
#endif
}
// x with bit b cleared, b wave-uniform (s_bitset0_b64: one SALU operation instead of the three of x & (x - 1))
__device__ __forceinline__ uint64_t clear_bit64(uint64_t x, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm("s_bitset0_b64 %0, %1" : "+s"(x) : "s"(b));
    return x;
#else
    return x & ~(1ull << b);
#endif
}

// wave64 min of non-negative floats (or +inf) -> wave-uniform value.  Their bit patterns order like unsigned
// integers, so the reduction is 4 v_min_u32 with DPP operands inside rows of 16, then 4 v_readlane + 3 s_min_u32
// (no NaN canonicalisation, which fminf would add to every step).
__device__ __forceinline__ float wave_min(float v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t x = __float_as_uint(v);
    // bound_ctrl:1 lets the compiler fold each DPP move into the v_min_u32 itself (one VALU op per step)
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, true)); // row_half_mirror
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xF, 0xF, true)); // row_mirror
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)x, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)x, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)x, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)x, 48);
    return __uint_as_float(min(min(a, b), min(c, d)));
#else
    return v;
#endif
}

// four independent wave minima in lockstep: the DPP steps of different reductions interleave, so the two wait states a
// DPP operand needs after its producer are filled with useful work instead of s_nop
__device__ __forceinline__ void wave_min4(float a, float b, float c, float d, float& ra, float& rb, float& rc, float& rd)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t x0 = __float_as_uint(a), x1 = __float_as_uint(b), x2 = __float_as_uint(c), x3 = __float_as_uint(d);
#define GRT_DPP_STEP(CTRL)                                                                                 \
    {                                                                                                      \
        const uint32_t y0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x0, CTRL, 0xF, 0xF, true);       \
        const uint32_t y1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x1, CTRL, 0xF, 0xF, true);       \
        const uint32_t y2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x2, CTRL, 0xF, 0xF, true);       \
        const uint32_t y3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x3, CTRL, 0xF, 0xF, true);       \
        x0 = min(x0, y0); x1 = min(x1, y1); x2 = min(x2, y2); x3 = min(x3, y3);                            \
    }
    GRT_DPP_STEP(0xB1)  // quad_perm [1,0,3,2]
    GRT_DPP_STEP(0x4E)  // quad_perm [2,3,0,1]
    GRT_DPP_STEP(0x141) // row_half_mirror
    GRT_DPP_STEP(0x140) // row_mirror
#undef GRT_DPP_STEP
#define GRT_ROWS(x)                                                                                        \
    __uint_as_float(min(min((uint32_t)__builtin_amdgcn_readlane((int)x, 0), (uint32_t)__builtin_amdgcn_readlane((int)x, 16)), \
                        min((uint32_t)__builtin_amdgcn_readlane((int)x, 32), (uint32_t)__builtin_amdgcn_readlane((int)x, 48))))
    ra = GRT_ROWS(x0); rb = GRT_ROWS(x1); rc = GRT_ROWS(x2); rd = GRT_ROWS(x3);
#undef GRT_ROWS
#else
    ra = a; rb = b; rc = c; rd = d;
#endif
}

// votes straight on the lane mask (the __any/__ballot wrappers go through an int and cost two extra VALU ops)
__device__ __forceinline__ uint64_t wave_ballot(bool p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(p);
#else
    return p ? 1ull : 0ull;
#endif
}
__device__ __forceinline__ bool wave_any(bool p) { return wave_ballot(p) != 0ull; }
// votes on ONE compare, as the lane mask the compare instruction writes (v_cmp -> SGPR pair): whatever else uses the
// condition, and wherever it was computed, the vote costs one VALU operation (wave_ballot of a bool that arrives from
// another block costs two on top of its compare)
__device__ __forceinline__ uint64_t vote_eq_u32(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_uicmp(a, b, 32); // ICMP_EQ
#else
    return a == b;
#endif
}
__device__ __forceinline__ uint64_t vote_lt_f32(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fcmpf(a, b, 4); // FCMP_OLT (fcmpf: the float form; fcmp is the double one)
#else
    return a < b;
#endif
}
// max(v, +0) for a non-NaN float as one integer max (negative floats are negative ints)
__device__ __forceinline__ float clamp0(float v) { return __int_as_float(max(__float_as_int(v), 0)); }

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); } // v_pk_fma_f32

// two wave-uniform floats as one 64-bit scalar (an aligned SGPR pair) and back
__device__ __forceinline__ uint64_t pack2(float a, float b)
{
    return (uint64_t)__float_as_uint(a) | ((uint64_t)__float_as_uint(b) << 32);
}
__device__ __forceinline__ v2f unpack2(uint64_t u)
{
    return v2f{__uint_as_float((uint32_t)u), __uint_as_float((uint32_t)(u >> 32))};
}

} // namespace
} // namespace grt
