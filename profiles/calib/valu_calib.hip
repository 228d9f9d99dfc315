// valu_calib.hip — issue-rate calibration microbenchmark for gfx950 (MI355X).
//
// Question it answers ("fix the roofline story"): what does a SIMD sustain in wave64 VALU
// instructions per clock at 1/2/4/8 resident waves, i.e. what value of  n_waves x SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES
// means "100 % VALU busy"?  Also prices the instruction kinds the render kernel is made of.
//
// Each workgroup is ONE wave64; the grid is 1024 x W workgroups (256 CUs x 4 SIMDs x W), so W waves share a SIMD.
// Every wave runs REPS x UNROLL copies of one instruction (independent register chains) between two s_memtime stamps;
// output: median over waves of cycles / instruction seen by ONE wave, and W / that = instructions per cycle per SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_calib valu_calib.hip     Run: ./valu_calib > valu_calib.json
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define REPS 200
#define UNROLL 64

#define BODY(NAME, ASM, ...)                                                                       \
    __global__ __launch_bounds__(64) void NAME(unsigned long long* out, int reps)                  \
    {                                                                                               \
        float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f; \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                          \
        for (int r = 0; r < reps; r++) {                                                            \
            asm volatile(".rept 8\n" ASM "\n.endr"                                                  \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)::__VA_ARGS__); \
        }                                                                                           \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                          \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                            \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[0] = 0;                        \
    }

// 8 instructions per .rept body -> 64 per loop trip
BODY(k_fma, "v_fma_f32 %0, %0, %0, %0\nv_fma_f32 %1, %1, %1, %1\nv_fma_f32 %2, %2, %2, %2\nv_fma_f32 %3, %3, %3, %3\n"
            "v_fma_f32 %4, %4, %4, %4\nv_fma_f32 %5, %5, %5, %5\nv_fma_f32 %6, %6, %6, %6\nv_fma_f32 %7, %7, %7, %7", "memory")
BODY(k_fma_dep, "v_fma_f32 %0, %0, %0, %0\nv_fma_f32 %0, %0, %0, %0\nv_fma_f32 %0, %0, %0, %0\nv_fma_f32 %0, %0, %0, %0\n"
                "v_fma_f32 %0, %0, %0, %0\nv_fma_f32 %0, %0, %0, %0\nv_fma_f32 %0, %0, %0, %0\nv_fma_f32 %0, %0, %0, %0", "memory")
BODY(k_min_u32, "v_min_u32 %0, %0, %1\nv_min_u32 %1, %1, %2\nv_min_u32 %2, %2, %3\nv_min_u32 %3, %3, %4\n"
                "v_min_u32 %4, %4, %5\nv_min_u32 %5, %5, %6\nv_min_u32 %6, %6, %7\nv_min_u32 %7, %7, %0", "memory")
BODY(k_min_dpp, "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                "v_min_u32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                "v_min_u32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                "v_min_u32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                "v_min_u32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                "v_min_u32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                "v_min_u32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                "v_min_u32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1", "memory")
BODY(k_exp, "v_exp_f32 %0, %0\nv_exp_f32 %1, %1\nv_exp_f32 %2, %2\nv_exp_f32 %3, %3\n"
            "v_exp_f32 %4, %4\nv_exp_f32 %5, %5\nv_exp_f32 %6, %6\nv_exp_f32 %7, %7", "memory")
BODY(k_salu, "s_add_u32 s20, s20, s21\ns_add_u32 s21, s21, s22\ns_add_u32 s22, s22, s23\ns_add_u32 s23, s23, s20\n"
             "s_add_u32 s20, s20, s21\ns_add_u32 s21, s21, s22\ns_add_u32 s22, s22, s23\ns_add_u32 s23, s23, s20",
     "s20", "s21", "s22", "s23", "scc", "memory")
BODY(k_mix, "v_fma_f32 %0, %0, %0, %0\ns_add_u32 s20, s20, s21\nv_fma_f32 %1, %1, %1, %1\ns_add_u32 s21, s21, s22\n"
            "v_fma_f32 %2, %2, %2, %2\ns_add_u32 s22, s22, s23\nv_fma_f32 %3, %3, %3, %3\ns_add_u32 s23, s23, s20",
     "s20", "s21", "s22", "s23", "scc", "memory")
BODY(k_readlane, "v_readlane_b32 s20, %0, 3\nv_readlane_b32 s21, %1, 5\nv_readlane_b32 s22, %2, 7\nv_readlane_b32 s23, %3, 9\n"
                 "v_readlane_b32 s20, %4, 3\nv_readlane_b32 s21, %5, 5\nv_readlane_b32 s22, %6, 7\nv_readlane_b32 s23, %7, 9",
     "s20", "s21", "s22", "s23", "memory")

// 64-bit register pairs: v_pk_fma_f32, v_mov_b64, v_cmp_lt_u64 need aligned pairs -> separate kernel with double operands
#define BODY64(NAME, ASM, ...)                                                                     \
    __global__ __launch_bounds__(64) void NAME(unsigned long long* out, int reps)                  \
    {                                                                                               \
        double a0 = threadIdx.x, a1 = 1., a2 = 2., a3 = 3., a4 = 4., a5 = 5., a6 = 6., a7 = 7.;     \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                          \
        for (int r = 0; r < reps; r++) {                                                            \
            asm volatile(".rept 8\n" ASM "\n.endr"                                                  \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)::__VA_ARGS__); \
        }                                                                                           \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                          \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                            \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678) out[0] = 0;                         \
    }
BODY64(k_pk_fma, "v_pk_fma_f32 %0, %0, %0, %0\nv_pk_fma_f32 %1, %1, %1, %1\nv_pk_fma_f32 %2, %2, %2, %2\nv_pk_fma_f32 %3, %3, %3, %3\n"
                 "v_pk_fma_f32 %4, %4, %4, %4\nv_pk_fma_f32 %5, %5, %5, %5\nv_pk_fma_f32 %6, %6, %6, %6\nv_pk_fma_f32 %7, %7, %7, %7", "memory")
BODY64(k_pk_mul, "v_pk_mul_f32 %0, %0, %0\nv_pk_mul_f32 %1, %1, %1\nv_pk_mul_f32 %2, %2, %2\nv_pk_mul_f32 %3, %3, %3\n"
                 "v_pk_mul_f32 %4, %4, %4\nv_pk_mul_f32 %5, %5, %5\nv_pk_mul_f32 %6, %6, %6\nv_pk_mul_f32 %7, %7, %7", "memory")
BODY64(k_mov_b64, "v_mov_b64 %0, %1\nv_mov_b64 %1, %2\nv_mov_b64 %2, %3\nv_mov_b64 %3, %4\n"
                  "v_mov_b64 %4, %5\nv_mov_b64 %5, %6\nv_mov_b64 %6, %7\nv_mov_b64 %7, %0", "memory")
BODY64(k_cmp64, "v_cmp_lt_u64 vcc, %0, %1\nv_cmp_lt_u64 vcc, %1, %2\nv_cmp_lt_u64 vcc, %2, %3\nv_cmp_lt_u64 vcc, %3, %4\n"
                "v_cmp_lt_u64 vcc, %4, %5\nv_cmp_lt_u64 vcc, %5, %6\nv_cmp_lt_u64 vcc, %6, %7\nv_cmp_lt_u64 vcc, %7, %0", "vcc", "memory")
BODY64(k_fma_f64, "v_fma_f64 %0, %0, %0, %0\nv_fma_f64 %1, %1, %1, %1\nv_fma_f64 %2, %2, %2, %2\nv_fma_f64 %3, %3, %3, %3\n"
                  "v_fma_f64 %4, %4, %4, %4\nv_fma_f64 %5, %5, %5, %5\nv_fma_f64 %6, %6, %6, %6\nv_fma_f64 %7, %7, %7, %7", "memory")

typedef void (*K)(unsigned long long*, int);
struct Case { const char* name; K k; };

int main()
{
    const Case cases[] = {{"v_fma_f32", k_fma}, {"v_fma_f32_dependent", k_fma_dep}, {"v_min_u32", k_min_u32},
                          {"v_min_u32_dpp", k_min_dpp}, {"v_exp_f32", k_exp}, {"s_add_u32", k_salu},
                          {"v_fma_f32+s_add_u32 interleaved (per pair)", k_mix}, {"v_readlane_b32", k_readlane},
                          {"v_pk_fma_f32", k_pk_fma}, {"v_pk_mul_f32", k_pk_mul}, {"v_mov_b64", k_mov_b64},
                          {"v_cmp_lt_u64", k_cmp64}, {"v_fma_f64", k_fma_f64}};
    unsigned long long* d = nullptr;
    const int maxg = 1024 * 8;
    if (hipMalloc(&d, sizeof(unsigned long long) * maxg) != hipSuccess) { fprintf(stderr, "no GPU\n"); return 1; }
    std::vector<unsigned long long> h(maxg);
    printf("{\"device\": \"gfx950\", \"unit\": \"cycles per wave64 instruction, median over waves (s_memtime)\", \"cases\": [\n");
    bool first = true;
    for (const Case& c : cases) {
        for (int W : {1, 2, 4, 8}) {
            const int grid = 1024 * W;
            hipLaunchKernelGGL(c.k, dim3(grid), dim3(64), 0, 0, d, 20); // warm-up
            hipLaunchKernelGGL(c.k, dim3(grid), dim3(64), 0, 0, d, REPS);
            if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 2; }
            hipMemcpy(h.data(), d, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.begin() + grid);
            const double n = (double)REPS * UNROLL * (strstr(c.name, "per pair") ? 0.5 : 1.0);
            const double med = (double)h[grid / 2] / n, lo = (double)h[grid / 20] / n, hi = (double)h[grid - 1 - grid / 20] / n;
            printf("%s {\"inst\": \"%s\", \"waves_per_simd\": %d, \"cyc_per_inst_one_wave\": %.3f, \"p5\": %.3f, \"p95\": %.3f, "
                   "\"inst_per_cycle_per_simd\": %.4f}", first ? " " : ",\n ", c.name, W, med, lo, hi, W / med);
            first = false;
        }
    }
    printf("\n]}\n");
    hipFree(d);
    return 0;
}
