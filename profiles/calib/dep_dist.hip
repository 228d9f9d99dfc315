// How far apart must dependent wave64 VALU instructions be for full issue rate on gfx950?  N interleaved dependent chains.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
template <int N>
__global__ __launch_bounds__(64) void k(unsigned long long* out, int reps)
{
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int r = 0; r < reps; r++) {
        if (N == 1) asm volatile(".rept 64\nv_fma_f32 %0, %0, %0, %0\n.endr" : "+v"(a0)::"memory");
        if (N == 2) asm volatile(".rept 32\nv_fma_f32 %0, %0, %0, %0\nv_fma_f32 %1, %1, %1, %1\n.endr" : "+v"(a0), "+v"(a1)::"memory");
        if (N == 3) asm volatile(".rept 21\nv_fma_f32 %0, %0, %0, %0\nv_fma_f32 %1, %1, %1, %1\nv_fma_f32 %2, %2, %2, %2\n.endr\nv_fma_f32 %0, %0, %0, %0" : "+v"(a0), "+v"(a1), "+v"(a2)::"memory");
        if (N == 4) asm volatile(".rept 16\nv_fma_f32 %0, %0, %0, %0\nv_fma_f32 %1, %1, %1, %1\nv_fma_f32 %2, %2, %2, %2\nv_fma_f32 %3, %3, %3, %3\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)::"memory");
        if (N == 8) asm volatile(".rept 8\nv_fma_f32 %0, %0, %0, %0\nv_fma_f32 %1, %1, %1, %1\nv_fma_f32 %2, %2, %2, %2\nv_fma_f32 %3, %3, %3, %3\n"
                                 "v_fma_f32 %4, %4, %4, %4\nv_fma_f32 %5, %5, %5, %5\nv_fma_f32 %6, %6, %6, %6\nv_fma_f32 %7, %7, %7, %7\n.endr"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)::"memory");
        // 9: dependent through VCC: v_cmp writes vcc, v_cndmask reads it (the pattern of the slab loop)
        if (N == 9) asm volatile(".rept 32\nv_cmp_lt_f32 vcc, %0, %1\nv_cndmask_b32 %0, %0, %1, vcc\n.endr" : "+v"(a0), "+v"(a1)::"vcc", "memory");
        // 10: the same with an independent instruction between the compare and the select
        if (N == 10) asm volatile(".rept 21\nv_cmp_lt_f32 vcc, %0, %1\nv_fma_f32 %2, %2, %2, %2\nv_cndmask_b32 %0, %0, %1, vcc\n.endr\nv_fma_f32 %2, %2, %2, %2" : "+v"(a0), "+v"(a1), "+v"(a2)::"vcc", "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[0] = 0;
}
int main()
{
    unsigned long long* d; hipMalloc(&d, 8 * 8192);
    std::vector<unsigned long long> h(8192);
    typedef void (*K)(unsigned long long*, int);
    K ks[7] = {k<1>, k<2>, k<3>, k<4>, k<8>, k<9>, k<10>};
    const char* names[7] = {"1 chain", "2 chains", "3 chains", "4 chains", "8 chains", "cmp->cndmask via vcc", "cmp, fma, cndmask"};
    for (int m = 0; m < 7; m++)
        for (int W : {1, 2, 4, 5, 8}) {
            int grid = 1024 * W;
            hipLaunchKernelGGL(ks[m], dim3(grid), dim3(64), 0, 0, d, 20);
            hipLaunchKernelGGL(ks[m], dim3(grid), dim3(64), 0, 0, d, 200);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d, 8 * grid, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.begin() + grid);
            double med = (double)h[grid / 2] / (200.0 * 64);
            printf("%-24s W=%d  cyc/inst/wave %6.3f  IPC/SIMD %.4f\n", names[m], W, med, W / med);
        }
    return 0;
}
