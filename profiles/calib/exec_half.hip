// Does a wave64 VALU instruction whose EXEC mask covers only one 32-lane half issue faster on gfx950 (SIMD-32, two passes)?
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(64) void k(unsigned long long* out, int reps)
{
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (MODE == 1) asm volatile("s_mov_b64 exec, 0x00000000ffffffff" ::: "memory");
    if (MODE == 2) asm volatile("s_mov_b32 exec_lo, 0x0000ffff\ns_mov_b32 exec_hi, 0x0000ffff" ::: "memory");
    if (MODE == 3) asm volatile("s_mov_b32 exec_lo, 0\ns_mov_b32 exec_hi, 0xffffffff" ::: "memory");
    if (MODE == 4) asm volatile("s_mov_b64 exec, 1" ::: "memory");
    for (int r = 0; r < reps; r++) {
        asm volatile(".rept 8\n"
                     "v_fma_f32 %0, %0, %0, %0\nv_fma_f32 %1, %1, %1, %1\nv_fma_f32 %2, %2, %2, %2\nv_fma_f32 %3, %3, %3, %3\n"
                     "v_fma_f32 %4, %4, %4, %4\nv_fma_f32 %5, %5, %5, %5\nv_fma_f32 %6, %6, %6, %6\nv_fma_f32 %7, %7, %7, %7\n.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)::"memory");
    }
    asm volatile("s_mov_b64 exec, -1" ::: "memory");
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
    K ks[5] = {k<0>, k<1>, k<2>, k<3>, k<4>};
    const char* names[5] = {"exec full", "exec low half", "exec 16+16 (both halves partly)", "exec high half", "exec one lane"};
    for (int m = 0; m < 5; m++)
        for (int W : {1, 4, 8}) {
            int grid = 1024 * W;
            hipLaunchKernelGGL(ks[m], dim3(grid), dim3(64), 0, 0, d, 20);
            hipLaunchKernelGGL(ks[m], dim3(grid), dim3(64), 0, 0, d, 200);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d, 8 * grid, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.begin() + grid);
            double med = (double)h[grid / 2] / (200.0 * 64);
            printf("%-34s W=%d  cyc/inst/wave %.3f  IPC/SIMD %.4f\n", names[m], W, med, W / med);
        }
    return 0;
}
