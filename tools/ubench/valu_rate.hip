// VALU issue cost of the instructions the f16 split is made of (gfx950) with 1, 2 and 4 waves per SIMD, independent chains.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/ubench/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(X) X X X X X X X X
template <int OP>
__global__ __launch_bounds__(1024) void k(float* out, long long* clk, int iters)
{
    float a0 = threadIdx.x * 0.001f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float sc = 1.0009765625f;
    unsigned u0 = 1, u1 = 2, u2 = 3, u3 = 4, u4 = 5, u5 = 6, u6 = 7, u7 = 8;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sc)); }
        if (OP == 1) { asm volatile("v_fma_mixlo_f16 %0, %8, %9, 0\n v_fma_mixlo_f16 %1, %8, %9, 0\n v_fma_mixlo_f16 %2, %8, %9, 0\n v_fma_mixlo_f16 %3, %8, %9, 0\n v_fma_mixlo_f16 %4, %8, %9, 0\n v_fma_mixlo_f16 %5, %8, %9, 0\n v_fma_mixlo_f16 %6, %8, %9, 0\n v_fma_mixlo_f16 %7, %8, %9, 0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(a0), "v"(sc)); }
        if (OP == 2) { asm volatile("v_cvt_pkrtz_f16_f32 %0, %8, %9\n v_cvt_pkrtz_f16_f32 %1, %8, %9\n v_cvt_pkrtz_f16_f32 %2, %8, %9\n v_cvt_pkrtz_f16_f32 %3, %8, %9\n v_cvt_pkrtz_f16_f32 %4, %8, %9\n v_cvt_pkrtz_f16_f32 %5, %8, %9\n v_cvt_pkrtz_f16_f32 %6, %8, %9\n v_cvt_pkrtz_f16_f32 %7, %8, %9" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(a0), "v"(sc)); }
        if (OP == 3) { asm volatile("v_cvt_f32_f16 %0, %8\n v_cvt_f32_f16 %1, %8\n v_cvt_f32_f16 %2, %8\n v_cvt_f32_f16 %3, %8\n v_cvt_f32_f16 %4, %8\n v_cvt_f32_f16 %5, %8\n v_cvt_f32_f16 %6, %8\n v_cvt_f32_f16 %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(u0)); }
        if (OP == 4) { asm volatile("v_max3_f32 %0, %0, %8, %1\n v_max3_f32 %1, %1, %8, %2\n v_max3_f32 %2, %2, %8, %3\n v_max3_f32 %3, %3, %8, %4\n v_max3_f32 %4, %4, %8, %5\n v_max3_f32 %5, %5, %8, %6\n v_max3_f32 %6, %6, %8, %7\n v_max3_f32 %7, %7, %8, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sc)); }
        if (OP == 5) { asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sc)); }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7);
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
// waves per SIMD = threads / 256 (one workgroup per CU: 100 KB of LDS each; waves i, i + 4, ... share a SIMD: simd_map.hip)
template <int OP>
void run(const char* tag)
{
    float* out; long long* clk; const int iters = 20000;
    (void)hipMalloc(&out, sizeof(float) * 256 * 1024); (void)hipMalloc(&clk, 8);
    printf("%-22s", tag);
    for (int threads = 256; threads <= 1024; threads *= 2) {
        k<OP><<<256, threads, 100 * 1024>>>(out, clk, iters); (void)hipDeviceSynchronize();
        k<OP><<<256, threads, 100 * 1024>>>(out, clk, iters); (void)hipDeviceSynchronize();
        long long c; (void)hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        printf("  %d wave%s/SIMD: %.2f cycles per instruction of a wave, %.2f per SIMD", threads / 256, threads > 256 ? "s" : "", (double)c / (iters * 8.0), (double)c / (iters * 8.0) / (threads / 256));
    }
    printf("\n");
    (void)hipFree(out); (void)hipFree(clk);
}
int main()
{
    run<0>("v_fma_f32"); run<5>("v_mul_f32"); run<4>("v_max3_f32"); run<1>("v_fma_mixlo_f16"); run<2>("v_cvt_pkrtz_f16_f32"); run<3>("v_cvt_f32_f16");
    return 0;
}
