// How many waves with how many dependent accumulator chains each does it take to fill a SIMD's matrix pipe (gfx950)?
// One workgroup per CU of 4 x W waves (W waves per SIMD), every wave runs NACC independent chains of v_mfma_f32_32x32x16_f16.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_waves tools/ubench/mfma_waves.hip && /tmp/mfma_waves
// Reported: matrix-pipe cycles per MFMA = wave-0 clock ticks of the loop / (MFMAs issued by the W waves of its SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int NACC, int T>
__global__ __launch_bounds__(T) void k(float* out, long long* clk, int iters)
{
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (_Float16)(threadIdx.x * 0.001f + e); y[e] = (_Float16)(e * 0.5f); }
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[a], 0, 0, 0);
    }
    __syncthreads();
    const long long t1 = clock64();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
template <int NACC, int T>
void run(int iters)
{
    float* out; long long* clk;
    hipMalloc(&out, sizeof(float) * 256 * T); hipMalloc(&clk, 8);
    k<NACC, T><<<256, T>>>(out, clk, iters); hipDeviceSynchronize();
    k<NACC, T><<<256, T>>>(out, clk, iters); hipDeviceSynchronize();
    long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const int W = T / 256;
    printf("%d wave(s) per SIMD x %d chain(s) per wave: %.1f pipe cycles per MFMA (%.1f per MFMA and wave)\n", W, NACC, (double)c / ((double)iters * NACC * W),
           (double)c / ((double)iters * NACC));
    hipFree(out); hipFree(clk);
}
int main()
{
    run<1, 256>(8000); run<2, 256>(4000); run<4, 256>(2000);
    run<1, 512>(8000); run<2, 512>(4000); run<4, 512>(2000);
    run<1, 1024>(8000); run<2, 1024>(4000); run<3, 1024>(3000);
    return 0;
}
