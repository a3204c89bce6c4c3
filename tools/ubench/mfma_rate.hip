// MFMA issue-rate / clock microbenchmark (gfx950): waves of independent v_mfma_f32_32x32x16_f16 chains.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate tools/ubench/mfma_rate.hip && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, long long* clk, int iters)
{
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (_Float16)(threadIdx.x * 0.001f + e); y[e] = (_Float16)(e * 0.5f); }
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[a], 0, 0, 0);
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
template <int NACC>
void run(int blocks, int iters, const char* tag)
{
    float* out; long long* clk;
    hipMalloc(&out, sizeof(float) * blocks * 256); hipMalloc(&clk, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, 256>>>(out, clk, iters); hipDeviceSynchronize();
    hipEventRecord(e0); k<NACC><<<blocks, 256>>>(out, clk, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double mfma_per_wave = (double)iters * NACC;
    const double flop = mfma_per_wave * 4 * blocks * 32.0 * 32 * 16 * 2;
    printf("%-34s blocks %5d: %.1f us, %lld clk64 ticks (%.1f per MFMA per wave), %.0f TFLOP/s, clk64 rate %.2f GHz\n", tag, blocks, ms * 1e3, c,
           (double)c / mfma_per_wave, flop / (ms * 1e-3) / 1e12, (double)c / (ms * 1e-3) / 1e9);
    hipFree(out); hipFree(clk);
}
int main()
{
    run<4>(256, 4000, "4 acc, 1 WG/CU (1 wave/SIMD)");
    run<4>(512, 4000, "4 acc, 2 WG/CU (2 waves/SIMD)");
    run<1>(256, 16000, "1 acc (dependent), 1 wave/SIMD");
    run<2>(256, 8000, "2 acc, 1 wave/SIMD");
    run<4>(1024, 4000, "4 acc, 4 WG/CU");
    return 0;
}
