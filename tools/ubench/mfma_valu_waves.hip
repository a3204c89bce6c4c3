// Does one wave's VALU work hide under ANOTHER wave's MFMAs on the same SIMD (gfx950)?
// One workgroup of 512 threads per CU: waves 0-3 and waves 4-7 share SIMDs 0-3 (tools/ubench/simd_map.hip).  Waves 0-3 run MFMAs (4 independent
// chains), waves 4-7 run independent v_fma_f32 (or v_fma_mixlo_f16) -- each group alone, then both together.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu_waves tools/ubench/mfma_valu_waves.hip && /tmp/mfma_valu_waves
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <bool MIX>
__global__ __launch_bounds__(512) void k(float* out, long long* clk, int n_mfma, int n_valu)
{
    const int wave = threadIdx.x >> 6;
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (_Float16)(threadIdx.x * 0.001f + e); y[e] = (_Float16)(e * 0.5f); }
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = threadIdx.x * 0.5f + e;
    unsigned hh[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const float sc = 1.0009765625f;
    __syncthreads();
    const long long t0 = clock64();
    if (wave < 4) {
        for (int i = 0; i < n_mfma; ++i) {
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[a], 0, 0, 0);
        }
    } else {
        for (int i = 0; i < n_valu; ++i) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (MIX) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(hh[e]) : "v"(v[e]), "v"(sc));
                else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[e]) : "v"(sc));
            }
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int e = 0; e < 8; ++e) s += v[e] + (float)hh[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) clk[wave] = t1 - t0;
}
template <bool MIX>
void run(int n_mfma, int n_valu, const char* tag)
{
    float* out; long long* clk;
    hipMalloc(&out, sizeof(float) * 256 * 512); hipMalloc(&clk, 64);
    k<MIX><<<256, 512>>>(out, clk, n_mfma, n_valu); hipDeviceSynchronize();
    k<MIX><<<256, 512>>>(out, clk, n_mfma, n_valu); hipDeviceSynchronize();
    long long c[8]; hipMemcpy(c, clk, 64, hipMemcpyDeviceToHost);
    printf("%-46s MFMA wave: %7lld ticks (%.1f per MFMA)   VALU wave: %7lld ticks (%.1f per instruction)\n", tag, c[0], n_mfma ? (double)c[0] / (4.0 * n_mfma) : 0.0,
           c[4], n_valu ? (double)c[4] / (8.0 * n_valu) : 0.0);
    hipFree(out); hipFree(clk);
}
int main()
{
    run<false>(1000, 0, "MFMAs alone (4 chains)");
    run<false>(0, 2000, "v_fma_f32 alone");
    run<false>(1000, 2000, "both, v_fma_f32 (16 per 4 MFMAs)");
    run<false>(1000, 1000, "both, v_fma_f32 (8 per 4 MFMAs)");
    run<true>(0, 2000, "v_fma_mixlo_f16 alone");
    run<true>(1000, 1000, "both, v_fma_mixlo_f16 (8 per 4 MFMAs)");
    run<true>(1000, 500, "both, v_fma_mixlo_f16 (4 per 4 MFMAs)");
    return 0;
}
