// Can ONE wave hide its own VALU work in the shadow of its own MFMAs?  Loop body: 4 independent v_mfma_f32_32x32x16_f16, each followed by
// K independent VALU instructions (v_fma_f32 or v_fma_mixlo_f16); one wave per SIMD.  Prints cycles per MFMA for K = 0 .. 5.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mvo tools/ubench/mfma_valu_overlap.hip && /tmp/mvo
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int K, int MIX>
__global__ __launch_bounds__(256) void k(float* out, long long* clk, int iters)
{
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }
    f16x8 a, b;
    for (int r = 0; r < 8; ++r) { a[r] = (_Float16)(threadIdx.x * 0.001f); b[r] = (_Float16)0.5f; }
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
    const float sc = 1.0009765625f;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#define VAL(x) do { if (MIX == 1) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(x) : "v"(v0), "v"(sc)); else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(sc)); } while (0)
#define GROUP(c, x0, x1, x2, x3, x4) do { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); \
        if (MIX == 2) { if (K > 0) VAL(x0); if (K > 1) VAL(x0); if (K > 2) VAL(x0); if (K > 3) VAL(x0); if (K > 4) VAL(x0); } /* one dependent chain */ \
        else { if (K > 0) VAL(x0); if (K > 1) VAL(x1); if (K > 2) VAL(x2); if (K > 3) VAL(x3); if (K > 4) VAL(x4); } } while (0)
        GROUP(c0, v1, v2, v3, v4, v5); GROUP(c1, v6, v7, v1, v2, v3); GROUP(c2, v4, v5, v6, v7, v1); GROUP(c3, v2, v3, v4, v5, v6);
    }
    const long long t1 = clock64();
    float s = v1 + v2 + v3 + v4 + v5 + v6 + v7;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
template <int K, int MIX>
void run()
{
    float* out; long long* clk; const int iters = 20000;
    (void)hipMalloc(&out, sizeof(float) * 256 * 256); (void)hipMalloc(&clk, 8);
    k<K, MIX><<<256, 256>>>(out, clk, iters); (void)hipDeviceSynchronize();
    k<K, MIX><<<256, 256>>>(out, clk, iters); (void)hipDeviceSynchronize();
    long long c; (void)hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    printf("  K=%d: %.1f", K, (double)c / (iters * 4.0));
    (void)hipFree(out); (void)hipFree(clk);
}
int main()
{
    printf("cycles per MFMA, K x v_fma_f32 after each:      "); run<0, 0>(); run<1, 0>(); run<2, 0>(); run<3, 0>(); run<4, 0>(); run<5, 0>(); printf("\n");
    printf("cycles per MFMA, K DEPENDENT v_fma_f32 after each:  "); run<0, 2>(); run<1, 2>(); run<2, 2>(); run<3, 2>(); run<4, 2>(); run<5, 2>(); printf("\n");
    printf("cycles per MFMA, K x v_fma_mixlo_f16 after each:"); run<0, 1>(); run<1, 1>(); run<2, 1>(); run<3, 1>(); run<4, 1>(); run<5, 1>(); printf("\n");
    return 0;
}
