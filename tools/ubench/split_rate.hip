// The f16 hi / lo split of the policy tiles next to the MFMAs it feeds (gfx950): per 3 MFMAs (two accumulator chains) the split of 4 value pairs
// (= one K-chunk per 6 MFMAs, what k_pass does), written (a) with v_fma_mixlo/mixhi_f16, 4 instructions per pair (the tiles' split_pair) and
// (b) with plain VALU: v_pk_mul_f32, v_cvt_pk_f16_f32, 2 x v_cvt_f32_f16, v_pk_add_f32 (neg), v_cvt_pk_f16_f32 -- 6 per pair, the same bits.
// W = 1 or 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o /tmp/split_rate tools/ubench/split_rate.hip && /tmp/split_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ inline void split_mix(float x0, float x1, float sc, unsigned& hi, unsigned& lo)
{
    unsigned h = 0, l = 0;
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(sc));
    asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(sc));
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(sc), "v"(h));
    asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(sc), "v"(h));
    hi = h; lo = l;
}
__device__ inline void split_plain(float x0, float x1, float sc, unsigned& hi, unsigned& lo)
{
    f32x2 t, hf, d, x = {x0, x1}, s2 = {sc, sc};
    unsigned h, l;
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(x), "v"(s2));
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(t.x), "v"(t.y));
    asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(hf.x) : "v"(h));
    asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(hf.y) : "v"(h));
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(t), "v"(hf));
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l) : "v"(d.x), "v"(d.y));
    hi = h; lo = l;
}
template <int MODE, int T>   // MODE 0: MFMAs only; 1: + mix split; 2: + plain split
__global__ __launch_bounds__(T) void k(float* out, long long* clk, unsigned* bits, int iters)
{
    f32x16 a0, a1;
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
    f16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (_Float16)(threadIdx.x * 0.001f + e); y[e] = (_Float16)(e * 0.5f); }
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = (threadIdx.x * 37 % 101) * 13.37f + e * 0.123f - 300.f;
    const float sc = 0.5f;
    unsigned H[4] = {0, 0, 0, 0}, L[4] = {0, 0, 0, 0};
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
        if (MODE == 1) split_mix(v[0], v[1], sc, H[0], L[0]); else if (MODE == 2) split_plain(v[0], v[1], sc, H[0], L[0]);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a1, 0, 0, 0);
        if (MODE == 1) { split_mix(v[2], v[3], sc, H[1], L[1]); split_mix(v[4], v[5], sc, H[2], L[2]); } else if (MODE == 2) { split_plain(v[2], v[3], sc, H[1], L[1]); split_plain(v[4], v[5], sc, H[2], L[2]); }
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
        if (MODE == 1) split_mix(v[6], v[7], sc, H[3], L[3]); else if (MODE == 2) split_plain(v[6], v[7], sc, H[3], L[3]);
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0) { for (int q = 0; q < 4; ++q) { bits[(threadIdx.x * 4 + q) * 2] = H[q]; bits[(threadIdx.x * 4 + q) * 2 + 1] = L[q]; } }
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
template <int MODE, int T>
long long run(int iters, unsigned* host_bits)
{
    float* out; long long* clk; unsigned* bits;
    hipMalloc(&out, sizeof(float) * 256 * T); hipMalloc(&clk, 8); hipMalloc(&bits, sizeof(unsigned) * T * 8);
    k<MODE, T><<<256, T>>>(out, clk, bits, iters); hipDeviceSynchronize();
    k<MODE, T><<<256, T>>>(out, clk, bits, iters); hipDeviceSynchronize();
    long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    if (host_bits) hipMemcpy(host_bits, bits, sizeof(unsigned) * 256 * 8, hipMemcpyDeviceToHost);
    hipFree(out); hipFree(clk); hipFree(bits);
    return c;
}
int main()
{
    static unsigned b1[2048], b2[2048];
    const int it = 2000;
    printf("1 wave per SIMD:  MFMAs only %.1f cycles per MFMA | + v_fma_mix split %.1f | + plain-VALU split %.1f\n", run<0, 256>(it, nullptr) / (3.0 * it), run<1, 256>(it, b1) / (3.0 * it),
           run<2, 256>(it, b2) / (3.0 * it));
    int diff = 0;
    for (int i = 0; i < 2048; ++i) diff += b1[i] != b2[i];
    printf("split bits (256 lanes x 4 pairs x hi, lo): %d of 2048 words differ between the two sequences\n", diff);
    printf("2 waves per SIMD: MFMAs only %.1f pipe cycles per MFMA | + v_fma_mix split %.1f | + plain-VALU split %.1f\n", run<0, 512>(it, nullptr) / (6.0 * it), run<1, 512>(it, nullptr) / (6.0 * it),
           run<2, 512>(it, nullptr) / (6.0 * it));
    return 0;
}
