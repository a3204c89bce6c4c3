// Would 16-row half tiles (v_mfma_f32_16x16x32_f16) keep the canonical arithmetic of the 32-row tiles (v_mfma_f32_32x32x16_f16)?  (VERDICT r05 next #9,
// DESIGN.md section 9.)  The same dot products -- C[feature][row] += sum_k A[feature][k] * B[k][row], k = 0 .. 31 -- computed both ways:
//   (a) two v_mfma_f32_32x32x16_f16 in a row (k 0..15, then k 16..31 into the same accumulator): what every policy kernel does today;
//   (b) ONE v_mfma_f32_16x16x32_f16 per 16 x 16 block of the same 32 x 32 outputs (k 0..31 inside the instruction).
// Reported: how many of the 32 x 32 f32 results differ in bits, the largest difference in ulps, and the matrix pipe's cost of either per
// 32 features x 32 k of ONE 16-row half tile.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shapes tools/ubench/mfma_shapes.hip && /tmp/mfma_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A[trial][32 features][KK k], B[trial][KK k][32 rows] as f16 in memory; out32 / out16: C[trial][32][32] f32.  KK = 96: six chained 32x32x16 steps
// (a tile's K loop: three partial products of two 16-k chunks) against three chained 16x16x32 steps per 16 x 16 block, C starting from a non-zero value.
constexpr int KK = 96;
__global__ void k_values(const _Float16* A, const _Float16* B, float* out32, float* out16)
{
    const int t = blockIdx.x;
    A += (size_t)t * 32 * KK; B += (size_t)t * KK * 32; out32 += (size_t)t * 1024; out16 += (size_t)t * 1024;
    const int lane = threadIdx.x, l32 = lane & 31, h = lane >> 5;
    // ---- (a) 32x32x16: lane (l32, h) holds A[l32][16 s + 8 h ..] and B[16 s + 8 h ..][l32]; C register r of half h = feature (r & 3) + 8 (r >> 2) + 4 h, row l32
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.25f * (float)((r + l32 + t) % 7) - 0.5f;
    for (int step = 0; step < KK / 16; ++step) {
        f16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = A[l32 * KK + 16 * step + 8 * h + i]; b[i] = B[(16 * step + 8 * h + i) * 32 + l32]; }
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) out32[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l32] = c[r];
    // ---- (b) 16x16x32: lane (l16 = lane & 15, g = lane >> 4) holds A[f0 + l16][32 s + 8 g ..] and B[32 s + 8 g ..][r0 + l16]; C register r = feature f0 + 4 g + r, row r0 + l16
    const int l16 = lane & 15, g = lane >> 4;
    for (int f0 = 0; f0 < 32; f0 += 16)
        for (int r0 = 0; r0 < 32; r0 += 16) {
            f32x4 d;
            for (int r = 0; r < 4; ++r) {   // the same starting values as (a): feature f = f0 + 4 g + r sits in (a)'s register rr of half hh with f = (rr & 3) + 8 (rr >> 2) + 4 hh
                const int f = f0 + 4 * g + r, hh = (f >> 2) & 1, rr = (f & 3) + 4 * (f >> 3);
                (void)hh;
                d[r] = 0.25f * (float)((rr + (r0 + l16) + t) % 7) - 0.5f;
            }
            for (int step = 0; step < KK / 32; ++step) {
                f16x8 a, b;
                for (int i = 0; i < 8; ++i) { a[i] = A[(f0 + l16) * KK + 32 * step + 8 * g + i]; b[i] = B[(32 * step + 8 * g + i) * 32 + r0 + l16]; }
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
            }
            for (int r = 0; r < 4; ++r) out16[(f0 + 4 * g + r) * 32 + r0 + l16] = d[r];
        }
}
template <int SHAPE>
__global__ __launch_bounds__(512) void k_rate(float* sink, long long* clk, int n)
{
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    f32x16 c0 = {}, c1 = {}; c1[0] = 1.f;
    f32x4 d0 = {}, d1 = {}; d1[0] = 1.f;
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        if (SHAPE == 32) { c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0); }
        else { d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, d1, 0, 0, 0); }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 448 && blockIdx.x == 3) clk[0] = t1 - t0;   // (the younger wave of its SIMD: the SIMD's MFMAs are over when it is)
    sink[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + d0[0] + d1[1];
}
int main()
{
    const int TR = 400, NA = 32 * KK, N = 1024;
    static _Float16 hA[400 * 32 * KK], hB[400 * 32 * KK];
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f * 2.0f - 1.0f; };
    // operands as the policy produces them: hi halves up to 2^11, lo halves of a few units (block-scaled split), mixed signs; the scale varies
    // from trial to trial over eight binades
    for (int t = 0; t < TR; ++t) {
        const float sa = ldexpf(1.0f, (t % 9) - 4), sb = ldexpf(1.0f, ((t / 9) % 9) - 4);
        for (int i = 0; i < NA; ++i) { hA[t * NA + i] = (_Float16)(rnd() * (i % 3 ? 1500.0f : 0.7f) * sa * 0.02f); hB[t * NA + i] = (_Float16)(rnd() * (i % 5 ? 1900.0f : 0.9f) * sb * 0.02f); }
    }
    _Float16 *A, *B; float *o32, *o16, *sink; long long* clk;
    (void)hipMalloc(&A, sizeof(hA)); (void)hipMalloc(&B, sizeof(hB)); (void)hipMalloc(&o32, (size_t)TR * N * 4); (void)hipMalloc(&o16, (size_t)TR * N * 4);
    (void)hipMalloc(&sink, 256 * 512 * 4); (void)hipMalloc(&clk, 8);
    (void)hipMemcpy(A, hA, sizeof(hA), hipMemcpyHostToDevice); (void)hipMemcpy(B, hB, sizeof(hB), hipMemcpyHostToDevice);
    k_values<<<TR, 64>>>(A, B, o32, o16);
    static float r32[400 * 1024], r16[400 * 1024];
    (void)hipMemcpy(r32, o32, sizeof(r32), hipMemcpyDeviceToHost); (void)hipMemcpy(r16, o16, sizeof(r16), hipMemcpyDeviceToHost);
    long differ = 0; double worst_ulp = 0, worst_rel = 0;
    for (int t = 0; t < TR; ++t)
        for (int i = 0; i < N; ++i) {
            const int f = i / 32, row = i % 32;
            const int rr = (f & 3) + 4 * (f >> 3);
            double ref = 0.25 * (double)((rr + row + t) % 7) - 0.5;   // exact (double) dot product of the same f16 operands + the starting value
            for (int k = 0; k < KK; ++k) ref += (double)(float)hA[t * NA + f * KK + k] * (double)(float)hB[t * NA + k * 32 + row];
            const float x = r32[t * N + i], y = r16[t * N + i];
            uint32_t u, v; memcpy(&u, &x, 4); memcpy(&v, &y, 4);
            if (u != v) { ++differ; const double ulp = fabs((double)x - (double)y) / ldexp(1.0, ilogb((double)x) - 23); if (ulp > worst_ulp) worst_ulp = ulp; }
            double sumabs = 0; for (int k = 0; k < KK; ++k) sumabs += fabs((double)(float)hA[t * NA + f * KK + k] * (double)(float)hB[t * NA + k * 32 + row]);
            const double rel = fabs((double)y - ref) / (sumabs + 1e-30); if (rel > worst_rel) worst_rel = rel;
        }
    printf("# tools/ubench/mfma_shapes.hip: %d trials x 32 x 32 dot products over k = 0..%d from a non-zero accumulator: six chained v_mfma_f32_32x32x16_f16 against three chained v_mfma_f32_16x16x32_f16 per 16 x 16 block\n", TR, KK - 1);
    printf("results_differing_in_bits %ld of %ld   largest_difference_ulp %.1f   (16x16x32 against the exact sum: worst error / sum of |products| %.2e)\n", differ, (long)TR * N, worst_ulp, worst_rel);
    const int n = 20000;
    long long c32, c16;
    for (int rep = 0; rep < 2; ++rep) { k_rate<32><<<256, 512>>>(sink, clk, n); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(&c32, clk, 8, hipMemcpyDeviceToHost);
    for (int rep = 0; rep < 2; ++rep) { k_rate<16><<<256, 512>>>(sink, clk, n); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(&c16, clk, 8, hipMemcpyDeviceToHost);
    const double p32 = (double)c32 / (4.0 * n), p16 = (double)c16 / (4.0 * n);
    printf("pipe_counts_per_mfma 32x32x16 %.2f   16x16x32 %.2f\n", p32, p16);
    printf("a 16-row half tile, per 32 features x 32 k: two 32x32x16 (half their columns padding) %.1f counts, two 16x16x32 %.1f counts\n", 2 * p32, 2 * p16);
    return 0;
}
