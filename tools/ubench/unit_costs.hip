// Unit costs for the latency-bound model of one k_run tick (bench.py roofline.latency_bound_us): what ONE world on ONE CU pays
// per barrier, per dependent LDS / L2 round trip, per dependent VALU instruction of a wave that works alone, per MFMA -- in shader-clock
// counts (clock64 = s_memtime) AND in ns (wall_clock64: the constant 100 MHz counter), measured on a busy chip (256 workgroups of 512
// threads = k_run's launch shape, 100 KB of LDS each: one workgroup per CU).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/unit_costs tools/ubench/unit_costs.hip && /tmp/unit_costs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Out { long long clk, wall; };
#define BEGIN const long long w0 = wall_clock64(); const long long t0 = clock64();
#define END(slot) do { const long long t1 = clock64(); const long long w1 = wall_clock64(); if (threadIdx.x == 0 && blockIdx.x == 7) { o[slot].clk = t1 - t0; o[slot].wall = w1 - w0; } } while (0)

// 0: the barrier alone -- n x (s_waitcnt lgkmcnt(0); s_barrier), 8 waves in lockstep (no skew: the floor of an interval)
__global__ __launch_bounds__(512) void k_barrier(Out* o, int n, int* sink)
{
    extern __shared__ int lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    BEGIN
    for (int i = 0; i < n; ++i) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    END(0);
    if (lds[threadIdx.x] == -1) sink[0] = 1;
}
// 1: barrier + one LDS write and one LDS read of another wave's word per interval (what an exchange costs at least)
__global__ __launch_bounds__(512) void k_barrier_xchg(Out* o, int n, int* sink)
{
    extern __shared__ int lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    int v = threadIdx.x;
    BEGIN
    for (int i = 0; i < n; ++i) {
        lds[threadIdx.x] = v;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        v = lds[(threadIdx.x + 64) & 511] + 1;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    END(1);
    if (v == -1) sink[0] = 1;
}
// 2: dependent LDS reads of ONE wave (the other seven parked at a barrier): latency of a ds_read_b32 round trip
__global__ __launch_bounds__(512) void k_lds_chain(Out* o, int n, int* sink)
{
    extern __shared__ int lds[];
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = (i * 67 + 13) & 4095;
    __syncthreads();
    int idx = threadIdx.x & 63;
    if (threadIdx.x < 64) {
        BEGIN
        for (int i = 0; i < n; ++i) idx = lds[idx];
        END(2);
    }
    __syncthreads();
    if (idx == -1) sink[0] = 1;
}
// 3: dependent global loads that hit L2 (sc1: not the CU's L1) by ONE wave: an L2 round trip
__global__ __launch_bounds__(512) void k_l2_chain(Out* o, int n, const int* tbl, int* sink)
{
    int idx = (threadIdx.x & 63) + 64 * (blockIdx.x & 15);
    if (threadIdx.x < 64) {
        BEGIN
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)tbl, 0, 0x7fffffff, 0x00027000);
        for (int i = 0; i < n; ++i) idx = __builtin_amdgcn_raw_buffer_load_b32(rsrc, idx * 4, 0, 16 /* sc1: past the CU's L1, as k_run's row and weight reads */);
        END(3);
    }
    __syncthreads();
    if (idx == -1) sink[0] = 1;
}
// 4 / 5: VALU by ONE wave alone on its SIMD (seven waves parked): a DEPENDENT chain (4), and eight independent chains (5)
__global__ __launch_bounds__(512) void k_valu(Out* o, int n, float* fs, int dep)
{
    float a0 = threadIdx.x * 0.001f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float sc = 1.0009765625f;
    if (threadIdx.x < 64) {
        BEGIN
        if (dep)
            for (int i = 0; i < n; ++i)
                asm volatile("v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0" : "+v"(a0) : "v"(sc));
        else
            for (int i = 0; i < n; ++i)
                asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sc));
        END(dep ? 4 : 5);
    }
    __syncthreads();
    fs[blockIdx.x * 512 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
// 6: v_mfma_f32_32x32x16_f16, every wave of the workgroup TWO accumulator chains (two waves per SIMD, as the tile pair): pipe cycles per MFMA
__global__ __launch_bounds__(512) void k_mfma(Out* o, int n, float* fs)
{
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    f32x16 c0 = {}, c1 = {};
    c1[0] = 1.0f;   // (two DIFFERENT chains: identical ones are one chain after common-subexpression elimination)
    BEGIN
    for (int i = 0; i < n; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
    }
    END(6);
    {   // the LAST wave's end against the same start: of the two waves that share a SIMD the older one wins the arbitration and runs at the pipe's full
        // rate by itself (its two chains fill it), the younger one follows -- the SIMD's MFMAs are over when the younger wave is
        const long long t1 = clock64(); const long long w1 = wall_clock64();
        if (threadIdx.x == 448 && blockIdx.x == 7) { o[8].clk = t1 - t0; o[8].wall = w1 - w0; }
    }
    fs[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[3];
}
// 7: a wave's clock rate under the MFMA load above is the rate of slot 6; idle-ish: n x s_sleep(1) -> counts per ns of the two clocks
__global__ __launch_bounds__(512) void k_clock(Out* o, int n)
{
    BEGIN
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(8);
    END(7);
}

int main()
{
    Out* o; int* sink; float* fs; int* tbl;
    (void)hipMalloc(&o, sizeof(Out) * 16); (void)hipMalloc(&sink, 64); (void)hipMalloc(&fs, sizeof(float) * 256 * 512);
    const int TN = 1024;
    int h[TN]; for (int i = 0; i < TN; ++i) h[i] = (i * 389 + 17) % TN;
    (void)hipMalloc(&tbl, sizeof(h)); (void)hipMemcpy(tbl, h, sizeof(h), hipMemcpyHostToDevice);
    const size_t L = 100 * 1024;
    (void)hipFuncSetAttribute((const void*)k_barrier, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L);
    (void)hipFuncSetAttribute((const void*)k_barrier_xchg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L);
    (void)hipFuncSetAttribute((const void*)k_lds_chain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L);
    const int N = 4000;
    for (int rep = 0; rep < 3; ++rep) {   // (the last repetition is reported: clocks settled)
        k_mfma<<<256, 512>>>(o, 20000, fs);
        k_barrier<<<256, 512, L>>>(o, N, sink);
        k_barrier_xchg<<<256, 512, L>>>(o, N, sink);
        k_lds_chain<<<256, 512, L>>>(o, N, sink);
        k_l2_chain<<<256, 512>>>(o, N, tbl, sink);
        k_valu<<<256, 512>>>(o, N, fs, 1);
        k_valu<<<256, 512>>>(o, N, fs, 0);
        k_clock<<<256, 512>>>(o, N);
        (void)hipDeviceSynchronize();
    }
    Out r[16]; (void)hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    auto rate = [&](int s) { return (double)r[s].clk / ((double)r[s].wall * 10.0); };   // clock64 counts per ns (wall_clock64: 100 MHz = 10 ns per count)
    printf("# tools/ubench/unit_costs.hip: 256 workgroups x 512 threads, one per CU; clock64 counts (and ns by wall_clock64, 100 MHz)\n");
    printf("clock64_counts_per_ns_under_mfma_load %.4f\n", rate(6));
    printf("clock64_counts_per_ns_barrier_loop %.4f\n", rate(0));
    printf("clock64_counts_per_ns_sleeping %.4f\n", rate(7));
    printf("barrier_counts %.1f  ns %.2f   (s_waitcnt lgkmcnt(0); s_barrier, eight waves in lockstep)\n", (double)r[0].clk / N, r[0].wall * 10.0 / N);
    printf("barrier_exchange_counts %.1f  ns %.2f   (per barrier of: LDS write, barrier, LDS read of another wave's word, barrier)\n", (double)r[1].clk / (2.0 * N), r[1].wall * 10.0 / (2.0 * N));
    printf("lds_round_trip_counts %.1f  ns %.2f   (dependent ds_read_b32, one wave)\n", (double)r[2].clk / N, r[2].wall * 10.0 / N);
    printf("l2_round_trip_counts %.1f  ns %.2f   (dependent buffer_load_dword sc1, 4 KB table: L2 hit)\n", (double)r[3].clk / N, r[3].wall * 10.0 / N);
    printf("valu_dependent_counts %.2f  ns %.3f   (dependent v_fma_f32, one wave alone on its SIMD)\n", (double)r[4].clk / (8.0 * N), r[4].wall * 10.0 / (8.0 * N));
    printf("valu_independent_counts %.2f  ns %.3f   (eight independent v_fma_f32 chains, one wave alone on its SIMD)\n", (double)r[5].clk / (8.0 * N), r[5].wall * 10.0 / (8.0 * N));
    printf("mfma_pipe_counts %.2f  ns %.3f   (v_mfma_f32_32x32x16_f16: the SIMD's 2 waves x 2 chains x 20000 MFMAs over the younger wave's span)\n", (double)r[8].clk / (4.0 * 20000), r[8].wall * 10.0 / (4.0 * 20000));
    printf("mfma_older_wave_counts %.2f   (the older wave of the SIMD alone: its own 2 x 20000 MFMAs over its own span)\n", (double)r[6].clk / (2.0 * 20000));
    return 0;
}
