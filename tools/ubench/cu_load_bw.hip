// What can ONE CU pull through its vector-memory path when all 256 CUs stream the SAME L2-resident weights the way the policy tiles do?
// (round 6: is k_run's policy half bound by its weight stream?  Per world and tick four 32-row tiles read 4 x 240 KB of packed fragments:
// 960 KB per CU in ~10.5 us = 91 GB/s per CU, 23 TB/s over the chip.)  256 workgroups x 512 threads, one per CU; every wave reads 1 KB
// fragments (16 B per lane, `global_load_dwordx4`) round-robin from a buffer of `kb` KB with `depth` loads in flight, and sums them so that
// nothing is dropped.  Reported: GB/s per CU and for the chip, bytes per shader clock per CU.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/cu_load_bw tools/ubench/cu_load_bw.hip && /tmp/cu_load_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const f32x4 __attribute__((address_space(1))) gf32x4;

template <int DEPTH, bool SC1>
__global__ __launch_bounds__(512) void k(const float* buf, int n_frag, int iters, float* sink, long long* clk)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    gf32x4* p = (gf32x4*)buf + lane;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int f = (wave * 37 + blockIdx.x * 11) % n_frag;
    const long long t0 = clock64(); const long long w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        f32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (SC1) { const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, 0x7fffffff, 0x00027000);
                       v[d] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (f * 64 + lane) * 16, 0, 16)); }
            else v[d] = p[f * 64];
            f += 1; if (f >= n_frag) f -= n_frag;
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += v[d];
    }
    const long long t1 = clock64(); const long long w1 = wall_clock64();
    sink[blockIdx.x * 512 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
    if (threadIdx.x == 0 && blockIdx.x == 5) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}
template <int DEPTH, bool SC1>
void run(const float* buf, int kb, float* sink, long long* clk)
{
    const int n_frag = kb, iters = 4000 / DEPTH;   // 1 KB fragments
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<DEPTH, SC1><<<256, 512>>>(buf, n_frag, iters, sink, clk); (void)hipDeviceSynchronize();
    hipEventRecord(e0); k<DEPTH, SC1><<<256, 512>>>(buf, n_frag, iters, sink, clk); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c[2]; (void)hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    const double bytes_cu = 8.0 * iters * DEPTH * 1024.0;
    printf("buffer %5d KB  depth %2d  %s : %7.1f GB/s per CU  %6.2f TB/s chip  %5.1f B per shader clock per CU  (kernel %.1f us; in-kernel clock %.2f GHz)\n", kb, DEPTH, SC1 ? "sc1 " : "plain",
           bytes_cu / (ms * 1e-3) / 1e9, 256 * bytes_cu / (ms * 1e-3) / 1e12, bytes_cu / (double)c[0], ms * 1e3, (double)c[0] / ((double)c[1] * 10.0));
}
int main()
{
    float *buf, *sink; long long* clk;
    (void)hipMalloc(&buf, 64 << 20); (void)hipMemset(buf, 0, 64 << 20); (void)hipMalloc(&sink, 256 * 512 * 4); (void)hipMalloc(&clk, 16);
    printf("# tools/ubench/cu_load_bw.hip: 256 workgroups x 8 waves, every wave streaming 1 KB fragments (16 B per lane) of one shared buffer\n");
    run<4, false>(buf, 16, sink, clk);      // fits the CU's 32 KB L1
    run<4, false>(buf, 480, sink, clk);     // two brains' fragments: L2
    run<8, false>(buf, 480, sink, clk);
    run<12, false>(buf, 480, sink, clk);
    run<12, false>(buf, 4096, sink, clk);   // 4 MB: one XCD's L2
    run<12, false>(buf, 65536, sink, clk);  // 64 MB: beyond L2 (MALL / HBM)
    return 0;
}
