// Microbenchmark: cost of one barrier-separated LDS phase for a 1024/512/256-thread workgroup per CU (tuning aid).
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int T, int FULL>
__global__ __launch_bounds__(T) void k(int n_phases, int work, int* out)
{
    __shared__ int a[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += T) a[i] = i;
    __syncthreads();
    int acc = 0;
    for (int p = 0; p < n_phases; ++p) {
        int idx = (tid * 7 + p) & 4095;
        for (int wk = 0; wk < work; ++wk) idx = a[idx] & 4095;  // dependent LDS chain of length `work`
        acc += idx;
        a[(tid + p) & 4095] = acc;
        if (FULL) __syncthreads(); else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (acc == -1) out[0] = acc;
}
template <int T, int FULL>
void run(const char* name, int work)
{
    int* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        const int NP = 200;
        hipLaunchKernelGGL((k<T, FULL>), dim3(256), dim3(T), 0, 0, 0, work, d);
        hipEventRecord(e0); hipLaunchKernelGGL((k<T, FULL>), dim3(256), dim3(T), 0, 0, 0, work, d); hipEventRecord(e1); hipEventSynchronize(e1);
        float t0; hipEventElapsedTime(&t0, e0, e1);
        hipEventRecord(e0); hipLaunchKernelGGL((k<T, FULL>), dim3(256), dim3(T), 0, 0, NP, work, d); hipEventRecord(e1); hipEventSynchronize(e1);
        float t1; hipEventElapsedTime(&t1, e0, e1);
        if (rep) printf("%-28s T=%4d work=%2d : %.3f us per phase (empty launch %.1f us)\n", name, T, work, (t1 - t0) * 1e3 / NP, t0 * 1e3);
    }
}
int main()
{
    for (int work : {0, 4, 16}) {
        run<1024, 1>("__syncthreads", work); run<1024, 0>("lgkmcnt+s_barrier", work);
        run<512, 0>("lgkmcnt+s_barrier", work); run<256, 0>("lgkmcnt+s_barrier", work);
    }
    return 0;
}
