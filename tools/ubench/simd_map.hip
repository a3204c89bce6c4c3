// Which SIMD does wave w of a 512-thread workgroup run on?  (HW_REG_HW_ID: simd_id = bits 5:4 on gfx9-class hardware)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void k(unsigned* out)
{
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main()
{
    unsigned* d; (void)hipMalloc(&d, 4 * 8 * 64);
    k<<<64, 512, 100 * 1024>>>(d); (void)hipDeviceSynchronize();
    unsigned h[8 * 64]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 4; ++b) { printf("block %d: simd of waves 0..7:", b); for (int w = 0; w < 8; ++w) printf(" %u", (h[b * 8 + w] >> 4) & 3); printf("   cu %u\n", (h[b * 8] >> 8) & 15); }
    return 0;
}
