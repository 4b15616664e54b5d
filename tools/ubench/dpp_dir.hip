// Which way do the gfx9 wave-shift DPP controls move data?  Prints what lanes 0, 1, 15, 16, 62, 63 receive.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* o)
{
    const int lane = threadIdx.x;
    o[lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x130, 0xf, 0xf, false);        // wave_shl:1
    o[64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x138, 0xf, 0xf, false);   // wave_shr:1
}
int main()
{
    int* d;
    int h[128];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int m = 0; m < 2; m++) {
        printf("%s:", m ? "wave_shr:1" : "wave_shl:1");
        for (int l : {0, 1, 15, 16, 31, 32, 62, 63}) printf(" lane%d<-%d", l, h[64 * m + l]);
        printf("\n");
    }
    return 0;
}
