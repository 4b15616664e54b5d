// store_stride.hip — does it matter that tw_pyr_k3f's level-0 stores are two 16-byte stores per lane at a 32-byte lane stride
// (each instruction writes every other 16-byte piece of a 2 KB span) instead of two instructions that each write 1 KB contiguously?
// 1 GiB of floats written per launch, rows of 1920 floats like the kernel's (one wave = 512 consecutive pixels of a row).
//   mode 0: lane i writes pieces 2i and 2i+1 (the kernel's pattern)      mode 1: lane i writes pieces i and 64+i
//   mode 2: mode 0 with nontemporal stores                                mode 3: mode 1 with nontemporal stores
// hipcc --offload-arch=gfx950 -O3 -o build/store_stride store_stride.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* __restrict__ d, size_t nspans)
{
    const size_t span = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // one wave = one 2 KB span (512 floats)
    if (span >= nspans) return;
    const int lane = threadIdx.x & 63;
    float* p = d + span * 512;
    const v4f a = {1.f, 2.f, 3.f, (float)lane}, b = {5.f, 6.f, 7.f, (float)lane};
    v4f* q0 = (v4f*)(p + ((MODE & 1) ? 4 * lane : 8 * lane));
    v4f* q1 = (v4f*)(p + ((MODE & 1) ? 256 + 4 * lane : 8 * lane + 4));
    if (MODE & 2) { __builtin_nontemporal_store(a, q0); __builtin_nontemporal_store(b, q1); }
    else { *q0 = a; *q1 = b; }
}
int main()
{
    const size_t bytes = 1ull << 30, nspans = bytes / 2048;
    float* d;
    CK(hipMalloc(&d, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; rep++)
        for (int mode = 0; mode < 4; mode++) {
            const dim3 grid((unsigned)((nspans + 3) / 4));
            for (int w = 0; w < 2; w++) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, grid, dim3(256), 0, 0, d, nspans);
                if (mode == 1) hipLaunchKernelGGL(k<1>, grid, dim3(256), 0, 0, d, nspans);
                if (mode == 2) hipLaunchKernelGGL(k<2>, grid, dim3(256), 0, 0, d, nspans);
                if (mode == 3) hipLaunchKernelGGL(k<3>, grid, dim3(256), 0, 0, d, nspans);
            }
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 10; i++) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, grid, dim3(256), 0, 0, d, nspans);
                if (mode == 1) hipLaunchKernelGGL(k<1>, grid, dim3(256), 0, 0, d, nspans);
                if (mode == 2) hipLaunchKernelGGL(k<2>, grid, dim3(256), 0, 0, d, nspans);
                if (mode == 3) hipLaunchKernelGGL(k<3>, grid, dim3(256), 0, 0, d, nspans);
            }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("mode %d: %.1f us per GiB  %.2f TB/s\n", mode, ms * 100.0, bytes / (ms * 1e-4) / 1e12);
        }
    return 0;
}
