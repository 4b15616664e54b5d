// anyorder.hip — do kernels of ONE stream overlap when launched with hipExtAnyOrderLaunch on this runtime / part?
// (hip_ext.h says the flag is "not supported on AMD GFX9xx boards"; this measures what actually happens.)  Each kernel is one
// workgroup that spins for ~100 us; 4 launches back to back take ~400 us if serialised, ~100 us if they overlap.  For
// comparison: the same 4 launches on 4 streams.  hipcc --offload-arch=gfx950 -O3 -o build/anyorder anyorder.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(64) void k_spin(unsigned long long ticks, unsigned long long* out)
{
    const unsigned long long t0 = wall_clock64();  // s_memrealtime: 100 MHz constant clock
    unsigned long long t = t0;
    int guard = 0;
    while (t - t0 < ticks && guard < (1 << 24)) { t = wall_clock64(); guard++; }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t0; out[2 * blockIdx.x + 1] = t; }
}

int main()
{
    unsigned long long* d;
    CK(hipMalloc(&d, 4096));
    hipStream_t st, s4[4];
    CK(hipStreamCreate(&st));
    for (int i = 0; i < 4; i++) CK(hipStreamCreate(&s4[i]));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const unsigned long long ticks = 10000;  // 100 us at 100 MHz
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 4; i++) {
                if (mode == 0) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, ticks, d + 8 * i);
                else if (mode == 1) hipExtLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, d + 8 * i);
                else hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s4[i], ticks, d + 8 * i);
            }
            if (mode == 2) for (int i = 0; i < 4; i++) CK(hipStreamSynchronize(s4[i]));
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h[32];
            CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
            printf("%s rep %d: events %.1f us; kernel starts (us after the first): %.1f %.1f %.1f\n",
                   mode == 0 ? "plain launches, one stream  " : mode == 1 ? "hipExtAnyOrderLaunch, one st" : "four streams                ", rep,
                   ms * 1e3, ((long long)h[8] - (long long)h[0]) / 100.0, ((long long)h[16] - (long long)h[0]) / 100.0, ((long long)h[24] - (long long)h[0]) / 100.0);
        }
    }
    return 0;
}
