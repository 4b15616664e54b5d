// copy_rate.hip — which float4 device copy is the fair yardstick for bench.py's `frac_of_measured_copy`?
// (MI355X_MICROARCH.md quotes 6.29 TB/s for "float4 copy"; round 4's torch copy_ reached 5.5, round 5's first
// grid-stride kernel 5.1.)  Variants: grid-stride with 8 / 16 / 32 workgroups per CU, 4 loads in flight per thread,
// one float4 per thread (no loop), nontemporal stores.  hipcc --offload-arch=gfx950 -O3 -o build/copy_rate copy_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_stride(const float4* __restrict__ s, float4* __restrict__ d, size_t n)
{
    const size_t st = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += st) d[i] = s[i];
}
__global__ __launch_bounds__(256) void k_stride4(const float4* __restrict__ s, float4* __restrict__ d, size_t n)
{
    const size_t st = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * st < n; i += 4 * st) {
        const float4 a = s[i], b = s[i + st], c = s[i + 2 * st], e = s[i + 3 * st];
        d[i] = a; d[i + st] = b; d[i + 2 * st] = c; d[i + 3 * st] = e;
    }
    for (; i < n; i += st) d[i] = s[i];
}
__global__ __launch_bounds__(256) void k_one(const float4* __restrict__ s, float4* __restrict__ d, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = s[i];
}
__global__ __launch_bounds__(256) void k_four(const float4* __restrict__ s, float4* __restrict__ d, size_t n)
{
    // a workgroup copies 4 consecutive 4 KiB chunks: 4 loads in flight per thread, no loop
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    if (i + 768 < n) {
        const float4 a = s[i], b = s[i + 256], c = s[i + 512], e = s[i + 768];
        d[i] = a; d[i + 256] = b; d[i + 512] = c; d[i + 768] = e;
    }
}
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_four_nt(const v4f* __restrict__ s, v4f* __restrict__ d, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    if (i + 768 < n) {
        const v4f a = __builtin_nontemporal_load(s + i), b = __builtin_nontemporal_load(s + i + 256),
                  c = __builtin_nontemporal_load(s + i + 512), e = __builtin_nontemporal_load(s + i + 768);
        __builtin_nontemporal_store(a, d + i); __builtin_nontemporal_store(b, d + i + 256);
        __builtin_nontemporal_store(c, d + i + 512); __builtin_nontemporal_store(e, d + i + 768);
    }
}
int main()
{
    const size_t bytes = (size_t)1 << 30, n = bytes / 16;
    float4 *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cu = pr.multiProcessorCount;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        launch(); launch();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; i++) launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.1f GB/s (read + write)\n", name, 2.0 * bytes * 20 / (ms * 1e-3) / 1e9);
        return 0;
    };
    for (int wg : {8, 16, 32, 64}) {
        char nm[64]; snprintf(nm, sizeof nm, "grid-stride, %d workgroups per CU", wg);
        run(nm, [&] { hipLaunchKernelGGL(k_stride, dim3(cu * wg), dim3(256), 0, 0, a, b, n); });
        snprintf(nm, sizeof nm, "grid-stride x4 in flight, %d workgroups per CU", wg);
        run(nm, [&] { hipLaunchKernelGGL(k_stride4, dim3(cu * wg), dim3(256), 0, 0, a, b, n); });
    }
    run("one float4 per thread", [&] { hipLaunchKernelGGL(k_one, dim3((unsigned)(n / 256)), dim3(256), 0, 0, a, b, n); });
    run("four float4 per thread", [&] { hipLaunchKernelGGL(k_four, dim3((unsigned)(n / 1024)), dim3(256), 0, 0, a, b, n); });
    run("four float4 per thread, nontemporal", [&] { hipLaunchKernelGGL(k_four_nt, dim3((unsigned)(n / 1024)), dim3(256), 0, 0, (const v4f*)a, (v4f*)b, n); });
    run("hipMemcpyDtoD", [&] { (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
    CK(hipDeviceSynchronize());
    return 0;
}
