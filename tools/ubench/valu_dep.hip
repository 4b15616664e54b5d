// Micro-benchmark: dependent-chain issue interval of wave64 f32 VALU ops on gfx950 vs waves per SIMD and ILP.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 512
template <int ILP>
__global__ __launch_bounds__(256) void k(float* out, float a, int n)
{
    float x[8];
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x + i;
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
#pragma unroll
            for (int i = 0; i < ILP; i++) x[i] += a;
        }
    }
    float s = 0;
    for (int i = 0; i < ILP; i++) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// the blur inner pattern: s += (p + q) * k, two rows interleaved
__global__ __launch_bounds__(256) void kpat(float* out, const float* in, float k0, int n)
{
    float w[40];
    for (int i = 0; i < 40; i++) w[i] = in[threadIdx.x + i * 256];
    float s0 = 0, s1 = 0;
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int i = 1; i <= 15; i++) {
                s0 += (w[16 + i] + w[16 - i]) * k0;
                s1 += (w[17 + i] + w[17 - i]) * k0;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = s0 + s1;
}
template <int ILP>
void run(int bpc)
{
    float* out;
    const int nb = 256 * bpc;
    (void)hipMalloc(&out, nb * 256 * 4);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const int n = 32;
    k<ILP><<<nb, 256>>>(out, 1.0001f, 2);
    (void)hipEventRecord(a);
    k<ILP><<<nb, 256>>>(out, 1.0001f, n);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    double winst = (double)nb * 4 * n * REP * ILP;
    printf("ILP %d waves/SIMD %d: %.2f cycles per wave-instr per SIMD; per-wave issue interval %.1f cycles\n", ILP, bpc,
           ms * 1e-3 * 2.4e9 / (winst / 1024.0), ms * 1e-3 * 2.4e9 / ((double)n * REP * ILP));
    (void)hipFree(out);
}
int main()
{
    for (int bpc : {1, 2, 4, 8}) {
        run<1>(bpc);
        run<2>(bpc);
        run<4>(bpc);
        run<8>(bpc);
    }
    float *out, *in;
    (void)hipMalloc(&out, 2048 * 256 * 4);
    (void)hipMalloc(&in, 256 * 64 * 4);
    (void)hipMemset(in, 0, 256 * 64 * 4);
    for (int bpc : {1, 2, 4, 8}) {
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        const int nb = 256 * bpc, n = 64;
        kpat<<<nb, 256>>>(out, in, 0.3f, 2);
        (void)hipEventRecord(a);
        kpat<<<nb, 256>>>(out, in, 0.3f, n);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        double winst = (double)nb * 4 * n * 16 * 15 * 6;
        printf("blur pattern waves/SIMD %d: %.2f cycles per wave-instr per SIMD\n", bpc, ms * 1e-3 * 2.4e9 / (winst / 1024.0));
    }
    return 0;
}
