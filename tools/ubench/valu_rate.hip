// Micro-benchmark: wave64 VALU issue rates on gfx950 (non-fused f32 add/mul, packed f32, f64, cvt).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float __attribute__((ext_vector_type(2))) f2;
#define REP 256
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int n)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
    double d0 = x0, d1 = x1, d2 = x2, d3 = x3, da = a;
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
            if (MODE == 0) { x0 += a; x1 += a; x2 += a; x3 += a; x4 += a; x5 += a; x6 += a; x7 += a; }
            if (MODE == 1) { x0 *= a; x1 *= a; x2 *= a; x3 *= a; x4 *= a; x5 *= a; x6 *= a; x7 *= a; }
            if (MODE == 2) { x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                             x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b); }
            if (MODE == 3) { p0 += pa; p1 += pa; p2 += pa; p3 += pa; }
            if (MODE == 4) { p0 *= pa; p1 *= pa; p2 *= pa; p3 *= pa; }
            if (MODE == 5) { d0 += da; d1 += da; d2 += da; d3 += da; }
            if (MODE == 6) { d0 = __builtin_fma(d0, da, da); d1 = __builtin_fma(d1, da, da); d2 = __builtin_fma(d2, da, da); d3 = __builtin_fma(d3, da, da); }
            if (MODE == 7) { d0 = (double)(float)d0; d1 = (double)(float)d1; d2 = (double)(float)d2; d3 = (double)(float)d3;
                             asm volatile("" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)); }
            if (MODE == 8) { p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb); p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb); }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1] + (float)(d0 + d1 + d2 + d3);
}
template <int MODE>
void run(const char* name, int ops_per_rep, int blocks_per_cu)
{
    float* out;
    const int nb = 256 * blocks_per_cu;
    hipMalloc(&out, nb * 256 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int n = 64;
    k<MODE><<<nb, 256>>>(out, 1.0001f, 0.5f, 2);
    hipEventRecord(a);
    k<MODE><<<nb, 256>>>(out, 1.0001f, 0.5f, n);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double winst = (double)nb * 4 * n * REP * ops_per_rep;  // wave-instructions
    double per_simd_cyc = ms * 1e-3 * 2.4e9 / (winst / 1024.0);
    printf("%-22s blocks/CU %d: %.3f ms  %.2f cycles/wave-instr/SIMD (at 2.4 GHz)  %.1f T lane-ops/s\n", name, blocks_per_cu, ms,
           per_simd_cyc, winst * 64 / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main()
{
    for (int bpc : {1, 2, 4, 8}) {
        run<0>("v_add_f32", 8, bpc);
        run<1>("v_mul_f32", 8, bpc);
        run<2>("v_fma_f32", 8, bpc);
        run<3>("v_pk_add_f32", 4, bpc);
        run<4>("v_pk_mul_f32", 4, bpc);
        run<8>("v_pk_fma_f32", 4, bpc);
        run<5>("v_add_f64", 4, bpc);
        run<6>("v_fma_f64", 4, bpc);
        run<7>("cvt f64<->f32 (x2)", 8, bpc);
    }
    return 0;
}
