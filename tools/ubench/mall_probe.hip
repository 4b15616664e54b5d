// Micro-benchmark (round 4): does the 256 MB Infinity Cache (MALL) hold ONE pair's level-0 working set between launches?
// The refreshing blur launch's memory shape (tools/ubench/blur_shape.hip mode 0: 38-row M windows + R0 + R1 2x2 taps in,
// M out; 166 MB per 1920x1080 pair) is launched over `np` pairs, 20 launches back to back, either on the SAME np pairs every
// time (hot: what a pair-major schedule would see — the next kernel of a pair's chain finds M / R0 / R1 where the previous
// one left them) or on a DIFFERENT set of np pairs every launch (cold: the level-major batch schedule, 21 GB between two
// touches of a byte).  Same grid, same tails: the difference is the cache.   hipcc --offload-arch=gfx950 -O3 -w
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang diagnostic ignored "-Wunused-value"
constexpr int W = 1920, H = 1080, LD = 1920, TW = 224, TH = 8, MH = 15, HALO = 16, NW = TH + 2 * MH;
__device__ __forceinline__ void xcd_remap(int& bx, int& by, int& bz)
{
    const unsigned gx = gridDim.x, gy = gridDim.y, n = gx * gy * gridDim.z;
    unsigned b = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned xcd = b & 7u, q = n >> 3, r = n & 7u;
    b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    bx = (int)(b % gx);
    const unsigned t = b / gx;
    by = (int)(t % gy);
    bz = (int)(t / gy);
}
// MODE 0: windows + refresh (80 B/px); 2: refresh only (60 B/px); 3: M windows in, flow out (28 B/px)
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ Min, const float* __restrict__ R, float* __restrict__ Mout,
                                         float* __restrict__ flow, long long ps, int zbase)
{
    int bx, by, z;
    xcd_remap(bx, by, z);
    z += zbase;
    const int x0 = bx * TW - 16, y0 = by * TH, tid = threadIdx.x;
    float acc = 0.f;
    if (MODE != 2) {
        const float* M = Min + (long long)z * 5 * ps;
        const int x = min(max(x0 - HALO + tid, 0), W - 1);
        if (x0 - HALO + tid >= -MH && x0 - HALO + tid <= W - 1 + MH) {
#pragma unroll
            for (int ch = 0; ch < 5; ch++) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(M + ch * ps), 0, 0xFFFFFFFFu, 0x00020000);
#pragma unroll
                for (int i = 0; i < NW; i++) {
                    const unsigned ro = (unsigned)min(max(y0 - MH + i, 0), H - 1) * (LD * 4u);
                    acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)x * 4u, ro, 0));
                }
            }
        }
    }
    const float* R0 = R + (long long)(2 * z) * 5 * ps;
    const float* R1 = R0 + 5 * ps;
    float* Mo = Mout + (long long)z * 5 * ps;
    float* fl = flow + (long long)z * 2 * ps;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const int p = tid + i * 256, r = p / TW, cx = p - r * TW;
        const int x = x0 + cx, y = y0 + r;
        const bool valid = x >= 0 && x < W && y < H;
        const int xc = min(max(x, 0), W - 2), yc = min(y, H - 2);
        const long long o = (long long)yc * LD + xc;
        if (MODE == 3) {
            if (valid) {
                fl[o] = acc;
                fl[o + ps] = acc + 1.f;
            }
            continue;
        }
        float s = acc;
#pragma unroll
        for (int c = 0; c < 5; c++) {
            s += R0[o + c * ps];
            s += R1[o + c * ps] + R1[o + c * ps + 1] + R1[o + c * ps + LD] + R1[o + c * ps + LD + 1];
        }
        if (valid) {
#pragma unroll
            for (int c = 0; c < 5; c++) Mo[o + c * ps] = s + (float)c;
        }
    }
}
int main()
{
    const int NP = 64;
    const long long ps = (long long)LD * H;
    float *M0, *M1, *R, *fl;
    hipMalloc(&M0, ps * 5 * NP * 4);
    hipMalloc(&M1, ps * 5 * NP * 4);
    hipMalloc(&R, ps * 10 * NP * 4 + 4096);
    hipMalloc(&fl, ps * 2 * NP * 4);
    hipMemset(M0, 0, ps * 5 * NP * 4);
    hipMemset(M1, 0, ps * 5 * NP * 4);
    hipMemset(R, 0, ps * 10 * NP * 4 + 4096);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int nps[] = {1, 2, 3, 4, 8, 16};
    printf("%-44s %4s %10s %10s %8s\n", "shape (LDS: workgroups per CU)", "np", "hot us/pr", "cold us/pr", "hot/cold");
    for (int lds : {0, 40960}) {
        for (int mode : {0, 2, 3}) {
            for (int np : nps) {
                const dim3 grid((W + 16 + TW - 1) / TW, (H + TH - 1) / TH, np);
                double us[2];
                for (int cold = 0; cold < 2; cold++) {
                    const int iters = 20;
                    auto launch = [&](int i) {
                        // hot: the same np pairs, M0 -> M1 then M1 -> M0 (a chain: what one launch wrote the next one reads)
                        const int zb = cold ? (i * np) % (NP - np + 1) : 0;
                        const float* mi = (i & 1) ? M1 : M0;
                        float* mo = (i & 1) ? M0 : M1;
                        if (mode == 0) hipLaunchKernelGGL(k<0>, grid, dim3(256), lds, 0, mi, R, mo, fl, ps, zb);
                        if (mode == 2) hipLaunchKernelGGL(k<2>, grid, dim3(256), lds, 0, mi, R, mo, fl, ps, zb);
                        if (mode == 3) hipLaunchKernelGGL(k<3>, grid, dim3(256), lds, 0, mi, R, mo, fl, ps, zb);
                    };
                    for (int i = 0; i < 4; i++) launch(i);
                    double best = 1e30;
                    for (int rep = 0; rep < 2; rep++) {
                        hipEventRecord(a);
                        for (int i = 0; i < iters; i++) launch(i);
                        hipEventRecord(b);
                        hipEventSynchronize(b);
                        float ms = 0;
                        hipEventElapsedTime(&ms, a, b);
                        best = ms * 1e3 / iters / np < best ? ms * 1e3 / iters / np : best;
                    }
                    us[cold] = best;
                }
                const char* names[4] = {"windows + refresh (80 B/px)", "", "refresh only (60 B/px)", "windows + flow out (28 B/px)"};
                char nm[96];
                snprintf(nm, sizeof nm, "%s, %s", names[mode], lds ? "4 wg/CU" : "8 wg/CU");
                printf("%-44s %4d %10.2f %10.2f %8.3f\n", nm, np, us[0], us[1], us[0] / us[1]);
            }
        }
    }
    return 0;
}
