// Micro-benchmark: the MEMORY SHAPE of the fused window-average + solve + refresh launch (tw_blur_solve4<15,256,16,8>)
// with no arithmetic — what do its loads and stores alone cost on this MI355X?  Per 224x8 tile and 256 threads:
//   V  each lane loads the 38-row window of its column in each of the 5 M planes (190 loads, rows shared with the
//      tiles above / below: L2 hits), raw buffer loads like the real kernel
//   S  each lane, for 7 pixels: R0 (5 loads), the 2x2 R1 taps at zero flow (20 loads), 5 M stores
// Modes: 0 = V + S (the refreshing launch: 80 B/px as built), 1 = V only + the 8 B/px flow store (the last launch:
// 28 B/px), 2 = S only (the refresh's traffic alone: 60 B/px).  64 pairs of 1920x1080.
// hipcc --offload-arch=gfx950 -O3 -w
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
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ Min, const float* __restrict__ R, float* __restrict__ Mout,
                                         float* __restrict__ flow, long long ps)
{
    int bx, by, z;
    xcd_remap(bx, by, z);
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
        if (MODE == 1) {
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
// the refresh's traffic alone in another lane shape: every lane owns TWO horizontally adjacent pixels (8-byte loads of
// R0 and of the R1 rows, 8-byte M stores): half the vector-memory instructions for the same bytes
__global__ __launch_bounds__(256) void krefresh2(const float* __restrict__ R, float* __restrict__ Mout, long long ps)
{
    typedef float __attribute__((ext_vector_type(2))) f2;
    int bx, by, z;
    xcd_remap(bx, by, z);
    const int x0 = bx * TW - 16, y0 = by * TH, tid = threadIdx.x;
    const float* R0 = R + (long long)(2 * z) * 5 * ps;
    const float* R1 = R0 + 5 * ps;
    float* Mo = Mout + (long long)z * 5 * ps;
#pragma unroll
    for (int i = 0; i < 4; i++) {  // 112 pixel pairs x 8 rows = 896 items over 256 lanes
        const int p = tid + i * 256;
        if (p >= 112 * TH) break;
        const int r = p / 112, cp = p - r * 112;
        const int x = x0 + 2 * cp, y = y0 + r;
        const bool valid = x >= 0 && x + 1 < W && y < H;
        const int xc = min(max(x, 0), W - 4) & ~1, yc = min(y, H - 2);
        const long long o = (long long)yc * LD + xc;
        f2 s = {0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 5; c++) {
            s += *(const f2*)(R0 + o + c * ps);
            const f2 a = *(const f2*)(R1 + o + c * ps), b = *(const f2*)(R1 + o + c * ps + 2);
            const f2 d = *(const f2*)(R1 + o + c * ps + LD), e = *(const f2*)(R1 + o + c * ps + LD + 2);
            s += a + b + d + e;
        }
        if (valid) {
#pragma unroll
            for (int c = 0; c < 5; c++) *(f2*)(Mo + o + c * ps) = s + (float)c;
        }
    }
}
// ROLLING WINDOW (round 4): a workgroup marches NT tiles down its column strip and keeps the window, so
// every M row enters the CU once: per step only the TH new rows of each plane are loaded (NT*TH + 2*MH rows per NT*TH
// output rows instead of 38 per 8).  The S phase is the refreshing launch's, unchanged.  No LDS, no arithmetic, the
// occupancy of the real kernel (256 threads, 4 workgroups per CU): the most a rolling structure could gain.
//   DMA = true: the new rows go to LDS by LDS-DMA (buffer_load ... lds, no VGPR round trip) and are read back from there.
template <int NT, bool DMA, bool REFRESH>
__global__ __launch_bounds__(256) void kroll(const float* __restrict__ Min, const float* __restrict__ R, float* __restrict__ Mout,
                                             float* __restrict__ flow, long long ps)
{
    __shared__ __attribute__((aligned(16))) float ring[DMA ? 5 * TH * 256 : 1];
    int bx, seg, z;
    xcd_remap(bx, seg, z);
    const int x0 = bx * TW - 16, tid = threadIdx.x;
    const float* M = Min + (long long)z * 5 * ps;
    const int xw = min(max(x0 - HALO + tid, 0), W - 1);
    const bool von = x0 - HALO + tid >= -MH && x0 - HALO + tid <= W - 1 + MH;
    const float* R0 = R + (long long)(2 * z) * 5 * ps;
    const float* R1 = R0 + 5 * ps;
    float* Mo = Mout + (long long)z * 5 * ps;
    float* fl = flow + (long long)z * 2 * ps;
    float acc = 0.f;
    const int ys = seg * NT * TH;
    // prologue: the 2*MH rows above the first new block
    if (von) {
#pragma unroll
        for (int ch = 0; ch < 5; ch++) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(M + ch * ps), 0, 0xFFFFFFFFu, 0x00020000);
#pragma unroll
            for (int i = 0; i < 2 * MH; i++) {
                const unsigned ro = (unsigned)min(max(ys - MH + i, 0), H - 1) * (LD * 4u);
                acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)xw * 4u, ro, 0));
            }
        }
    }
#pragma unroll 1
    for (int st = 0; st < NT; st++) {
        const int y0 = ys + st * TH;
        if (y0 >= H) break;
        if (DMA) {
            // one wave-instruction = one 256-byte row segment -> 256 consecutive LDS bytes (wave-uniform base + lane * 4)
#pragma unroll
            for (int ch = 0; ch < 5; ch++) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(M + ch * ps), 0, 0xFFFFFFFFu, 0x00020000);
#pragma unroll
                for (int i = 0; i < TH; i++) {
                    const unsigned ro = (unsigned)min(max(y0 + MH + i, 0), H - 1) * (LD * 4u);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)&ring[(ch * TH + i) * 256 + (tid & ~63)], 4,
                                                         (unsigned)xw * 4u, ro, 0, 0);
                }
            }
            __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): the wave's own pieces have landed
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 5 * TH; j++) acc += ring[j * 256 + tid];
            __syncthreads();
        } else if (von) {
#pragma unroll
            for (int ch = 0; ch < 5; ch++) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(M + ch * ps), 0, 0xFFFFFFFFu, 0x00020000);
#pragma unroll
                for (int i = 0; i < TH; i++) {
                    const unsigned ro = (unsigned)min(max(y0 + MH + i, 0), H - 1) * (LD * 4u);
                    acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)xw * 4u, ro, 0));
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 7; i++) {
            const int p = tid + i * 256, r = p / TW, cx = p - r * TW;
            const int x = x0 + cx, y = y0 + r;
            const bool valid = x >= 0 && x < W && y < H;
            const int xc = min(max(x, 0), W - 2), yc = min(y, H - 2);
            const long long o = (long long)yc * LD + xc;
            if (!REFRESH) {
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
}
template <int NT, bool DMA, bool REFRESH>
static void run_roll(const char* name, const float* M0, const float* R, float* M1, float* fl, long long ps, int np, int lds = 0)
{
    const int nseg = ((H + TH - 1) / TH + NT - 1) / NT;
    const dim3 grid((W + 16 + TW - 1) / TW, nseg, np);
    double best = 1e30;
    for (int rep = 0; rep < 2; rep++) {
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipLaunchKernelGGL((kroll<NT, DMA, REFRESH>), grid, dim3(256), lds, 0, M0, R, M1, fl, ps);
        hipEventRecord(a);
        for (int i = 0; i < 10; i++) hipLaunchKernelGGL((kroll<NT, DMA, REFRESH>), grid, dim3(256), lds, 0, M0, R, M1, fl, ps);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        best = ms * 1e3 / 10 < best ? ms * 1e3 / 10 : best;
    }
    const double bpp = REFRESH ? 80 : 28;
    printf("rolling window, %-42s %8.1f us per %d pairs  %6.2f us/pair  %5.2f TB/s  (%d workgroups)\n", name, best, np, best / np,
           bpp * W * H * np / best / 1e6, (int)(grid.x * grid.y * grid.z));
}
// NEVER MATERIALISE M (round 4, the structural idea left): an iteration that reads the previous iteration's FLOW (8 B/px),
// R0 and R1 (40 B/px) and recomputes M on the fly for the rows entering a rolling window, then writes the new flow (8 B/px):
// 56 B/px instead of 80, no separate first-update launch, every launch the same.  Its memory shape, no arithmetic: a
// workgroup of SC columns (SC - 30 of them outputs) marches NT steps of 8 rows; per step every thread handles one pixel of
// the 8 new rows (R0, flow, the 2x2 R1 taps), then the step's 8 x (SC - 30) outputs store two flow planes.
template <int SC, int NT>
__global__ __launch_bounds__(SC * 8) void kfused(const float* __restrict__ R, const float* __restrict__ flin, float* __restrict__ flout,
                                                 long long ps)
{
    constexpr int OUT = SC - 2 * MH;
    const unsigned gx = gridDim.x, gy = gridDim.y, nb = gx * gy * gridDim.z;
    unsigned b = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned xcd = b & 7u, qq = nb >> 3, rr = nb & 7u;
    b = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (b >> 3);
    const int bx = (int)(b % gx), seg = (int)((b / gx) % gy), z = (int)(b / gx / gy);
    const int tid = threadIdx.x, col = tid % SC, rw = tid / SC;  // 8 rows x SC columns per step
    const int x = min(max(bx * OUT - MH + col, 0), W - 2);
    const float* R0 = R + (long long)(2 * z) * 5 * ps;
    const float* R1 = R0 + 5 * ps;
    const float* fi = flin + (long long)z * 2 * ps;
    float* fo = flout + (long long)z * 2 * ps;
    const int ys = seg * NT * TH;
    float acc = 0.f;
    auto row_px = [&](int y) {
        const int yc = min(max(y, 0), H - 2);
        const long long o = (long long)yc * LD + x;
        float s = fi[o] + fi[o + ps];
#pragma unroll
        for (int c = 0; c < 5; c++) {
            s += R0[o + c * ps];
            s += R1[o + c * ps] + R1[o + c * ps + 1] + R1[o + c * ps + LD] + R1[o + c * ps + LD + 1];
        }
        return s;
    };
    // prologue: the 30 rows above the first new block (4 rounds of 8 rows)
    for (int i = 0; i < 2 * MH; i += 8)
        if (i + rw < 2 * MH) acc += row_px(ys - MH + i + rw);
#pragma unroll 1
    for (int st = 0; st < NT; st++) {
        const int y0 = ys + st * TH;
        if (y0 >= H) break;
        acc += row_px(y0 + MH + rw);
        const int ox = bx * OUT + col - MH, oy = y0 + rw;
        if (col >= MH && col < SC - MH && ox < W && oy < H) {
            const long long o = (long long)oy * LD + ox;
            fo[o] = acc;
            fo[o + ps] = acc + 1.f;
        }
    }
}
template <int SC, int NT>
static void run_fused(const char* name, const float* R, const float* fl, float* fl2, long long ps, int np)
{
    constexpr int OUT = SC - 2 * MH;
    const int nseg = ((H + TH - 1) / TH + NT - 1) / NT;
    const dim3 grid((W + OUT - 1) / OUT, nseg, np);
    double best = 1e30;
    for (int rep = 0; rep < 2; rep++) {
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipLaunchKernelGGL((kfused<SC, NT>), grid, dim3(SC * 8), 0, 0, R, fl, fl2, ps);
        hipEventRecord(a);
        for (int i = 0; i < 10; i++) hipLaunchKernelGGL((kfused<SC, NT>), grid, dim3(SC * 8), 0, 0, R, fl, fl2, ps);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        best = ms * 1e3 / 10 < best ? ms * 1e3 / 10 : best;
    }
    printf("M never materialised, %-40s %8.1f us per %d pairs  %6.2f us/pair  %5.2f TB/s on 56 B/px  (%d workgroups of %d threads)\n",
           name, best, np, best / np, 56.0 * W * H * np / best / 1e6, (int)(grid.x * grid.y * grid.z), SC * 8);
}
// window loads only, in other shapes: VEC floats per lane and load (b32 / b64 / b128: a wave then covers 64 * VEC columns),
// ROWS output rows per tile (window = ROWS + 30 rows).  Same bytes per output row for a given ROWS.
template <int VEC, int ROWS>
__global__ __launch_bounds__(256) void kwin(const float* __restrict__ Min, float* __restrict__ out, long long ps)
{
    int bx, by, z;
    xcd_remap(bx, by, z);
    constexpr int TWV = 256 * VEC - 2 * HALO;
    const int x0 = bx * TWV - 16, y0 = by * ROWS, tid = threadIdx.x;
    const float* M = Min + (long long)z * 5 * ps;
    const int x = min(max(x0 - HALO + tid * VEC, 0), W - VEC);
    float acc = 0.f;
    if (x0 - HALO + tid * VEC <= W - 1 + MH) {
#pragma unroll
        for (int ch = 0; ch < 5; ch++) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(M + ch * ps), 0, 0xFFFFFFFFu, 0x00020000);
#pragma unroll
            for (int i = 0; i < ROWS + 2 * MH; i++) {
                const unsigned ro = (unsigned)min(max(y0 - MH + i, 0), H - 1) * (LD * 4u);
                if (VEC == 1) acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)x * 4u, ro, 0));
                if (VEC == 2) {
                    typedef float __attribute__((ext_vector_type(2))) f2;
                    const f2 v = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rs, (unsigned)x * 4u, ro, 0));
                    acc += v[0] + v[1];
                }
                if (VEC == 4) {
                    typedef float __attribute__((ext_vector_type(4))) f4;
                    const f4 v = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)x * 4u, ro, 0));
                    acc += v[0] + v[1] + v[2] + v[3];
                }
            }
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}
template <int VEC, int ROWS>
static void run_win(const char* name, const float* M0, float* fl, long long ps, int np)
{
    constexpr int TWV = 256 * VEC - 2 * HALO;
    const dim3 grid((W + 16 + TWV - 1) / TWV, (H + ROWS - 1) / ROWS, np);
    double best = 1e30;
    for (int rep = 0; rep < 2; rep++) {
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipLaunchKernelGGL((kwin<VEC, ROWS>), grid, dim3(256), 0, 0, M0, fl, ps);
        hipEventRecord(a);
        for (int i = 0; i < 10; i++) hipLaunchKernelGGL((kwin<VEC, ROWS>), grid, dim3(256), 0, 0, M0, fl, ps);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        best = ms * 1e3 / 10 < best ? ms * 1e3 / 10 : best;
    }
    const double l1_bytes = 5.0 * 4 * W * H * np * (double)(ROWS + 30) / ROWS;
    printf("window loads only, %-34s %8.1f us per 64 pairs  %6.2f us/pair  L2->L1 %5.1f TB/s\n", name, best, best / np, l1_bytes / best / 1e6);
}
template <typename F>
static double time_us(F launch, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch();
    launch();
    hipEventRecord(a);
    for (int i = 0; i < iters; i++) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e3 / iters;
}
int main()
{
    const int np = 64;
    const long long ps = (long long)LD * H;
    float *M0, *M1, *R, *fl;
    hipMalloc(&M0, ps * 5 * np * 4);
    hipMalloc(&M1, ps * 5 * np * 4);
    hipMalloc(&R, ps * 10 * np * 4);
    hipMalloc(&fl, ps * 2 * np * 4);
    hipMemset(M0, 0, ps * 5 * np * 4);
    hipMemset(R, 0, ps * 10 * np * 4);
    const dim3 grid((W + 16 + TW - 1) / TW, (H + TH - 1) / TH, np);
    const double px = (double)W * H * np;
    const char* names[3] = {"refreshing launch: windows + R0 + R1 + M out (80 B/px)", "last launch: windows + flow out (28 B/px)",
                            "refresh traffic alone: R0 + R1 + M out (60 B/px)"};
    const double bpp[3] = {80, 28, 60};
    for (int rep = 0; rep < 2; rep++) {
        double us[3];
        us[0] = time_us([&] { hipLaunchKernelGGL(k<0>, grid, dim3(256), 0, 0, M0, R, M1, fl, ps); }, 10);
        us[1] = time_us([&] { hipLaunchKernelGGL(k<1>, grid, dim3(256), 0, 0, M0, R, M1, fl, ps); }, 10);
        us[2] = time_us([&] { hipLaunchKernelGGL(k<2>, grid, dim3(256), 0, 0, M0, R, M1, fl, ps); }, 10);
        for (int i = 0; i < 3; i++)
            printf("%-58s %8.1f us per 64 pairs  %6.2f us/pair  %5.2f TB/s = %.3f of 8 TB/s\n", names[i], us[i], us[i] / np,
                   bpp[i] * px / us[i] / 1e6, bpp[i] * px / us[i] / 1e6 / 8.0);
    }
    {
        const double u = time_us([&] { hipLaunchKernelGGL(krefresh2, grid, dim3(256), 0, 0, R, M1, ps); }, 10);
        printf("%-58s %8.1f us per 64 pairs  %6.2f us/pair  %5.2f TB/s\n", "refresh traffic alone, 2 pixels per lane (8-byte accesses)", u,
               u / np, 60.0 * px / u / 1e6);
    }
    {
        // the as-built shape again with the real kernel's residency (40 KB of LDS -> 4 workgroups per CU)
        const double u = time_us([&] { hipLaunchKernelGGL(k<0>, grid, dim3(256), 40960, 0, M0, R, M1, fl, ps); }, 10);
        printf("%-58s %8.1f us per 64 pairs  %6.2f us/pair\n", "refreshing launch as built, 4 workgroups per CU", u, u / np);
    }
    run_roll<5, false, true>("refreshing, NT=5 (40 rows)", M0, R, M1, fl, ps, np);
    run_roll<15, false, true>("refreshing, NT=15 (120 rows)", M0, R, M1, fl, ps, np);
    run_roll<27, false, true>("refreshing, NT=27 (216 rows)", M0, R, M1, fl, ps, np);
    run_roll<45, false, true>("refreshing, NT=45 (360 rows)", M0, R, M1, fl, ps, np);
    run_roll<135, false, true>("refreshing, NT=135 (whole column)", M0, R, M1, fl, ps, np);
    run_roll<27, true, true>("refreshing, NT=27, new rows by LDS-DMA", M0, R, M1, fl, ps, np);
    run_roll<135, true, true>("refreshing, NT=135, new rows by LDS-DMA", M0, R, M1, fl, ps, np);
    run_roll<15, false, true>("refreshing, NT=15, 4 workgroups per CU", M0, R, M1, fl, ps, np, 40960);
    run_roll<27, false, true>("refreshing, NT=27, 4 workgroups per CU", M0, R, M1, fl, ps, np, 40960);
    run_roll<135, false, true>("refreshing, NT=135, 4 workgroups per CU", M0, R, M1, fl, ps, np, 40960);
    run_roll<27, false, true>("refreshing, NT=27, 2 workgroups per CU", M0, R, M1, fl, ps, np, 65536);
    run_roll<27, false, false>("last launch, NT=27", M0, R, M1, fl, ps, np);
    run_roll<135, false, false>("last launch, NT=135", M0, R, M1, fl, ps, np);
    float* fl2;
    hipMalloc(&fl2, ps * 2 * np * 4);
    hipMemset(fl, 0, ps * 2 * np * 4);
    run_fused<128, 27>("128-column strips (98 outputs), NT=27", R, fl, fl2, ps, np);
    run_fused<128, 15>("128-column strips (98 outputs), NT=15", R, fl, fl2, ps, np);
    run_fused<128, 135>("128-column strips (98 outputs), NT=135", R, fl, fl2, ps, np);
    run_fused<96, 27>("96-column strips (66 outputs), NT=27", R, fl, fl2, ps, np);
    run_fused<64, 27>("64-column strips (34 outputs), NT=27", R, fl, fl2, ps, np);
    run_win<1, 8>("b32, 8-row tiles (as built)", M0, fl, ps, np);
    run_win<2, 8>("b64, 8-row tiles", M0, fl, ps, np);
    run_win<4, 8>("b128, 8-row tiles", M0, fl, ps, np);
    run_win<1, 16>("b32, 16-row tiles", M0, fl, ps, np);
    run_win<2, 16>("b64, 16-row tiles", M0, fl, ps, np);
    run_win<1, 32>("b32, 32-row tiles", M0, fl, ps, np);
    return 0;
}
