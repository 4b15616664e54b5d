// twflow_kernels.hip.h — hand-written gfx950 (CDNA4, wave64) kernels of the Farneback image-diff path.
//
// Every kernel mirrors, operation by operation, the float/double arithmetic of OpenCV 2.4.9's CPU
// cv::calcOpticalFlowFarneback (the call at /root/reference/src/opticalflow.cpp:83-85) so that results are
// bit-identical to the CPU path; this translation unit is compiled with -ffp-contract=off and uses an
// explicit fma only where the product is exact in the wider type (so fused == unfused).
//
// Layout in HBM: every float image is a dense plane with row pitch `ld` (multiple of 32 floats); R (polynomial
// coefficients) and M (G11,G12,G22,h1,h2) are 5 planes at plane stride `ps`; flow is 2 planes.
// This is a stencil / gather path: HBM- and LDS-bound, no MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace twk {

typedef float __attribute__((ext_vector_type(4))) f32x4;
// explicit global-address-space pointers: keeps `global_load … v_off, s[base]` (SGPR base + 32-bit VGPR
// offset) selectable after the pointer has been made opaque to the optimiser
typedef const float __attribute__((address_space(1))) * gptr_cf;
typedef const char __attribute__((address_space(1))) * gptr_cc;

// wave-uniform pointer pinned to an SGPR pair and hidden from loop-invariant hoisting / strength reduction
__device__ __forceinline__ gptr_cf sgpr_base(const float* p)
{
    gptr_cf g = (gptr_cf)p;
    asm volatile("" : "+s"(g));
    return g;
}
__device__ __forceinline__ float gload(gptr_cf row, unsigned byte_off)
{
    return *(gptr_cf)((gptr_cc)row + byte_off);
}
typedef float __attribute__((address_space(1))) * gptr_f;
typedef char __attribute__((address_space(1))) * gptr_c;
__device__ __forceinline__ gptr_f sgpr_base_w(float* p)
{
    gptr_f g = (gptr_f)p;
    asm volatile("" : "+s"(g));
    return g;
}
__device__ __forceinline__ void gstore(gptr_f row, unsigned byte_off, float v) { *(gptr_f)((gptr_c)row + byte_off) = v; }
// Cache policy of the once-per-launch streams (round 4 A/B, -DTW_NT=<bits>): bit 0 the M planes a launch writes (read
// again only by the NEXT launch, a whole batch later), bit 1 the R0 coefficients (read once per launch), bit 2 the R planes
// polyexp writes.  Non-temporal = the `nt` bit of the global load / store: values are unchanged.
#ifndef TW_NT
#define TW_NT 0
#endif
template <int BIT, typename T>
__device__ __forceinline__ void st_stream(T* p, T v)
{
    if constexpr ((TW_NT >> BIT) & 1) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <int BIT>
__device__ __forceinline__ float ld_stream(const float* p)
{
    if constexpr ((TW_NT >> BIT) & 1) return __builtin_nontemporal_load(p);
    else return *p;
}

// Raw buffer access: one resource descriptor (4 SGPRs) per image set, a wave-uniform 32-bit byte offset in
// an SGPR (row / plane) and the lane's column as a 32-bit VGPR byte offset.  No per-load VALU address math
// and no 64-bit pointer per row (a 2N+1-row register window otherwise costs 2 SGPRs or 2 VGPRs per row).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p)
{
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0xFFFFFFFFu, 0x00020000);
}
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
typedef float __attribute__((ext_vector_type(2))) f32x2;
typedef unsigned __attribute__((ext_vector_type(2))) u32x2;
__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}
__device__ __forceinline__ f32x2 bload2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// XCD-aware workgroup remap.  Workgroups are dealt round-robin over the 8 XCDs (each with a private 4 MiB L2),
// so tiles that share halo rows land on 8 different L2s and every halo row is fetched from the Infinity
// Cache / HBM once per XCD.  Re-labelling gives every XCD one contiguous band of the (x fastest, then y, then
// pair) tile order, so vertically adjacent tiles hit in the same L2.  Bijective for any grid size; placement
// only affects speed, never results.
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

// The same bands, walked COLUMN-major (y fastest, then x, then pair): the tiles an XCD runs at any one time are then
// vertical neighbours.  For the 51-tap window at 4K (66-row register window per 16 output rows, 20 tiles per tile row) a
// row-major band puts 6.7 MB of other tiles' windows between a tile and the one below it — more than the XCD's 4 MB L2:
// rocprofv3 FETCH_SIZE + WRITE_SIZE read 1.5 x the 80 B/px the launch has to move (profiles/r05_cfg5_traffic.md).
__device__ __forceinline__ void xcd_remap_cm(int& bx, int& by, int& bz)
{
    const unsigned gx = gridDim.x, gy = gridDim.y, n = gx * gy * gridDim.z;
    unsigned b = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned xcd = b & 7u, q = n >> 3, r = n & 7u;
    b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    by = (int)(b % gy);
    const unsigned t = b / gy;
    bx = (int)(t % gx);
    bz = (int)(t / gx);
}

// A kernel BODY's view of its launch grid: the plain kernels hand their body the hardware grid, the twin kernels (two bodies
// in one launch, below) give each body a virtual one — gx x gy x gz workgroups numbered b = 0 .. n-1 in launch order, whose
// tile rows / images start at (y0, z0) of the body's full problem (a launch may cover a band of it).
struct VGrid {
    unsigned gx, gy, gz, y0, z0;
};
__device__ __forceinline__ VGrid hw_grid() { return VGrid{gridDim.x, gridDim.y, gridDim.z, 0u, 0u}; }
__device__ __forceinline__ unsigned hw_block() { return blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); }
// xcd_remap over a virtual grid (the b-th workgroup of the launch goes to XCD b & 7: a twin launch pads its first part to a
// multiple of 8 workgroups so that this also holds for the second part's own numbering)
__device__ __forceinline__ void xcd_remap_v(const VGrid& g, unsigned b, int& bx, int& by, int& bz)
{
    const unsigned n = g.gx * g.gy * g.gz;
    const unsigned xcd = b & 7u, q = n >> 3, r = n & 7u;
    b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    bx = (int)(b % g.gx);
    const unsigned t = b / g.gx;
    by = (int)(t % g.gy + g.y0);
    bz = (int)(t / g.gy + g.z0);
}

// borderInterpolate(p, len, BORDER_REFLECT_101)
__device__ __forceinline__ int reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

// =====================================================================================================
// K1  tw_pyr_level : u8 full-res image -> f32 pyramid level
//   convertTo(CV_32F) + GaussianBlur(smooth_sz, sigma, REFLECT_101) on the FULL-RES image + resize(INTER_LINEAR)
//   (OpenCV 2.4.9 optflowgf.cpp level loop).  Fused: the blur is evaluated only at the <=2x2 full-res samples
//   each output pixel reads.  Row filter of the needed source rows/columns goes to LDS, column filter +
//   bilinear (or 2x2 area-fast) combine reads it back.
// =====================================================================================================
constexpr int PYR_TW = 32;  // output tile
constexpr int PYR_TH = 8;
constexpr int PYR_MAXK = 128;

struct PyrArgs {
    const uint8_t* const* srcs;  // device table of image pointers, one per blockIdx.z
    float* dst;                  // level images, z-th at dst + z*dst_zs
    long long dst_zs;
    long long stride;  // bytes
    int w0, h0;
    int w, h, ld;
    const int* xofs;     // [w]   source column of the left tap
    const float* alpha;  // [2w]  (1-fx, fx)
    const int* yofs;     // [h]   UNCLIPPED source row of the top tap (rows are clipped, weights are not)
    const float* beta;   // [2h]
    const float* kern;   // [ksize] getGaussianKernel(ksize, sigma, CV_32F)
    int ksize;
    int mode;  // 0: same size (identity resize), 1: INTER_LINEAR, 2: 2x2 INTER_AREA fast path
    int xmax;  // dx >= xmax: single-tap columns (HResizeLinear tail loop)
    int nrows_max;
    int th;    // tw_pyr_level only: output rows per tile (PYR_TH, or 4 for the deep levels of a large image)
};

__global__ __launch_bounds__(256) void tw_pyr_level(PyrArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float pyr_sm[];
    float* skern = pyr_sm;               // [PYR_MAXK]
    float* rowbuf = pyr_sm + PYR_MAXK;   // [nrows][P]
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * PYR_TW, y0 = blockIdx.y * a.th;
    const uint8_t* __restrict__ src = a.srcs[blockIdx.z];
    float* __restrict__ dst = a.dst + blockIdx.z * a.dst_zs;
    const int ksize = a.ksize, r = ksize >> 1;
    const int P = (a.mode == 0) ? PYR_TW : 2 * PYR_TW;

    if (tid < ksize) skern[tid] = a.kern[tid];
    const int yA = y0, yB = min(y0 + a.th - 1, a.h - 1);
    const int ylo = clampi(a.yofs[yA], 0, a.h0 - 1) - r;
    const int yhi = clampi(a.yofs[yB] + 1, 0, a.h0 - 1) + r;
    const int nrows = yhi - ylo + 1;
    __syncthreads();

    // ---- row filter (float) of source rows [ylo,yhi] at the needed columns ----
    const float* kc = skern + r;
    // Round 6 — kernels of more than 5 taps (the coarsest levels of a deep pyramid whose regions are too large to stage:
    // BASELINE configs[4]'s 39- and 79-tap smoothing of a 4K image, 9.2 % of that configuration's GPU time): the loop below
    // fetched one byte per tap, each through reflect101 and each waited for.  Positions whose window lies inside the row take
    // their taps four at a time from unaligned dwords, same left-to-right sum; the FEW positions at the image's left / right
    // border (X is monotonic in p: a prefix [0, pL) and a suffix [pR, P) of the tile's positions) are left to the loop below,
    // which then runs over those positions only — in one loop the border lanes' byte-by-byte path would be executed by every
    // wave that holds one of them (measured: no gain at all that way).
    int pL = 0, pR = P;
    const bool wide_taps = ksize > 5;
    if (wide_taps) {
        auto Xof = [&](int p) {
            const int ox = min(x0 + ((a.mode == 0) ? p : (p >> 1)), a.w - 1);
            return (a.mode == 0) ? ox : (a.xofs[ox] + (p & 1));
        };
        while (pL < P && Xof(pL) - r < 0) pL++;
        while (pR > pL && Xof(pR - 1) - r + ksize + 3 > a.w0) pR--;
        typedef unsigned u32u __attribute__((aligned(1)));
        const int PI = pR - pL;  // interior positions of a row
        for (int it = tid; it < nrows * PI; it += 256) {
            const int rr = it / PI, p = pL + it - rr * PI;
            const int Y = reflect101(ylo + rr, a.h0);
            const uint8_t* __restrict__ q = src + (long long)Y * a.stride + (Xof(p) - r);
            // (the compiler merges four of these dwords into one 16-byte load; requesting ALL of a window's bytes before the
            // first tap — up to eight 16-byte loads in flight per lane — measured 20 % slower: gpurun_out/r6y)
            unsigned wd = *(const u32u*)q;
            float s = skern[0] * (float)(wd & 0xffu);
            s += skern[1] * (float)((wd >> 8) & 0xffu);
            s += skern[2] * (float)((wd >> 16) & 0xffu);
            s += skern[3] * (float)(wd >> 24);
            int j = 4;
            for (; j + 3 < ksize; j += 4) {
                wd = *(const u32u*)(q + j);
                s += skern[j] * (float)(wd & 0xffu);
                s += skern[j + 1] * (float)((wd >> 8) & 0xffu);
                s += skern[j + 2] * (float)((wd >> 16) & 0xffu);
                s += skern[j + 3] * (float)(wd >> 24);
            }
            for (; j < ksize; j++) s += skern[j] * (float)q[j];
            rowbuf[rr * P + p] = s;
        }
    }
    const int PB = wide_taps ? pL + (P - pR) : P;  // positions the general loop handles: the border ones, or all
    for (int it = tid; it < nrows * PB; it += 256) {
        const int rr = it / PB;
        int p = it - rr * PB;
        if (wide_taps && p >= pL) p += pR - pL;
        int ox = x0 + ((a.mode == 0) ? p : (p >> 1));
        ox = min(ox, a.w - 1);
        const int X = (a.mode == 0) ? ox : (a.xofs[ox] + (p & 1));
        const int Y = reflect101(ylo + rr, a.h0);
        const uint8_t* __restrict__ S = src + (long long)Y * a.stride;
        float s;
        if (ksize == 3) {
            const float c = (float)S[reflect101(X, a.w0)];
            const float l = (float)S[reflect101(X - 1, a.w0)], rt = (float)S[reflect101(X + 1, a.w0)];
            s = c * kc[0] + (l + rt) * kc[1];
        } else if (ksize == 5) {
            const float c = (float)S[reflect101(X, a.w0)];
            const float l1 = (float)S[reflect101(X - 1, a.w0)], r1 = (float)S[reflect101(X + 1, a.w0)];
            const float l2 = (float)S[reflect101(X - 2, a.w0)], r2 = (float)S[reflect101(X + 2, a.w0)];
            s = c * kc[0] + (l1 + r1) * kc[1] + (l2 + r2) * kc[2];
        } else if (ksize == 1) {
            s = (float)S[reflect101(X, a.w0)] * kc[0];
        } else {
            s = skern[0] * (float)S[reflect101(X - r, a.w0)];
            for (int j = 1; j < ksize; j++) s += skern[j] * (float)S[reflect101(X - r + j, a.w0)];
        }
        rowbuf[rr * P + p] = s;
    }
    __syncthreads();

    // ---- column filter at the sampled rows + resize combine ----
    const int tx = tid & (PYR_TW - 1), ty = tid / PYR_TW;
    const int ox = x0 + tx, oy = y0 + ty;
    if (ox >= a.w || oy >= a.h || ty >= a.th) return;
    const int sy = a.yofs[oy];
    const int s0 = clampi(sy, 0, a.h0 - 1) - ylo, s1 = clampi(sy + 1, 0, a.h0 - 1) - ylo;

    auto colf = [&](int srow, int p) -> float {
        const float* R = rowbuf + srow * P + p;
        if (ksize == 3) return (R[-P] + R[P]) * kc[1] + R[0] * kc[0];
        if (ksize == 1) return kc[0] * R[0];
        float s = kc[0] * R[0];
        for (int j = 1; j <= r; j++) s += kc[j] * (R[j * P] + R[-j * P]);
        return s;
    };

    float out;
    if (a.mode == 0) {
        out = colf(s0, tx);
    } else if (a.mode == 2) {
        float sum = 0.f;
        sum += colf(s0, 2 * tx) + colf(s0, 2 * tx + 1) + colf(s1, 2 * tx) + colf(s1, 2 * tx + 1);
        out = sum * 0.25f;
    } else {
        float t0, t1;
        if (ox < a.xmax) {
            const float a0 = a.alpha[2 * ox], a1 = a.alpha[2 * ox + 1];
            t0 = colf(s0, 2 * tx) * a0 + colf(s0, 2 * tx + 1) * a1;
            t1 = colf(s1, 2 * tx) * a0 + colf(s1, 2 * tx + 1) * a1;
        } else {
            t0 = colf(s0, 2 * tx) * 1.f;
            t1 = colf(s1, 2 * tx) * 1.f;
        }
        out = t0 * a.beta[2 * oy] + t1 * a.beta[2 * oy + 1];
    }
    dst[(long long)oy * a.ld + ox] = out;
}

// -----------------------------------------------------------------------------------------------------
// tw_pyr_level_lds : K1 with the u8 source region of the tile staged in LDS first (coalesced dword loads),
//   so the (2r+1)-tap row filter reads bytes from LDS instead of issuing one global byte load per tap.
//   Same tables, same arithmetic and order as tw_pyr_level.
// -----------------------------------------------------------------------------------------------------
struct PyrLdsArgs {
    PyrArgs p;
    int pitch_b;    // bytes per staged row (multiple of 4)
    int aligned4;
};

__global__ __launch_bounds__(256) void tw_pyr_level_lds(PyrLdsArgs aa)
{
    const PyrArgs& a = aa.p;
    extern __shared__ __attribute__((aligned(16))) float pyr_sm2[];
    float* skern = pyr_sm2;                                   // [PYR_MAXK]
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * PYR_TW, y0 = blockIdx.y * PYR_TH;
    const uint8_t* __restrict__ src = a.srcs[blockIdx.z];
    float* __restrict__ dst = a.dst + blockIdx.z * a.dst_zs;
    const int ksize = a.ksize, r = ksize >> 1;
    const int P = (a.mode == 0) ? PYR_TW : 2 * PYR_TW;
    const int yA = y0, yB = min(y0 + PYR_TH - 1, a.h - 1);
    const int ylo = clampi(a.yofs[yA], 0, a.h0 - 1) - r;
    const int yhi = clampi(a.yofs[yB] + 1, 0, a.h0 - 1) + r;
    const int nrows = yhi - ylo + 1;
    float* rowbuf = pyr_sm2 + PYR_MAXK;                               // [nrows_max][P]
    unsigned* tile = (unsigned*)(rowbuf + (size_t)a.nrows_max * P);   // [nrows_max][pitch_b/4]
    const uint8_t* tileb = (const uint8_t*)tile;
    // source columns of the tile: positions 0 .. P-1
    const int xl = min(x0, a.w - 1), xr = min(x0 + PYR_TW - 1, a.w - 1);
    const int Xfirst = (a.mode == 0) ? xl : a.xofs[xl];
    const int Xlast = (a.mode == 0) ? xr : a.xofs[xr] + 1;
    const int xlo_a = (Xfirst - r) & ~3;  // floor to a multiple of 4 (also for negatives)
    const int ndw = (Xlast + r - xlo_a) / 4 + 1;
    const int pd = aa.pitch_b >> 2;

    if (tid < ksize) skern[tid] = a.kern[tid];
    int* xpos = (int*)(skern + PYR_MAXK - 64);  // source column of each of the P positions (ksize <= 63 here)
    if (tid < P) {
        const int oxp = min(x0 + ((a.mode == 0) ? tid : (tid >> 1)), a.w - 1);
        xpos[tid] = (a.mode == 0) ? oxp : (a.xofs[oxp] + (tid & 1));
    }
    // ---- stage the u8 region: one dword per item, 8 items per thread in flight ----
    {
        const int total = nrows * ndw;
        const float inv_ndw = 1.0f / (float)ndw;
        // it / ndw without an integer divide (it < 2^20: the float estimate is off by at most one)
        auto divmod = [&](int it, int& rr, int& dw) {
            rr = (int)(((float)it + 0.5f) * inv_ndw);
            dw = it - rr * ndw;
            if (dw < 0) { rr--; dw += ndw; }
            if (dw >= ndw) { rr++; dw -= ndw; }
        };
        for (int base = tid; base < total; base += 256 * 8) {
            unsigned v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int it = base + u * 256;
                v[u] = 0;
                if (it < total) {
                    int rr, dw;
                    divmod(it, rr, dw);
                    const int Y = reflect101(ylo + rr, a.h0);
                    const uint8_t* __restrict__ S = src + (long long)Y * a.stride;
                    const int c0 = xlo_a + 4 * dw;
                    if (aa.aligned4 && c0 >= 0 && c0 + 3 < a.w0) {
                        v[u] = *(const unsigned*)(S + c0);
                    } else {
                        v[u] = (unsigned)S[reflect101(c0, a.w0)] | ((unsigned)S[reflect101(c0 + 1, a.w0)] << 8) |
                               ((unsigned)S[reflect101(c0 + 2, a.w0)] << 16) |
                               ((unsigned)S[reflect101(c0 + 3, a.w0)] << 24);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int it = base + u * 256;
                if (it < total) {
                    int rr, dw;
                    divmod(it, rr, dw);
                    tile[rr * pd + dw] = v[u];
                }
            }
        }
    }
    __syncthreads();

    // ---- row filter from LDS bytes ----
    const float* kc = skern + r;
    const int lgP = (a.mode == 0) ? 5 : 6;  // P = 32 or 64
    const int p = tid & (P - 1);
    const int X = xpos[p];
    for (int rr = tid >> lgP; rr < nrows; rr += (256 >> lgP)) {
        const uint8_t* T = tileb + rr * aa.pitch_b + (X - r - xlo_a);  // tap j at T[j]
        float s;
        if (ksize == 3) {
            s = (float)T[1] * kc[0] + ((float)T[0] + (float)T[2]) * kc[1];
        } else if (ksize == 5) {
            s = (float)T[2] * kc[0] + ((float)T[1] + (float)T[3]) * kc[1] + ((float)T[0] + (float)T[4]) * kc[2];
        } else if (ksize == 1) {
            s = (float)T[0] * kc[0];
        } else {
            // left-to-right accumulation; taps are fetched 8 at a time so that the LDS reads of a group are
            // in flight together (a tap-by-tap loop is bound by one LDS round trip per tap)
            s = skern[0] * (float)T[0];
            int j0 = 1;
            for (; j0 + 8 <= ksize; j0 += 8) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; u++) t[u] = (float)T[j0 + u];
#pragma unroll
                for (int u = 0; u < 8; u++) s += skern[j0 + u] * t[u];
            }
            for (; j0 < ksize; j0++) s += skern[j0] * (float)T[j0];
        }
        rowbuf[rr * P + p] = s;
    }
    __syncthreads();

    // ---- column filter at the sampled rows + resize combine (as tw_pyr_level) ----
    const int tx = tid & (PYR_TW - 1), ty = tid / PYR_TW;
    const int ox = x0 + tx, oy = y0 + ty;
    if (ox >= a.w || oy >= a.h) return;
    const int sy = a.yofs[oy];
    const int s0 = clampi(sy, 0, a.h0 - 1) - ylo, s1 = clampi(sy + 1, 0, a.h0 - 1) - ylo;
    auto colf = [&](int srow, int p) -> float {
        const float* R = rowbuf + srow * P + p;
        if (ksize == 3) return (R[-P] + R[P]) * kc[1] + R[0] * kc[0];
        if (ksize == 1) return kc[0] * R[0];
        float s = kc[0] * R[0];
        int j = 1;
        for (; j + 4 <= r + 1; j += 4) {
            float up[4], dn[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                up[u] = R[(j + u) * P];
                dn[u] = R[-(j + u) * P];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) s += kc[j + u] * (up[u] + dn[u]);
        }
        for (; j <= r; j++) s += kc[j] * (R[j * P] + R[-j * P]);
        return s;
    };
    float out;
    if (a.mode == 0) {
        out = colf(s0, tx);
    } else if (a.mode == 2) {
        float sum = 0.f;
        sum += colf(s0, 2 * tx) + colf(s0, 2 * tx + 1) + colf(s1, 2 * tx) + colf(s1, 2 * tx + 1);
        out = sum * 0.25f;
    } else {
        float t0, t1;
        if (ox < a.xmax) {
            const float a0 = a.alpha[2 * ox], a1 = a.alpha[2 * ox + 1];
            t0 = colf(s0, 2 * tx) * a0 + colf(s0, 2 * tx + 1) * a1;
            t1 = colf(s1, 2 * tx) * a0 + colf(s1, 2 * tx + 1) * a1;
        } else {
            t0 = colf(s0, 2 * tx) * 1.f;
            t1 = colf(s1, 2 * tx) * 1.f;
        }
        out = t0 * a.beta[2 * oy] + t1 * a.beta[2 * oy + 1];
    }
    dst[(long long)oy * a.ld + ox] = out;
}

// -----------------------------------------------------------------------------------------------------
// tw_pyr_taps : K1 for the coarse levels (resizing modes, 7 <= ksize <= 63), where the row filter dominates.
//   Same tile, tables, float operations and order as tw_pyr_level_lds; what changes is how the taps are fed:
//   * staging: one wave per source row, one dword per lane through a buffer load (no divide, no 64-bit
//     address math), four rows in flight;
//   * row filter: one thread produces BOTH source columns X and X+1 an output pixel samples; their tap
//     windows overlap in all but one byte, so the bytes are read once, as aligned LDS dwords realigned with
//     v_alignbyte and unpacked with v_cvt_f32_ubyteN;
//   * coefficients come from the kernel arguments (scalar loads).  kext = [0, k0 .. k(ksize-1), 0 ...]:
//     window byte j feeds column X with kext[j+1] and column X+1 with kext[j]; the zero entries make the
//     padded taps add +0 to a non-negative sum, which leaves it unchanged.
// -----------------------------------------------------------------------------------------------------
constexpr int PYR_KEXT = 72;
struct PyrTapsArgs {
    PyrArgs p;
    int pd;  // dwords per staged row (odd: the two rows a wave filters start on different banks)
    int aligned4;
    unsigned long long* dbg;  // diagnostic build only (TW_DEBUG_STAMPS): s_memtime at the phase boundaries of the first 64 workgroups
    float kext[PYR_KEXT];
};

// KS = the smoothing kernel size when it is one of the sizes a pyr_scale = 0.5 pyramid uses (9, 19, 39: the tap loops
// unroll and their LDS reads are issued back to back instead of one round trip per group of taps), 0 = any size.
template <int KS>
__global__ __launch_bounds__(256) void tw_pyr_taps(PyrTapsArgs aa)
{
    const PyrArgs& a = aa.p;
    extern __shared__ __attribute__((aligned(16))) float pyr_sm3[];
    constexpr int P = 2 * PYR_TW;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int x0 = blockIdx.x * PYR_TW, y0 = blockIdx.y * PYR_TH;
    const uint8_t* __restrict__ src = a.srcs[blockIdx.z];
    float* __restrict__ dst = a.dst + blockIdx.z * a.dst_zs;
    if (aa.dbg && tid == 0 && blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z) < 64)
        aa.dbg[(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 4] = __builtin_amdgcn_s_memtime();
    const int ksize = KS ? KS : a.ksize, r = ksize >> 1;
    const int yA = y0, yB = min(y0 + PYR_TH - 1, a.h - 1);
    const int ylo = clampi(a.yofs[yA], 0, a.h0 - 1) - r;
    const int yhi = clampi(a.yofs[yB] + 1, 0, a.h0 - 1) + r;
    const int nrows = yhi - ylo + 1;
    float* rowbuf = pyr_sm3;                                             // [nrows_max][P]
    unsigned* tile = (unsigned*)(rowbuf + (size_t)a.nrows_max * P);      // [nrows_max][pd] (+ 4 dwords slack)
    const int xl = min(x0, a.w - 1), xr = min(x0 + PYR_TW - 1, a.w - 1);
    const int Xfirst = a.xofs[xl], Xlast = a.xofs[xr] + 1;
    const int xlo_a = (Xfirst - r) & ~3;
    const int ndw = (Xlast + r - xlo_a) / 4 + 1;
    const int pd = aa.pd;

    // ---- stage the u8 region ----
    // A wave takes the rows wv, wv+4, ...; a lane the dwords lane, lane+64 of a row.  Eight rows x two dwords are
    // loaded before anything is stored, so a tile pays two or three memory round trips instead of one per four
    // rows and per 64 dwords (the coarse levels of a single pair are latency-bound: BASELINE config 2).
    {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(src);
        const int stride = (int)a.stride;
        constexpr int NB = 8;
        for (int dwb = lane; dwb < ndw; dwb += 128) {
            int c0[2], cc[2][4];
            bool fast[2], live[2];
#pragma unroll
            for (int d = 0; d < 2; d++) {
                const int dw = dwb + 64 * d;
                live[d] = dw < ndw;
                c0[d] = xlo_a + 4 * dw;
                fast[d] = aa.aligned4 && c0[d] >= 0 && c0[d] + 3 < a.w0;
#pragma unroll
                for (int u = 0; u < 4; u++) cc[d][u] = reflect101(c0[d] + u, a.w0);
            }
            for (int rb = wv; rb < nrows; rb += 4 * NB) {
                unsigned v[NB][2];
#pragma unroll
                for (int u = 0; u < NB; u++) {
                    const int rr = rb + 4 * u;
                    const int Y = reflect101(ylo + min(rr, nrows - 1), a.h0);
                    const unsigned ro = (unsigned)(Y * stride);
#pragma unroll
                    for (int d = 0; d < 2; d++) {
                        v[u][d] = 0;
                        if (!live[d]) continue;
                        if (fast[d]) {
                            v[u][d] = __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)c0[d] + ro, 0, 0);
                        } else {
                            const uint8_t* __restrict__ S = src + ro;
                            v[u][d] = (unsigned)S[cc[d][0]] | ((unsigned)S[cc[d][1]] << 8) | ((unsigned)S[cc[d][2]] << 16) |
                                      ((unsigned)S[cc[d][3]] << 24);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < NB; u++) {
                    const int rr = rb + 4 * u;
                    if (rr < nrows) {
                        if (live[0]) tile[rr * pd + dwb] = v[u][0];
                        if (live[1]) tile[rr * pd + dwb + 64] = v[u][1];
                    }
                }
            }
        }
    }
    __syncthreads();

    const unsigned blin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const bool stamp = aa.dbg && tid == 0 && blin < 64;
    if (stamp) aa.dbg[blin * 4 + 1] = __builtin_amdgcn_s_memtime();
    // ---- row filter: columns X and X+1 of one output pixel per thread ----
    {
        const int q = tid & 31;
        const int oxp = min(x0 + q, a.w - 1);
        const int o = a.xofs[oxp] - r - xlo_a;  // first window byte, >= 0
        const int sh = o & 3;
        const int ngroups = (ksize + 4) >> 2;   // window = ksize + 1 bytes
        for (int rr = tid >> 5; rr < nrows; rr += 8) {
            const unsigned* D = tile + rr * pd + (o >> 2);
            unsigned lo = D[0];
            float sA = 0.f, sB = 0.f;
#pragma unroll
            for (int g = 0; g < ngroups; g++) {
                const unsigned hi = D[g + 1];
                const unsigned wd = __builtin_amdgcn_alignbyte(hi, lo, sh);
                lo = hi;
                const float t0 = (float)(wd & 0xffu), t1 = (float)((wd >> 8) & 0xffu);
                const float t2 = (float)((wd >> 16) & 0xffu), t3 = (float)(wd >> 24);
                const float* kk = aa.kext + 4 * g;
                sA += kk[1] * t0;
                sB += kk[0] * t0;
                sA += kk[2] * t1;
                sB += kk[1] * t1;
                sA += kk[3] * t2;
                sB += kk[2] * t2;
                sA += kk[4] * t3;
                sB += kk[3] * t3;
            }
            f32x2 o2;
            o2.x = sA;
            o2.y = sB;
            *(f32x2*)(rowbuf + rr * P + 2 * q) = o2;
        }
    }
    __syncthreads();

    if (stamp) aa.dbg[blin * 4 + 2] = __builtin_amdgcn_s_memtime();
    // ---- column filter at the sampled rows + resize combine (as tw_pyr_level) ----
    const int tx = tid & (PYR_TW - 1), ty = tid / PYR_TW;
    const int ox = x0 + tx, oy = y0 + ty;
    if (ox >= a.w || oy >= a.h) return;
    const int sy = a.yofs[oy];
    const int s0 = clampi(sy, 0, a.h0 - 1) - ylo, s1 = clampi(sy + 1, 0, a.h0 - 1) - ylo;
    const float* kc = aa.kext + 1 + r;
    // both columns of a sample row at once (one 8-byte LDS read per tap row)
    auto colf2 = [&](int srow) -> f32x2 {
        const float* R = rowbuf + srow * P + 2 * tx;
        const f32x2 c = *(const f32x2*)R;
        float sa = kc[0] * c.x, sb = kc[0] * c.y;
#pragma unroll
        for (int j = 1; j <= r; j++) {
            const f32x2 up = *(const f32x2*)(R + j * P), dn = *(const f32x2*)(R - j * P);
            sa += kc[j] * (up.x + dn.x);
            sb += kc[j] * (up.y + dn.y);
        }
        f32x2 o2;
        o2.x = sa;
        o2.y = sb;
        return o2;
    };
    const f32x2 c0 = colf2(s0), c1 = colf2(s1);
    float out;
    if (a.mode == 2) {
        float sum = 0.f;
        sum += c0.x + c0.y + c1.x + c1.y;
        out = sum * 0.25f;
    } else {
        float t0, t1;
        if (ox < a.xmax) {
            const float a0 = a.alpha[2 * ox], a1 = a.alpha[2 * ox + 1];
            t0 = c0.x * a0 + c0.y * a1;
            t1 = c1.x * a0 + c1.y * a1;
        } else {
            t0 = c0.x * 1.f;
            t1 = c1.x * 1.f;
        }
        out = t0 * a.beta[2 * oy] + t1 * a.beta[2 * oy + 1];
    }
    dst[(long long)oy * a.ld + ox] = out;
    if (stamp) aa.dbg[blin * 4 + 3] = __builtin_amdgcn_s_memtime();
}

// -----------------------------------------------------------------------------------------------------
// tw_pyr_23 : K1 for pyramid levels 2 AND 3 of an exact pyr_scale = 0.5 pyramid from ONE read of the u8 image (round 5).
//   Levels 2 and 3 (9-tap / 19-tap smoothing of the FULL-RES image, INTER_LINEAR down to 1/4 and 1/8) used to be two
//   launches of tw_pyr_taps over 32 x 8-pixel output tiles: each staged its own 136..274-byte wide region of the same
//   image (1.9-2.6 x the algorithmic traffic, profiles/r04_pmc_hbm_traffic.md) and ran at 0.09-0.14 of the HBM roofline.
//   Here a workgroup stages a 252 x 52 byte region ONCE — 64 aligned dwords per row, one coalesced 256-byte row segment
//   per wave instruction — and produces the 30 x 5 level-3 pixels AND the 60 x 10 level-2 pixels of its 240 x 40 source
//   tile: row filter at the two source columns each output samples (shared byte window, v_alignbyte + v_cvt_f32_ubyteN,
//   exactly tw_pyr_taps' scheme) for every staged row, column filter at the two sampled rows, bilinear combine with the
//   0.5 / 0.5 weights the exact ratio gives.  Same float operations in the same order as tw_pyr_taps / the CPU:
//   RowFilter left to right, SymmColumnFilter centre-out, (c00*a0 + c01*a1)*b0 + (c10*a0 + c11*a1)*b1.
//   The host launches it only when its resize tables say xofs[x] = s*x + s/2 - 1, alpha = beta = 0.5 for s = 4 and 8.
//   Level 3 samples columns 8X+3, 8X+4 (+-9), level 2 columns 4x+1, 4x+2 (+-4); rows alike.
// -----------------------------------------------------------------------------------------------------
constexpr int P23_T3W = 30, P23_T3H = 5;          // level-3 tile; level 2: 60 x 10; source: 240 x 40
constexpr int P23_PD = 66;                        // dwords per staged row (64 + 2: even, so the b64 reads stay aligned)
constexpr int P23_R3 = 8 * P23_T3H + 12;          // 52 staged rows [r0 - 6, r0 + 46)
constexpr int P23_R2 = 8 * P23_T3H + 6;           // 46 of them, [r0 - 3, r0 + 43), feed level 2
struct Pyr23Args {
    const uint8_t* const* srcs;  // device table of image pointers, one per blockIdx.z
    float* dst3;                 // level-3 images, z-th at dst3 + z*zs3
    float* dst2;
    long long zs3, zs2;
    long long stride;  // bytes
    int w0, h0;
    int w3, h3, ld3, w2, h2, ld2;
    int aligned4;
    float k19[24];  // [0, k0 .. k18, 0, ...]: window byte j feeds column X with k19[j+1] and column X+1 with k19[j]
    float k9[16];   // [0, k0 .. k8, 0, ...]
};

// NT threads per workgroup (256 or 512: the same tile and LDS, twice the waves to cover the staging latency)
template <int NT>
__global__ __launch_bounds__(NT) void tw_pyr_23(Pyr23Args a)
{
    __shared__ __attribute__((aligned(16))) unsigned tile[P23_R3][P23_PD];
    __shared__ __attribute__((aligned(16))) float rb3[P23_R3][2 * P23_T3W];
    __shared__ __attribute__((aligned(16))) float rb2[P23_R2][4 * P23_T3W];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int bx, by, bz;
    xcd_remap(bx, by, bz);  // an XCD works on a contiguous band of tiles: the halo rows / cache lines two tiles share hit in its L2
    const uint8_t* __restrict__ src = a.srcs[bz];
    const int c0 = bx * (8 * P23_T3W), r0 = by * (8 * P23_T3H);

    // ---- stage the u8 region: rows r0-6 .. r0+45 (REFLECT101), columns c0-8 .. c0+247 as 64 dwords ----
    {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(src);
        const int stride = (int)a.stride;
        const int col = c0 - 8 + 4 * lane;
        const bool fast = a.aligned4 && col >= 0 && col + 3 < a.w0;
        int cc[4];
#pragma unroll
        for (int u = 0; u < 4; u++) cc[u] = reflect101(col + u, a.w0);
        constexpr int NWV = NT / 64;
        constexpr int NR = (P23_R3 + NWV - 1) / NWV;  // 13 (7) rows per wave, all loaded before anything is stored
        unsigned v[NR];
#pragma unroll
        for (int u = 0; u < NR; u++) {
            const int Y = reflect101(r0 - 6 + min(wv + NWV * u, P23_R3 - 1), a.h0);
            const unsigned ro = (unsigned)(Y * stride);
            if (fast) {
                v[u] = __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)col + ro, 0, 0);
            } else {
                const uint8_t* __restrict__ S = src + ro;
                v[u] = (unsigned)S[cc[0]] | ((unsigned)S[cc[1]] << 8) | ((unsigned)S[cc[2]] << 16) | ((unsigned)S[cc[3]] << 24);
            }
        }
#pragma unroll
        for (int u = 0; u < NR; u++)
            if (wv + NWV * u < P23_R3) tile[wv + NWV * u][lane] = v[u];
    }
    __syncthreads();

    // ---- row filter, level 3: columns 8x+3 and 8x+4 of output x, every staged row (1 560 tasks) ----
    for (int t = tid; t < P23_R3 * P23_T3W; t += NT) {
        const int rr = t / P23_T3W, x = t - rr * P23_T3W;
        // window = 20 bytes from staged byte 8x+2: dwords 2x .. 2x+5, shifted by 2
        const u32x2* D = (const u32x2*)&tile[rr][2 * x];
        const u32x2 d0 = D[0], d1 = D[1], d2 = D[2];
        const unsigned dw[6] = {d0.x, d0.y, d1.x, d1.y, d2.x, d2.y};
        float sA = 0.f, sB = 0.f;
#pragma unroll
        for (int g = 0; g < 5; g++) {
            const unsigned wd = __builtin_amdgcn_alignbyte(dw[g + 1], dw[g], 2);
            const float t0 = (float)(wd & 0xffu), t1 = (float)((wd >> 8) & 0xffu);
            const float t2 = (float)((wd >> 16) & 0xffu), t3 = (float)(wd >> 24);
            const float* kk = a.k19 + 4 * g;
            sA += kk[1] * t0;
            sB += kk[0] * t0;
            sA += kk[2] * t1;
            sB += kk[1] * t1;
            sA += kk[3] * t2;
            sB += kk[2] * t2;
            sA += kk[4] * t3;
            sB += kk[3] * t3;
        }
        f32x2 o2;
        o2.x = sA;
        o2.y = sB;
        *(f32x2*)&rb3[rr][2 * x] = o2;
    }
    // ---- row filter, level 2: columns 4x+1 and 4x+2 of output x, staged rows 3 .. 48 (2 760 tasks) ----
    for (int t = tid; t < P23_R2 * 2 * P23_T3W; t += NT) {
        const int rr = t / (2 * P23_T3W), x = t - rr * (2 * P23_T3W);
        // window = 10 bytes from staged byte 4x+5: dwords x+1 .. x+3, shifted by 1 (the bytes past the window weigh 0)
        const unsigned* D = &tile[rr + 3][x + 1];
        const unsigned dw[4] = {D[0], D[1], D[2], 0u};
        float sA = 0.f, sB = 0.f;
#pragma unroll
        for (int g = 0; g < 3; g++) {
            const unsigned wd = __builtin_amdgcn_alignbyte(dw[g + 1], dw[g], 1);
            const float t0 = (float)(wd & 0xffu), t1 = (float)((wd >> 8) & 0xffu);
            const float t2 = (float)((wd >> 16) & 0xffu), t3 = (float)(wd >> 24);
            const float* kk = a.k9 + 4 * g;
            sA += kk[1] * t0;
            sB += kk[0] * t0;
            sA += kk[2] * t1;
            sB += kk[1] * t1;
            if (g < 2) {  // window bytes 10, 11 lie past both columns' taps
                sA += kk[3] * t2;
                sB += kk[2] * t2;
                sA += kk[4] * t3;
                sB += kk[3] * t3;
            }
        }
        f32x2 o2;
        o2.x = sA;
        o2.y = sB;
        *(f32x2*)&rb2[rr][2 * x] = o2;
    }
    __syncthreads();

    // ---- column filter at the two sampled rows + bilinear combine ----
    // both columns of a sample row at once (one 8-byte LDS read per tap row): k0*c + sum k_j*(R[+j] + R[-j])
    if (tid < P23_T3W * P23_T3H) {
        const int y = tid / P23_T3W, x = tid - y * P23_T3W;
        const int ox = bx * P23_T3W + x, oy = by * P23_T3H + y;
        const float* kc = a.k19 + 1 + 9;
        f32x2 c[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const float* R = &rb3[8 * y + 9 + q][2 * x];
            const f32x2 ce = *(const f32x2*)R;
            float sa = kc[0] * ce.x, sb = kc[0] * ce.y;
#pragma unroll
            for (int j = 1; j <= 9; j++) {
                const f32x2 up = *(const f32x2*)(R + j * (2 * P23_T3W)), dn = *(const f32x2*)(R - j * (2 * P23_T3W));
                sa += kc[j] * (up.x + dn.x);
                sb += kc[j] * (up.y + dn.y);
            }
            c[q].x = sa;
            c[q].y = sb;
        }
        const float t0 = c[0].x * 0.5f + c[0].y * 0.5f, t1 = c[1].x * 0.5f + c[1].y * 0.5f;
        if (ox < a.w3 && oy < a.h3) a.dst3[bz * a.zs3 + (long long)oy * a.ld3 + ox] = t0 * 0.5f + t1 * 0.5f;
    }
    // level 2: 600 outputs; 256 threads: the third round goes to the threads that had no level-3 pixel; 512 threads: the
    // second round (88 outputs) likewise
    // (1024 threads — the single-pair schedule: one round, by the 600 threads behind the 150 that had a level-3 pixel)
    constexpr int NRND = NT == 256 ? 3 : NT == 512 ? 2 : 1, LAST0 = NT == 256 ? 168 : NT == 512 ? 424 : P23_T3W * P23_T3H;
    for (int i = 0; i < NRND; i++) {
        const int t = i < NRND - 1 ? tid + NT * i : tid - LAST0 + NT * (NRND - 1);
        if (i == NRND - 1 && tid < LAST0) break;
        if (NT > 512 && t >= 4 * P23_T3W * P23_T3H) break;
        const int y = t / (2 * P23_T3W), x = t - y * (2 * P23_T3W);
        const int ox = bx * (2 * P23_T3W) + x, oy = by * (2 * P23_T3H) + y;
        const float* kc = a.k9 + 1 + 4;
        f32x2 c[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const float* R = &rb2[4 * y + 4 + q][2 * x];
            const f32x2 ce = *(const f32x2*)R;
            float sa = kc[0] * ce.x, sb = kc[0] * ce.y;
#pragma unroll
            for (int j = 1; j <= 4; j++) {
                const f32x2 up = *(const f32x2*)(R + j * (4 * P23_T3W)), dn = *(const f32x2*)(R - j * (4 * P23_T3W));
                sa += kc[j] * (up.x + dn.x);
                sb += kc[j] * (up.y + dn.y);
            }
            c[q].x = sa;
            c[q].y = sb;
        }
        const float t0 = c[0].x * 0.5f + c[0].y * 0.5f, t1 = c[1].x * 0.5f + c[1].y * 0.5f;
        if (ox < a.w2 && oy < a.h2) a.dst2[bz * a.zs2 + (long long)oy * a.ld2 + ox] = t0 * 0.5f + t1 * 0.5f;
    }
}

// -----------------------------------------------------------------------------------------------------
// tw_pyr_k3<MODE> : register-only fast path of K1 for 3-tap smoothing (the two finest levels of a
//   pyr_scale = 0.5 pyramid: MODE 0 = level 0, same size; MODE 2 = level 1, exact 2x2 area-fast resize).
//   A thread reads whole dwords of the u8 rows (coalesced), converts with v_cvt_f32_ubyteN and produces
//   4 x 2 (MODE 0) or 4 x 1 (MODE 2) output pixels, stored as 16-byte vectors.  Same float operation order
//   as the generic kernel: row filter S0*k0 + (S-1 + S+1)*k1, column filter (R-1 + R+1)*k1 + R0*k0.
// -----------------------------------------------------------------------------------------------------
struct PyrK3Args {
    const uint8_t* const* srcs;
    float* dst;
    long long dst_zs;
    long long stride;
    int w0, h0, w, h, ld;
    float k0, k1;  // centre / side tap
    int aligned4;  // rows start 4-byte aligned: whole-dword loads allowed
};

// v[j] = (float)S[reflect101(xs + j)], j = 0..NV-1, where xs = 4*m - 1 (so xs+1 is dword aligned)
template <int NV>
__device__ __forceinline__ void pyr_load_row(const uint8_t* __restrict__ S, int xs, int w0, bool fast, float* v)
{
    if (fast) {
        // dwords at xs-3, xs+1, xs+5, ... : last byte of the first, all of the middle ones, first of the last
        const unsigned* D = (const unsigned*)(S + xs - 3);
        constexpr int ND = (NV + 2) / 4 + 1;
        unsigned d[ND];
#pragma unroll
        for (int i = 0; i < ND; i++) d[i] = D[i];
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const int b = j + 3;  // byte index from xs-3
            v[j] = (float)((d[b >> 2] >> (8 * (b & 3))) & 0xffu);
        }
    } else {
#pragma unroll
        for (int j = 0; j < NV; j++) v[j] = (float)S[reflect101(xs + j, w0)];
    }
}

// own = the lane's aligned u8 word(s) of a row; the byte left of them comes from lane-1's last word and the byte
// right of them from lane+1's first word (DPP wave shifts) — every source byte is requested from memory once
// instead of three times.  Wave edges, row ends and unaligned images fall back to byte loads (REFLECT101).
__device__ __forceinline__ unsigned dpp_from_prev(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false);  // wave_shr:1
}
__device__ __forceinline__ unsigned dpp_from_next(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false);  // wave_shl:1
}

template <int MODE>
__global__ __launch_bounds__(256) void tw_pyr_k3(PyrK3Args a)
{
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // tx = lane: a wave is 64 consecutive 4-pixel groups
    const int ox = (blockIdx.x * 64 + tx) * 4;
    const uint8_t* __restrict__ src = a.srcs[blockIdx.z];
    float* __restrict__ dst = a.dst + blockIdx.z * a.dst_zs;
    const bool in_w = ox < a.w;  // lanes past the row end stay active for the lane exchange, they only skip memory
    const float k0 = a.k0, k1 = a.k1;
    if (MODE == 0) {
        const int oy = (blockIdx.y * 4 + ty) * 2;
        if (oy >= a.h) return;  // wave-uniform
        const bool own_ok = a.aligned4 && ox + 3 < a.w0;
        const bool left_ok = tx > 0 && a.aligned4 && ox - 1 < a.w0;
        const bool right_ok = tx < 63 && a.aligned4 && ox + 7 < a.w0;
        float rf[4][4];
        // all loads of the four rows first (own word, plus the edge bytes of the lanes that have no neighbour)
        unsigned own[4], lb[4], rb[4];
        const bool need_l = in_w && !left_ok, need_r = in_w && !right_ok, need_own = in_w && !own_ok;
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int Y = reflect101(oy - 1 + rr, a.h0);
            const uint8_t* __restrict__ S = src + (long long)Y * a.stride;
            own[rr] = 0;
            lb[rr] = rb[rr] = 0;
            if (own_ok) own[rr] = *(const unsigned*)(S + ox);
            if (need_l) lb[rr] = S[reflect101(ox - 1, a.w0)];
            if (need_r) rb[rr] = S[reflect101(ox + 4, a.w0)];
            if (need_own) {
#pragma unroll
                for (int j = 0; j < 4; j++) own[rr] |= (unsigned)S[reflect101(ox + j, a.w0)] << (8 * j);
            }
        }
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const unsigned lw = dpp_from_prev(own[rr]), rw = dpp_from_next(own[rr]);
            float v[6];
            v[0] = need_l ? (float)lb[rr] : (float)(lw >> 24);
            v[1] = (float)(own[rr] & 0xffu);
            v[2] = (float)((own[rr] >> 8) & 0xffu);
            v[3] = (float)((own[rr] >> 16) & 0xffu);
            v[4] = (float)(own[rr] >> 24);
            v[5] = need_r ? (float)rb[rr] : (float)(rw & 0xffu);
#pragma unroll
            for (int j = 0; j < 4; j++) rf[rr][j] = v[j + 1] * k0 + (v[j] + v[j + 2]) * k1;
        }
        if (!in_w) return;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            if (oy + q >= a.h) break;
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; j++) o[j] = (rf[q][j] + rf[q + 2][j]) * k1 + rf[q + 1][j] * k0;
            float* d = dst + (long long)(oy + q) * a.ld + ox;
            if (ox + 3 < a.w) *(f32x4*)d = f32x4{o[0], o[1], o[2], o[3]};
            else
                for (int j = 0; j < 4; j++)
                    if (ox + j < a.w) d[j] = o[j];
        }
    } else {
        const int oy = blockIdx.y * 4 + ty;
        if (oy >= a.h) return;  // wave-uniform
        const int sx = 2 * ox;  // first source column of the 8 blurred samples
        const bool own_ok = a.aligned4 && sx + 7 < a.w0;
        const bool left_ok = tx > 0 && a.aligned4 && sx - 1 < a.w0;
        const bool right_ok = tx < 63 && a.aligned4 && sx + 15 < a.w0;
        float rf[4][8];
        unsigned o0[4], o1[4], lb[4], rb[4];
        const bool need_l = in_w && !left_ok, need_r = in_w && !right_ok, need_own = in_w && !own_ok;
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int Y = reflect101(2 * oy - 1 + rr, a.h0);
            const uint8_t* __restrict__ S = src + (long long)Y * a.stride;
            o0[rr] = o1[rr] = lb[rr] = rb[rr] = 0;
            if (own_ok) {
                const u32x2 w2 = *(const u32x2*)(S + sx);
                o0[rr] = w2.x;
                o1[rr] = w2.y;
            }
            if (need_l) lb[rr] = S[reflect101(sx - 1, a.w0)];
            if (need_r) rb[rr] = S[reflect101(sx + 8, a.w0)];
            if (need_own) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    o0[rr] |= (unsigned)S[reflect101(sx + j, a.w0)] << (8 * j);
                    o1[rr] |= (unsigned)S[reflect101(sx + 4 + j, a.w0)] << (8 * j);
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const unsigned lw = dpp_from_prev(o1[rr]), rw = dpp_from_next(o0[rr]);
            float v[10];
            v[0] = need_l ? (float)lb[rr] : (float)(lw >> 24);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                v[1 + j] = (float)((o0[rr] >> (8 * j)) & 0xffu);
                v[5 + j] = (float)((o1[rr] >> (8 * j)) & 0xffu);
            }
            v[9] = need_r ? (float)rb[rr] : (float)(rw & 0xffu);
#pragma unroll
            for (int j = 0; j < 8; j++) rf[rr][j] = v[j + 1] * k0 + (v[j] + v[j + 2]) * k1;
        }
        if (!in_w) return;
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float b00 = (rf[0][2 * i] + rf[2][2 * i]) * k1 + rf[1][2 * i] * k0;
            const float b01 = (rf[0][2 * i + 1] + rf[2][2 * i + 1]) * k1 + rf[1][2 * i + 1] * k0;
            const float b10 = (rf[1][2 * i] + rf[3][2 * i]) * k1 + rf[2][2 * i] * k0;
            const float b11 = (rf[1][2 * i + 1] + rf[3][2 * i + 1]) * k1 + rf[2][2 * i + 1] * k0;
            float sum = 0.f;
            sum += b00 + b01 + b10 + b11;
            o[i] = sum * 0.25f;
        }
        float* d = dst + (long long)oy * a.ld + ox;
        if (ox + 3 < a.w) *(f32x4*)d = f32x4{o[0], o[1], o[2], o[3]};
        else
            for (int j = 0; j < 4; j++)
                if (ox + j < a.w) d[j] = o[j];
    }
}

// -----------------------------------------------------------------------------------------------------
// tw_pyr_k3f : levels 0 AND 1 of a ½-pyramid from one read of the u8 image (round 5) — tw_pyr_k3<2>'s thread (8 source
//   columns x 4 source rows, edge bytes by DPP wave shifts -> 4 level-1 pixels) also produces the 8 x 2 level-0 pixels of
//   its two centre rows from the same bytes, with level 0's own 3-tap kernel.  12.4 MB instead of 10.4 + 4.1 MB per 1080p
//   image.  Same float operation order per value as tw_pyr_k3<0> / <2>.
// -----------------------------------------------------------------------------------------------------
struct PyrK3fArgs {
    const uint8_t* const* srcs;
    float* dst0;  // level-0 images, z-th at dst0 + z*zs0
    float* dst1;
    long long zs0, zs1;
    long long stride;
    int w0, h0, ld0;   // level 0 = the image's size
    int w1, h1, ld1;   // level 1 = exactly half of it
    float a0, a1;      // level 0's centre / side tap
    float b0, b1;      // level 1's
    int aligned4;
};

__device__ __forceinline__ void tw_pyr_k3f_body(const PyrK3fArgs& a, const int bx_, const int by_, const int bz_)
{
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // tx = lane: a wave is 64 consecutive 4-pixel groups of level 1
    const int ox = (bx_ * 64 + tx) * 4;
    const uint8_t* __restrict__ src = a.srcs[bz_];
    const bool in_w = ox < a.w1;  // lanes past the row end stay active for the lane exchange, they only skip memory
    const int oy = by_ * 4 + ty;
    if (oy >= a.h1) return;  // wave-uniform
    const int sx = 2 * ox;   // first source column of the thread's 8
    const bool own_ok = a.aligned4 && sx + 7 < a.w0;
    const bool left_ok = tx > 0 && a.aligned4 && sx - 1 < a.w0;
    const bool right_ok = tx < 63 && a.aligned4 && sx + 15 < a.w0;
    unsigned o0[4], o1[4], lb[4], rb[4];
    const bool need_l = in_w && !left_ok, need_r = in_w && !right_ok, need_own = in_w && !own_ok;
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
        const int Y = reflect101(2 * oy - 1 + rr, a.h0);
        const uint8_t* __restrict__ S = src + (long long)Y * a.stride;
        o0[rr] = o1[rr] = lb[rr] = rb[rr] = 0;
        if (own_ok) {
            const u32x2 w2 = *(const u32x2*)(S + sx);
            o0[rr] = w2.x;
            o1[rr] = w2.y;
        }
        if (need_l) lb[rr] = S[reflect101(sx - 1, a.w0)];
        if (need_r) rb[rr] = S[reflect101(sx + 8, a.w0)];
        if (need_own) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                o0[rr] |= (unsigned)S[reflect101(sx + j, a.w0)] << (8 * j);
                o1[rr] |= (unsigned)S[reflect101(sx + 4 + j, a.w0)] << (8 * j);
            }
        }
    }
    float rfa[4][8], rfb[4][8];  // row-filtered values with level 0's / level 1's kernel
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
        const unsigned lw = dpp_from_prev(o1[rr]), rw = dpp_from_next(o0[rr]);
        float v[10];
        v[0] = need_l ? (float)lb[rr] : (float)(lw >> 24);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            v[1 + j] = (float)((o0[rr] >> (8 * j)) & 0xffu);
            v[5 + j] = (float)((o1[rr] >> (8 * j)) & 0xffu);
        }
        v[9] = need_r ? (float)rb[rr] : (float)(rw & 0xffu);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            rfa[rr][j] = v[j + 1] * a.a0 + (v[j] + v[j + 2]) * a.a1;
            rfb[rr][j] = v[j + 1] * a.b0 + (v[j] + v[j + 2]) * a.b1;
        }
    }
    if (!in_w) return;
    // level 1: column filter at rows 2oy, 2oy+1 and the 2x2 area (tw_pyr_k3<2>)
    {
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float b00 = (rfb[0][2 * i] + rfb[2][2 * i]) * a.b1 + rfb[1][2 * i] * a.b0;
            const float b01 = (rfb[0][2 * i + 1] + rfb[2][2 * i + 1]) * a.b1 + rfb[1][2 * i + 1] * a.b0;
            const float b10 = (rfb[1][2 * i] + rfb[3][2 * i]) * a.b1 + rfb[2][2 * i] * a.b0;
            const float b11 = (rfb[1][2 * i + 1] + rfb[3][2 * i + 1]) * a.b1 + rfb[2][2 * i + 1] * a.b0;
            float sum = 0.f;
            sum += b00 + b01 + b10 + b11;
            o[i] = sum * 0.25f;
        }
        float* d = a.dst1 + bz_ * a.zs1 + (long long)oy * a.ld1 + ox;
        if (ox + 3 < a.w1) *(f32x4*)d = f32x4{o[0], o[1], o[2], o[3]};
        else
            for (int j = 0; j < 4; j++)
                if (ox + j < a.w1) d[j] = o[j];
    }
    // level 0: rows 2oy, 2oy+1, columns sx .. sx+7 (tw_pyr_k3<0>'s column filter)
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int y0 = 2 * oy + q;
        if (y0 >= a.h0) break;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = (rfa[q][j] + rfa[q + 2][j]) * a.a1 + rfa[q + 1][j] * a.a0;
        float* d = a.dst0 + bz_ * a.zs0 + (long long)y0 * a.ld0 + sx;
        if (sx + 7 < a.w0) {
            *(f32x4*)d = f32x4{o[0], o[1], o[2], o[3]};
            *(f32x4*)(d + 4) = f32x4{o[4], o[5], o[6], o[7]};
        } else {
            for (int j = 0; j < 8; j++)
                if (sx + j < a.w0) d[j] = o[j];
        }
    }
}
__global__ __launch_bounds__(256) void tw_pyr_k3f(PyrK3fArgs a) { tw_pyr_k3f_body(a, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z); }

// =====================================================================================================
// K5  tw_polyexp<N> : FarnebackPolyExp (optflowgf.cpp) — the roofline-graded kernel, 24 B/px algorithmic.
//   Tile = 240 columns x 8 rows per 256-thread workgroup (240 + 2x8 halo columns = 256 = one column per
//   thread for the vertical pass).  Vertical pass (float, 2N+1 taps) runs from a register window of the
//   TH+2N source rows of the thread's column; its three moment rows go to LDS; the horizontal pass (double
//   accumulators, as the CPU code) reads 20-float windows per plane with ds_read_b128 and produces 4 pixels
//   per work item; 5 coefficient planes are stored as 16-byte vectors.
// =====================================================================================================
constexpr int PE_TW = 240, PE_TH = 8, PE_COLS = 256, PE_HALO = 8;

struct PolyCoef {
    float g[8], xg[8], xxg[8];  // index k = 0..N
    double ig11, ig03, ig33, ig55;
    double gd[8], xxgd[8];  // (double)g[k], (double)xxg[k]: f64 operands straight from SGPR pairs (no VALU convert)
};

struct PolyArgs {
    const float* src;  // z-th image at src + z*ps
    float* dst;        // z-th coefficient set (5 planes) at dst + z*5*ps
    int w, h, ld;
    long long ps;  // plane stride (elements)
    PolyCoef c;
};

#ifdef TW_VARIANTS  // A/B kernel (TW_POLY_VARIANT=0): only in `make VARIANTS=1` builds (libtwflow_variants.so)
template <int N>
__global__ __launch_bounds__(256) void tw_polyexp(PolyArgs a)
{
    __shared__ __attribute__((aligned(16))) float sm[3][PE_TH][PE_COLS];
    const int tid = threadIdx.x;
    int bx, by, bz;
    xcd_remap(bx, by, bz);
    const int x0 = bx * PE_TW, y0 = by * PE_TH;
    const float* __restrict__ src = a.src + bz * a.ps;
    float* __restrict__ dst = a.dst + bz * 5 * a.ps;
    const PolyCoef& c = a.c;

    // ---- vertical pass: one column per thread, rows y0-N .. y0+TH-1+N in registers ----
    {
        const int x = clampi(x0 - PE_HALO + tid, 0, a.w - 1);
        float win[PE_TH + 2 * N];
#pragma unroll
        for (int i = 0; i < PE_TH + 2 * N; i++) {
            const int y = clampi(y0 - N + i, 0, a.h - 1);
            win[i] = src[(long long)y * a.ld + x];
        }
#pragma unroll
        for (int r = 0; r < PE_TH; r++) {
            float t0 = win[r + N] * c.g[0], t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int k = 1; k <= N; k++) {
                const float s0 = win[r + N - k], s1 = win[r + N + k];  // rows y-k, y+k
                const float p = s0 + s1;
                t0 = t0 + c.g[k] * p;
                t1 = t1 + c.xg[k] * (s1 - s0);
                t2 = t2 + c.xxg[k] * p;
            }
            sm[0][r][tid] = t0;
            sm[1][r][tid] = t1;
            sm[2][r][tid] = t2;
        }
    }
    __syncthreads();

    // ---- horizontal pass: items = 8 rows x 60 groups of 4 pixels ----
    constexpr int GROUPS = PE_TW / 4;
    for (int it = tid; it < PE_TH * GROUPS; it += 256) {
        const int r = it / GROUPS, q = it - r * GROUPS;
        const int y = y0 + r, x = x0 + 4 * q;
        if (y >= a.h || x >= a.w) continue;
        float w0[20], w1[20], w2[20];
#pragma unroll
        for (int v = 0; v < 5; v++) {
            const f32x4 A = *(const f32x4*)&sm[0][r][4 * q + 4 * v];
            const f32x4 B = *(const f32x4*)&sm[1][r][4 * q + 4 * v];
            const f32x4 C = *(const f32x4*)&sm[2][r][4 * q + 4 * v];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                w0[4 * v + e] = A[e];
                w1[4 * v + e] = B[e];
                w2[4 * v + e] = C[e];
            }
        }
        float o0[4], o1[4], o2[4], o3[4], o4[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int li = PE_HALO + j;
            const float g0 = c.g[0];
            double b1 = (double)(w0[li] * g0), b2 = 0, b3 = (double)(w1[li] * g0), b4 = 0,
                   b5 = (double)(w2[li] * g0), b6 = 0;
#pragma unroll
            for (int k = 1; k <= N; k++) {
                const float gk = c.g[k], xgk = c.xg[k], xxgk = c.xxg[k];
                const double tg = (double)(w0[li + k] + w0[li - k]);
                // tg, gk, xxgk are floats widened to double: the products are exact, fma == mul+add
                b1 = __builtin_fma(tg, (double)gk, b1);
                b4 = __builtin_fma(tg, (double)xxgk, b4);
                b2 += (double)((w0[li + k] - w0[li - k]) * xgk);
                b3 += (double)((w1[li + k] + w1[li - k]) * gk);
                b6 += (double)((w1[li + k] - w1[li - k]) * xgk);
                b5 += (double)((w2[li + k] + w2[li - k]) * gk);
            }
            o1[j] = (float)(b2 * c.ig11);
            o0[j] = (float)(b3 * c.ig11);
            o3[j] = (float)(b1 * c.ig03 + b4 * c.ig33);
            o2[j] = (float)(b1 * c.ig03 + b5 * c.ig33);
            o4[j] = (float)(b6 * c.ig55);
        }
        float* d = dst + (long long)y * a.ld + x;
        if (x + 3 < a.w) {
            *(f32x4*)(d) = f32x4{o0[0], o0[1], o0[2], o0[3]};
            *(f32x4*)(d + a.ps) = f32x4{o1[0], o1[1], o1[2], o1[3]};
            *(f32x4*)(d + 2 * a.ps) = f32x4{o2[0], o2[1], o2[2], o2[3]};
            *(f32x4*)(d + 3 * a.ps) = f32x4{o3[0], o3[1], o3[2], o3[3]};
            *(f32x4*)(d + 4 * a.ps) = f32x4{o4[0], o4[1], o4[2], o4[3]};
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (x + j < a.w) {
                    d[j] = o0[j];
                    d[j + a.ps] = o1[j];
                    d[j + 2 * a.ps] = o2[j];
                    d[j + 3 * a.ps] = o3[j];
                    d[j + 4 * a.ps] = o4[j];
                }
        }
    }
}

#endif  // TW_VARIANTS

// -----------------------------------------------------------------------------------------------------
// tw_polyexp_pk<N> : the same arithmetic with every float add/sub/mul issued as a packed-f32 instruction
//   (v_pk_add_f32 / v_pk_mul_f32 do two lanes' worth of IEEE f32 work in one issue slot; no contraction, so the
//   bits are those of the scalar kernel).  The f64 part (v_cvt_f64_f32, v_add_f64, v_fma_f64) has no packed form
//   and stays as it is.  Pairs are always the two ROWS (2p, 2p+1) of one column:
//     V : one column per lane; the TH+2N window rows are held twice, as even-aligned pairs {w[2j], w[2j+1]} and as
//         the pairs shifted by one row {w[2j+1], w[2j+2]} (each row is loaded into both sets: loads cost no VALU
//         issue, a register copy would), so every tap of an output row pair is one aligned register pair;
//         the three moment rows go to LDS as float2 {row 2p, row 2p+1} per column (ds_write_b64)
//     H : an item = 2 columns x 2 rows; its 18-column window of a plane is 9 ds_read_b128 of aligned float2 pairs;
//         planes are processed one after the other (moment 0 -> b1,b4,b2; moment 1 -> b3,b6; moment 2 -> b5) so
//         that one window and at most 12 double accumulators are live
// -----------------------------------------------------------------------------------------------------
// F32ACC = true is a MEASUREMENT variant only (engine option TW_OPT_POLYEXP_F32, never the default): the horizontal
// accumulators b1..b6 are float instead of the CPU code's double — what OpenCV's own CUDA kernel does.  Not bit-exact;
// exists to put a number on "what would polyexp cost without the f64 half" (profiles/r03_polyexp_f32.md).
// F32ACC = 2 additionally FUSES every multiply-add of both passes (v_pk_fma_f32: what a CUDA build of OpenCV's own GPU
// kernel does by default) — the cheapest arithmetic the algorithm admits, even further from the CPU's bits.
template <bool FUSE>
__device__ __forceinline__ f32x2 pe_mad(f32x2 a, float b, f32x2 c)
{
    if constexpr (FUSE) return __builtin_elementwise_fma(a, f32x2{b, b}, c);
    else return a * b + c;
}
template <int N, int TH, int F32ACC = 0, bool EXT_LDS = false>
__device__ __forceinline__ void tw_polyexp_pk_body(const PolyArgs& a, const VGrid& vg, const unsigned vb, void* ext_lds = nullptr)
{
    constexpr bool FUSE = F32ACC == 2;
    constexpr int RP = TH / 2, NW = TH + 2 * N, NPA = NW / 2, NPB = NW / 2 - 1;
    static_assert(NW % 2 == 0 && PE_TW % 2 == 0 && N <= PE_HALO - 1, "tile shape");
    // EXT_LDS: the 3 x RP x PE_COLS float2 tile lives in the caller's LDS block (a twin launch overlays it with its other body's)
    f32x2 (*sm)[RP][PE_COLS];
    if constexpr (EXT_LDS) {
        sm = (f32x2 (*)[RP][PE_COLS])ext_lds;
    } else {
        __shared__ __attribute__((aligned(16))) f32x2 sm_own[3][RP][PE_COLS];
        sm = sm_own;
    }
    const int tid = threadIdx.x;
    int bx, by, bz;
    xcd_remap_v(vg, vb, bx, by, bz);
    const int x0 = bx * PE_TW, y0 = by * TH;
    const float* __restrict__ src = a.src + bz * a.ps;
    float* __restrict__ dst = a.dst + bz * 5 * a.ps;
    const PolyCoef& c = a.c;

    // ---- V ----
    {
        const unsigned xb = (unsigned)clampi(x0 - PE_HALO + tid, 0, a.w - 1) * 4u;
        unsigned xb2 = xb;
        asm volatile("" : "+v"(xb2));  // opaque copy: the second load of a row is not merged with the first
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(src);
        f32x2 pa[NPA], pb[NPB > 0 ? NPB : 1];
#pragma unroll
        for (int i = 0; i < NW; i++) {
            const unsigned ro = (unsigned)clampi(y0 - N + i, 0, a.h - 1) * ((unsigned)a.ld * 4u);
            pa[i >> 1][i & 1] = bload(rs, xb, ro);
            if (i >= 1 && i <= NW - 2) pb[(i - 1) >> 1][(i - 1) & 1] = bload(rs, xb2, ro);
        }
        // pair of window rows (t, t+1)
        auto P = [&](int t) -> f32x2 { return (t & 1) ? pb[t >> 1] : pa[t >> 1]; };
#pragma unroll
        for (int rp = 0; rp < RP; rp++) {
            const int ci = 2 * rp + N;
            f32x2 t0 = P(ci) * c.g[0], t1 = f32x2{0.f, 0.f}, t2 = f32x2{0.f, 0.f};
#pragma unroll
            for (int k = 1; k <= N; k++) {
                const f32x2 s0 = P(ci - k), s1 = P(ci + k);  // rows y-k, y+k
                const f32x2 p = s0 + s1;
                const f32x2 d = s1 - s0;
                t0 = pe_mad<FUSE>(p, c.g[k], t0);
                t1 = pe_mad<FUSE>(d, c.xg[k], t1);
                t2 = pe_mad<FUSE>(p, c.xxg[k], t2);
            }
            sm[0][rp][tid] = t0;
            sm[1][rp][tid] = t1;
            sm[2][rp][tid] = t2;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();

    // ---- H: items = RP row pairs x 120 column pairs ----
    constexpr int CPAIRS = PE_TW / 2;
    constexpr int WL = 2 + 2 * PE_HALO;  // LDS columns 2cp .. 2cp+17; pixel j of the item sits at window index HALO+j
    for (int it = tid; it < RP * CPAIRS; it += 256) {
        const int rp = it / CPAIRS, cp = it - rp * CPAIRS;
        const int y = y0 + 2 * rp, x = x0 + 2 * cp;
        if (y >= a.h || x >= a.w) continue;
        f32x2 v[WL];
        auto load_window = [&](int pl) {
#pragma unroll
            for (int u = 0; u < WL / 2; u++) {
                const f32x4 A = *(const f32x4*)&sm[pl][rp][2 * cp + 2 * u];
                v[2 * u] = f32x2{A[0], A[1]};
                v[2 * u + 1] = f32x2{A[2], A[3]};
            }
        };
        f32x2 o[5][2];    // [plane][row q] = {pixel 0, pixel 1}
        if constexpr (F32ACC != 0) {
            // all-float horizontal pass, packed over the row pair; same tap order as the double version
            const float ig11 = (float)c.ig11, ig03 = (float)c.ig03, ig33 = (float)c.ig33, ig55 = (float)c.ig55;
            f32x2 p03f[2];
            load_window(0);
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int li = PE_HALO + j;
                f32x2 b1 = v[li] * c.g[0], b2 = f32x2{0.f, 0.f}, b4 = f32x2{0.f, 0.f};
#pragma unroll
                for (int k = 1; k <= N; k++) {
                    const f32x2 tg = v[li + k] + v[li - k];
                    const f32x2 dd = v[li + k] - v[li - k];
                    b1 = pe_mad<FUSE>(tg, c.g[k], b1);
                    b4 = pe_mad<FUSE>(tg, c.xxg[k], b4);
                    b2 = pe_mad<FUSE>(dd, c.xg[k], b2);
                }
                p03f[j] = b1 * ig03;
                const f32x2 r1 = b2 * ig11, r3 = pe_mad<FUSE>(b4, ig33, p03f[j]);
                o[1][0][j] = r1[0]; o[1][1][j] = r1[1];
                o[3][0][j] = r3[0]; o[3][1][j] = r3[1];
            }
            __builtin_amdgcn_sched_barrier(0);
            load_window(1);
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int li = PE_HALO + j;
                f32x2 b3 = v[li] * c.g[0], b6 = f32x2{0.f, 0.f};
#pragma unroll
                for (int k = 1; k <= N; k++) {
                    b3 = pe_mad<FUSE>(v[li + k] + v[li - k], c.g[k], b3);
                    b6 = pe_mad<FUSE>(v[li + k] - v[li - k], c.xg[k], b6);
                }
                const f32x2 r0 = b3 * ig11, r4 = b6 * ig55;
                o[0][0][j] = r0[0]; o[0][1][j] = r0[1];
                o[4][0][j] = r4[0]; o[4][1][j] = r4[1];
            }
            __builtin_amdgcn_sched_barrier(0);
            load_window(2);
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int li = PE_HALO + j;
                f32x2 b5 = v[li] * c.g[0];
#pragma unroll
                for (int k = 1; k <= N; k++) b5 = pe_mad<FUSE>(v[li + k] + v[li - k], c.g[k], b5);
                const f32x2 r2 = pe_mad<FUSE>(b5, ig33, p03f[j]);
                o[2][0][j] = r2[0]; o[2][1][j] = r2[1];
            }
        } else {
        double p03[2][2];  // b1*ig03 of [pixel][row]
        // (the two pixels and the two rows advance in lockstep, one tap at a time: 12 independent f64 chains)
        // moment 0: b1, b4, b2
        __builtin_amdgcn_sched_barrier(0);
        load_window(0);
        {
            double b1[2][2], b2[2][2], b4[2][2];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                f32x2 c0 = v[PE_HALO + j] * c.g[0];
                asm("" : "+v"(c0));  // keep the packed multiply (both halves are only ever read one by one)
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    b1[j][q] = (double)c0[q];
                    b2[j][q] = 0;
                    b4[j][q] = 0;
                }
            }
#pragma unroll
            for (int k = 1; k <= N; k++) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int li = PE_HALO + j;
                    const f32x2 tg = v[li + k] + v[li - k];
                    const f32x2 pd = (v[li + k] - v[li - k]) * c.xg[k];
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const double T = (double)tg[q];
                        // T, gd, xxgd are floats widened to double: the products are exact, fma == mul+add
                        b1[j][q] = __builtin_fma(T, c.gd[k], b1[j][q]);
                        b4[j][q] = __builtin_fma(T, c.xxgd[k], b4[j][q]);
                        b2[j][q] += (double)pd[q];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    o[1][q][j] = (float)(b2[j][q] * c.ig11);
                    p03[j][q] = b1[j][q] * c.ig03;
                    o[3][q][j] = (float)(p03[j][q] + b4[j][q] * c.ig33);
                }
        }
        // moment 1: b3, b6
        __builtin_amdgcn_sched_barrier(0);
        load_window(1);
        {
            double b3[2][2], b6[2][2];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                f32x2 c0 = v[PE_HALO + j] * c.g[0];
                asm("" : "+v"(c0));  // keep the packed multiply (both halves are only ever read one by one)
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    b3[j][q] = (double)c0[q];
                    b6[j][q] = 0;
                }
            }
#pragma unroll
            for (int k = 1; k <= N; k++) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int li = PE_HALO + j;
                    const f32x2 ps = (v[li + k] + v[li - k]) * c.g[k];
                    const f32x2 pd = (v[li + k] - v[li - k]) * c.xg[k];
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        b3[j][q] += (double)ps[q];
                        b6[j][q] += (double)pd[q];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    o[0][q][j] = (float)(b3[j][q] * c.ig11);
                    o[4][q][j] = (float)(b6[j][q] * c.ig55);
                }
        }
        // moment 2: b5
        __builtin_amdgcn_sched_barrier(0);
        load_window(2);
        {
            double b5[2][2];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                f32x2 c0 = v[PE_HALO + j] * c.g[0];
                asm("" : "+v"(c0));  // keep the packed multiply (both halves are only ever read one by one)
#pragma unroll
                for (int q = 0; q < 2; q++) b5[j][q] = (double)c0[q];
            }
#pragma unroll
            for (int k = 1; k <= N; k++) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int li = PE_HALO + j;
                    f32x2 ps = (v[li + k] + v[li - k]) * c.g[k];
                    asm("" : "+v"(ps));  // as above: without this the pair is computed as four scalar instructions
#pragma unroll
                    for (int q = 0; q < 2; q++) b5[j][q] += (double)ps[q];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int q = 0; q < 2; q++) o[2][q][j] = (float)(p03[j][q] + b5[j][q] * c.ig33);
        }
        }  // !F32ACC
        // every result is materialised here: left alone, LLVM sinks the chains of the second row / second pixel into
        // the edge conditionals below (their halves of every packed value then wait in scratch)
#pragma unroll
        for (int pl = 0; pl < 5; pl++)
#pragma unroll
            for (int q = 0; q < 2; q++) asm volatile("" : "+v"(o[pl][q]));
#pragma unroll
        for (int q = 0; q < 2; q++) {
            if (y + q >= a.h) break;
            float* d = dst + (long long)(y + q) * a.ld + x;
            if (x + 1 < a.w) {
#pragma unroll
                for (int pl = 0; pl < 5; pl++) st_stream<2>((f32x2*)(d + pl * a.ps), o[pl][q]);
            } else {
#pragma unroll
                for (int pl = 0; pl < 5; pl++) d[pl * a.ps] = o[pl][q][0];
            }
        }
    }
}
template <int N, int TH, int F32ACC = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void tw_polyexp_pk(PolyArgs a)
{
    tw_polyexp_pk_body<N, TH, F32ACC>(a, hw_grid(), hw_block());
}

// =====================================================================================================
// FarnebackUpdateMatrices for one pixel (optflowgf.cpp): warp R1 by the flow (bilinear gather of 5
// coefficients), combine with R0, border attenuation, form G11,G12,G22,h1,h2.  All float.
// =====================================================================================================
struct UpdTaps {
    float t[5][4];  // the 2x2 neighbourhood of R1 at (x + dx, y + dy), 5 coefficients
    float fx, fy;   // fractional position
    bool inb;
};
// the flow-dependent gather of FarnebackUpdateMatrices (loads only)
__device__ __forceinline__ void update_matrices_gather(const float* __restrict__ R1, long long ps, int ld, int w, int h,
                                                       int x, int y, float dx, float dy, UpdTaps& T)
{
    float fx = (float)x + dx, fy = (float)y + dy;
    const float flx = floorf(fx), fly = floorf(fy);
    // (unsigned)x1 < (unsigned)(w-1) && (unsigned)y1 < (unsigned)(h-1), with cvFloor's INT_MIN for
    // out-of-range / NaN inputs: evaluated on the floats (exact: |.| < 2^24 inside the branch)
    const bool inb = flx >= 0.f && flx < (float)(w - 1) && fly >= 0.f && fly < (float)(h - 1);
    // Branch-free: the 4x5 taps are always gathered (from a clamped, valid position) and discarded when the
    // sample falls outside, so that the loads of several pixels can be in flight together.
    const int x1 = inb ? (int)flx : 0, y1 = inb ? (int)fly : 0;
    T.fx = fx - (float)x1;
    T.fy = fy - (float)y1;
    T.inb = inb;
    // (round 6: raw buffer loads — R1's five planes behind one resource, the plane a scalar byte offset, the lane's position a
    // 32-bit one; the pointer form cost a 64-bit VALU add per plane and row.  The two taps of a row as one 8-byte load.)
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(R1);
    const unsigned o0 = (unsigned)(y1 * ld + x1) * 4u, o1 = o0 + (unsigned)ld * 4u;
#pragma unroll
    for (int c = 0; c < 5; c++) {
        const unsigned so = (unsigned)(c * ps * 4);
        const f32x2 t01 = bload2(rs, o0, so), t23 = bload2(rs, o1, so);
        T.t[c][0] = t01.x;
        T.t[c][1] = t01.y;
        T.t[c][2] = t23.x;
        T.t[c][3] = t23.y;
    }
}
// The same gather with wave-uniform plane bases in SGPR pairs and ONE 32-bit byte offset per lane and pixel (planes are
// below 4 GiB: check_dims): 20 loads share two offset registers instead of carrying a 64-bit address each — what lets
// tw_blur_solve4p keep two pixels of taps in flight beside a 38-row register window.
__device__ __forceinline__ void update_matrices_gather_s(const float* __restrict__ R1, long long ps, int ld, int w, int h,
                                                         int x, int y, float dx, float dy, UpdTaps& T)
{
    float fx = (float)x + dx, fy = (float)y + dy;
    const float flx = floorf(fx), fly = floorf(fy);
    const bool inb = flx >= 0.f && flx < (float)(w - 1) && fly >= 0.f && fly < (float)(h - 1);
    const int x1 = inb ? (int)flx : 0, y1 = inb ? (int)fly : 0;
    T.fx = fx - (float)x1;
    T.fy = fy - (float)y1;
    T.inb = inb;
    const unsigned o0 = (unsigned)(y1 * ld + x1) * 4u, o1 = o0 + (unsigned)ld * 4u;
#pragma unroll
    for (int c = 0; c < 5; c++) {
        const gptr_cf b = sgpr_base(R1 + c * ps);
        T.t[c][0] = gload(b, o0);
        T.t[c][1] = gload(b, o0 + 4u);
        T.t[c][2] = gload(b, o1);
        T.t[c][3] = gload(b, o1 + 4u);
    }
}
// ... and its arithmetic
__device__ __forceinline__ void update_matrices_combine(const float q[5], const UpdTaps& T, int w, int h, int x, int y,
                                                        float dx, float dy, float M[5])
{
    const float fx = T.fx, fy = T.fy;
    const bool inb = T.inb;
    const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
    const float(&t)[5][4] = T.t;
    float r2, r3, r4, r5, r6;
    const float q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4];
    if (inb) {
        r2 = a00 * t[0][0] + a01 * t[0][1] + a10 * t[0][2] + a11 * t[0][3];
        r3 = a00 * t[1][0] + a01 * t[1][1] + a10 * t[1][2] + a11 * t[1][3];
        r4 = a00 * t[2][0] + a01 * t[2][1] + a10 * t[2][2] + a11 * t[2][3];
        r5 = a00 * t[3][0] + a01 * t[3][1] + a10 * t[3][2] + a11 * t[3][3];
        r6 = a00 * t[4][0] + a01 * t[4][1] + a10 * t[4][2] + a11 * t[4][3];
        r4 = (q2 + r4) * 0.5f;
        r5 = (q3 + r5) * 0.5f;
        r6 = (q4 + r6) * 0.25f;
    } else {
        r2 = r3 = 0.f;
        r4 = q2;
        r5 = q3;
        r6 = q4 * 0.5f;
    }
    r2 = (q0 - r2) * 0.5f;
    r3 = (q1 - r3) * 0.5f;
    r2 += r4 * dy + r6 * dx;
    r3 += r6 * dy + r5 * dx;
    constexpr int BORDER = 5;
    if ((unsigned)(x - BORDER) >= (unsigned)(w - BORDER * 2) || (unsigned)(y - BORDER) >= (unsigned)(h - BORDER * 2)) {
        const float border[BORDER] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
        float bx0 = 1.f, bx1 = 1.f, by0 = 1.f, by1 = 1.f;
#pragma unroll
        for (int i = 0; i < BORDER; i++) {
            if (x == i) bx0 = border[i];
            if (w - x - 1 == i) bx1 = border[i];
            if (y == i) by0 = border[i];
            if (h - y - 1 == i) by1 = border[i];
        }
        const float scale = bx0 * bx1 * by0 * by1;
        r2 *= scale;
        r3 *= scale;
        r4 *= scale;
        r5 *= scale;
        r6 *= scale;
    }
    M[0] = r4 * r4 + r6 * r6;
    M[1] = (r4 + r5) * r6;
    M[2] = r5 * r5 + r6 * r6;
    M[3] = r4 * r2 + r6 * r3;
    M[4] = r6 * r2 + r5 * r3;
}
__device__ __forceinline__ void update_matrices_core(const float q[5], const float* __restrict__ R1,
                                                     long long ps, int ld, int w, int h, int x, int y, float dx,
                                                     float dy, float M[5])
{
    UpdTaps T;
    update_matrices_gather(R1, ps, ld, w, h, x, y, dx, dy, T);
    update_matrices_combine(q, T, w, h, x, y, dx, dy, M);
}
__device__ __forceinline__ void update_matrices_px(const float* __restrict__ R0, const float* __restrict__ R1,
                                                   long long ps, int ld, int w, int h, int x, int y, float dx,
                                                   float dy, float M[5])
{
    const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(R0);
    const unsigned o = (unsigned)(y * ld + x) * 4u;
    const float q[5] = {bload(rs0, o, 0u), bload(rs0, o, (unsigned)(ps * 4)), bload(rs0, o, (unsigned)(2 * ps * 4)),
                        bload(rs0, o, (unsigned)(3 * ps * 4)), bload(rs0, o, (unsigned)(4 * ps * 4))};
    update_matrices_core(q, R1, ps, ld, w, h, x, y, dx, dy, M);
}

// =====================================================================================================
// K6  tw_update_matrices<UPSAMPLE, NY> : first FarnebackUpdateMatrices of a level.  UPSAMPLE fuses
//   resize(prevFlow, INTER_LINEAR) + flow *= 1/pyr_scale (and writes the level's initial flow);
//   otherwise the flow planes are read (zero-initialised at the coarsest level).
// =====================================================================================================
struct UpdArgs {
    const float* R;  // pair z: R0 = R + (2z)*5ps, R1 = R + (2z+1)*5ps
    float* flow;     // pair z: 2 planes at flow + z*2*fps
    float* M;        // pair z: 5 planes at M + z*5*ps
    int w, h, ld;
    long long ps, fps;
    // upsample source
    const float* prev;  // pair z: 2 planes at prev + z*2*pfps
    int pw, ph, pld;
    long long pfps;
    const int* xofs;
    const float* alpha;
    const int* yofs;
    const float* beta;
    int xmax;
    float scale;  // (float)(1/pyr_scale)
    int zero_flow;  // coarsest level: flow = 0
    int store_flow; // write the level's initial flow (only needed when no iteration follows: the first
                    // blur+solve overwrites every flow value without reading it)
};

template <bool UPSAMPLE, int NY>
__device__ __forceinline__ void tw_update_matrices_body(const UpdArgs& a, const VGrid& vg, const unsigned vb)
{
    // One lane = NY pixels of one column (rows 4 apart).  Everything that does not depend on the flow (the R0
    // coefficients, the coarse flow taps) is loaded first and without branches, so that a wave has two
    // dependent memory round trips (flow -> gather) instead of three and NY x the loads in flight.
    int bx, by, z;
    xcd_remap_v(vg, vb, bx, by, z);  // an XCD works on a contiguous band of rows: the R1 rows a tile gathers stay in its L2
    const int x = bx * 64 + (threadIdx.x & 63);
    const int yb = by * (4 * NY) + (threadIdx.x >> 6);
    if (x >= a.w || yb >= a.h) return;
    const float* __restrict__ R0 = a.R + (long long)(2 * z) * 5 * a.ps;
    const float* __restrict__ R1 = R0 + 5 * a.ps;
    float* __restrict__ flow = a.flow + (long long)z * 2 * a.fps;
    float* __restrict__ Mo = a.M + (long long)z * 5 * a.ps;
    int yy[NY];
    long long o[NY];
    float q[NY][5], dx[NY], dy[NY];
#pragma unroll
    for (int j = 0; j < NY; j++) {
        yy[j] = min(yb + 4 * j, a.h - 1);  // rows past the bottom repeat the last one (their stores are skipped)
        o[j] = (long long)yy[j] * a.ld + x;
#pragma unroll
        for (int c = 0; c < 5; c++) q[j][c] = ld_stream<1>(R0 + o[j] + c * a.ps);
    }
    if (UPSAMPLE) {
        const int sx = a.xofs[x];
        const int sx1 = min(sx + 1, a.pw - 1);
        const bool two = x < a.xmax;
        const float a0 = a.alpha[2 * x], a1 = a.alpha[2 * x + 1];
        const float* __restrict__ prev = a.prev + (long long)z * 2 * a.pfps;
        float p[NY][8], b0[NY], b1[NY];
#pragma unroll
        for (int j = 0; j < NY; j++) {
            const int sy = a.yofs[yy[j]];
            const int r0 = clampi(sy, 0, a.ph - 1), r1 = clampi(sy + 1, 0, a.ph - 1);
            b0[j] = a.beta[2 * yy[j]];
            b1[j] = a.beta[2 * yy[j] + 1];
            const float* P0 = prev + (long long)r0 * a.pld;
            const float* P1 = prev + (long long)r1 * a.pld;
            p[j][0] = P0[sx];
            p[j][1] = P0[sx1];
            p[j][2] = P1[sx];
            p[j][3] = P1[sx1];
            p[j][4] = P0[a.pfps + sx];
            p[j][5] = P0[a.pfps + sx1];
            p[j][6] = P1[a.pfps + sx];
            p[j][7] = P1[a.pfps + sx1];
        }
#pragma unroll
        for (int j = 0; j < NY; j++) {
            // dx >= xmax: HResizeLinear's single-tap tail (S[sx] * 1)
            const float t0x = two ? p[j][0] * a0 + p[j][1] * a1 : p[j][0] * 1.f;
            const float t1x = two ? p[j][2] * a0 + p[j][3] * a1 : p[j][2] * 1.f;
            const float t0y = two ? p[j][4] * a0 + p[j][5] * a1 : p[j][4] * 1.f;
            const float t1y = two ? p[j][6] * a0 + p[j][7] * a1 : p[j][6] * 1.f;
            dx[j] = (t0x * b0[j] + t1x * b1[j]) * a.scale + 0.f;
            dy[j] = (t0y * b0[j] + t1y * b1[j]) * a.scale + 0.f;
        }
    } else if (a.zero_flow) {
#pragma unroll
        for (int j = 0; j < NY; j++) dx[j] = dy[j] = 0.f;
    } else {
#pragma unroll
        for (int j = 0; j < NY; j++) {
            dx[j] = flow[o[j]];
            dy[j] = flow[o[j] + a.fps];
        }
    }
    float M[NY][5];
    UpdTaps T[NY];
#pragma unroll
    for (int j = 0; j < NY; j++) update_matrices_gather(R1, a.ps, a.ld, a.w, a.h, x, yy[j], dx[j], dy[j], T[j]);
#pragma unroll
    for (int j = 0; j < NY; j++) update_matrices_combine(q[j], T[j], a.w, a.h, x, yy[j], dx[j], dy[j], M[j]);
#pragma unroll
    for (int j = 0; j < NY; j++) {
        if (yb + 4 * j < a.h) {
            if ((UPSAMPLE || a.zero_flow) && a.store_flow) {
                flow[o[j]] = dx[j];
                flow[o[j] + a.fps] = dy[j];
            }
#pragma unroll
            for (int c = 0; c < 5; c++) st_stream<0>(Mo + o[j] + c * a.ps, M[j][c]);
        }
    }
}
template <bool UPSAMPLE, int NY>
__global__ __launch_bounds__(256) void tw_update_matrices(UpdArgs a)
{
    tw_update_matrices_body<UPSAMPLE, NY>(a, hw_grid(), hw_block());
}

// =====================================================================================================
// K7+K8 (+K6)  tw_blur_solve* : FarnebackUpdateFlow_GaussianBlur (optflowgf.cpp).
//   (2*MH+1)-tap window average of the 5 M planes (float, centre-out pair order, replicate borders),
//   2x2 solve in double with the +1e-3 regulariser, and — fused — the FarnebackUpdateMatrices refresh of
//   the same pixel into the other M buffer (the CPU code's stripe-wise refresh is equivalent to
//   "whole new flow from old M, then whole new M": optflowgf.cpp y1 = y - block_size bookkeeping).
//   Tile = (COLS-2*HALO) columns x 8 rows; vertical pass from a register window of 8+2*MH rows per column,
//   blurred rows of all 5 planes staged in LDS, horizontal pass 4 pixels per item from ds_read_b128 windows.
// =====================================================================================================
constexpr int BS_TH = 8;

struct WinCoef {
    float k[33];  // kernel[0..m]
};

struct BlurArgs {
    const float* Min;  // pair z at + z*5*ps
    float* Mout;
    float* flow;     // pair z at + z*2*fps
    const float* R;  // pair z: R0 = R + (2z)*5ps, R1 = R0 + 5ps
    int w, h, ld;
    long long ps, fps;
    int update;  // refresh M (i < iterations-1)
    int xsh;     // tw_blur_solve4: the tile grid starts this many pixels left of the image
    int rot;     // tw_blur_solve4: the S phase's pixel order is rotated by this many pixels within a row (= xsh)
    int store_flow;  // 1: store the flow of a refreshing launch too (nothing reads it: the refresh uses the value in
                     // registers and the next launch overwrites it; the engine stores only the last iteration's)
    int m;       // runtime m for the generic kernel
    int nomask;  // A/B switch (TW_BLUR_NOMASK=1): compute the lanes that overhang the image as well
    int cm;      // tw_blur_solve4y: column-major tile order inside an XCD's band (xcd_remap_cm)
    // tw_blur_solve4, single-pair schedule: the launch that stores the level-0 flow also stores its span-grid samples (what
    // tw_span_gather would read back): grid = pair z's gw x gh float2 at + z*gw*gh, gspan = span, gmagic = ceil(2^32 / span)
    // (exact quotients for coordinates < 65 536)
    float2* grid;
    int gspan, gw, gh;
    unsigned gmagic;
    WinCoef c;
};

// -----------------------------------------------------------------------------------------------------
// tw_blur_solve4<MH,COLS,HALO,TH,FUSED> : K7, the window average of the 5 M planes + 2x2 solve (+ the fused
//   FarnebackUpdateMatrices refresh for the next iteration).  A workgroup owns a (COLS-2*HALO) x TH tile.
//   V : one column per lane; the TH+2*MH clamped row offsets are computed once (SGPRs) and the plane is selected
//       by rebasing the buffer resource, so the vertical phase issues no address arithmetic at all; the next
//       plane's register window is loaded while the current plane is being blurred (two windows)
//   H : 4 pixels per item from LDS windows; the results of all 5 planes stay in registers (3 workgroup barriers),
//       then go back to the tile interiors
//   S : lane-consecutive pixels: solve in double, store the flow, refresh M (R0 fetched before H, two R1
//       gathers in flight per lane)
// -----------------------------------------------------------------------------------------------------
// (the body over a virtual grid and a caller-owned LDS tile, so that a twin launch can run it beside another body and overlay the two
//  bodies' LDS: tw_twin_s4_poly below; the plain kernel hands it the hardware grid and its own tile — same values, same code)
template <int MH, int COLS, int HALO, int TH, bool FUSED, int VILP = 2, int HILP = 2, int SUNROLL = 2, bool QPRE = true, bool VPRE = true>
__device__ __forceinline__ void tw_blur_solve4_body(const BlurArgs& a, const VGrid& vg, const unsigned vb, float (*sm)[TH][COLS])
{
    constexpr int TW = COLS - 2 * HALO;
    constexpr int NW = TH + 2 * MH;
    const int tid = threadIdx.x;
    int bx, by, z;
    xcd_remap_v(vg, vb, bx, by, z);
    // the tile grid starts XSH pixels left of the image so that a wave's 256-byte row segment (which begins HALO
    // pixels left of its tile) is 128-byte aligned: two cache lines per load instead of three
    const int x0 = bx * TW - a.xsh, y0 = by * TH;
    const WinCoef& c = a.c;
    const float* __restrict__ Min = a.Min + (long long)z * 5 * a.ps;
    float* __restrict__ flow = a.flow + (long long)z * 2 * a.fps;
    // A tile that overhangs the image by 32 columns or more (the last tile column: 144 of 224 columns at 1920, 64 of
    // 224 at 960) is PARTIAL: its horizontal items and its solve / refresh pixels are re-indexed over the valid
    // columns [cx0, cx0 + vw) only, so that whole waves drop out instead of lanes (wave-uniform switch; the index
    // arithmetic of a full tile keeps its compile-time divisors).
    const int cx0 = max(0, -x0), vw = min(TW, a.w - x0) - cx0;
    const bool partial = !a.nomask && vw <= TW - 32;
    const int gq0 = cx0 >> 2, gv = ((cx0 + vw + 3) >> 2) - gq0;  // 4-pixel groups with a valid pixel
    const float inv_gv = 1.f / (float)gv, inv_vw = 1.f / (float)vw;

    // ---- V ----
    // Columns further than MH outside the image feed no valid output pixel (the tile grid overhangs the right edge, and
    // the left one by xsh): their lanes are masked off — no loads, no arithmetic, no LDS stores.  The engine runs at
    // the board's power limit, so an idle lane is a saving even where it does not shorten the instruction stream.
    if (a.nomask || (x0 - HALO + tid >= -MH && x0 - HALO + tid <= a.w - 1 + MH)) {
        const unsigned xb = (unsigned)clampi(x0 - HALO + tid, 0, a.w - 1) * 4u;
        unsigned ro[NW];  // wave-uniform byte offsets of the clamped rows
#pragma unroll
        for (int i = 0; i < NW; i++) ro[i] = (unsigned)clampi(y0 - MH + i, 0, a.h - 1) * ((unsigned)a.ld * 4u);
        float wa[NW], wb[VPRE ? NW : 1];
        {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(Min);
#pragma unroll
            for (int i = 0; i < NW; i++) wa[i] = bload(rs, xb, ro[i]);
        }
#pragma unroll
        for (int ch = 0; ch < 5; ch++) {
            // VPRE: two windows, the next plane loads while this one is blurred; otherwise one window (wide
            // kernels: 2*MH+TH rows would not fit twice) and the other workgroups of the CU cover the load
            float* cur = (VPRE && (ch & 1)) ? wb : wa;
            float* nxt = (VPRE && (ch & 1)) ? wa : wb;
            if (VPRE && ch < 4) {
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(Min + (long long)(ch + 1) * a.ps);
#pragma unroll
                for (int i = 0; i < NW; i++) nxt[i] = bload(rs, xb, ro[i]);
            }
            if (!VPRE && ch > 0) {
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(Min + (long long)ch * a.ps);
#pragma unroll
                for (int i = 0; i < NW; i++) wa[i] = bload(rs, xb, ro[i]);
            }
#pragma unroll
            for (int r = 0; r < TH; r++) {
                float s0 = cur[r + MH] * c.k[0];
#pragma unroll
                for (int i = 1; i <= MH; i++) s0 += (cur[r + MH + i] + cur[r + MH - i]) * c.k[i];
                sm[ch][r][tid] = s0;
                if ((r & (VILP - 1)) == VILP - 1) __builtin_amdgcn_sched_barrier(0);  // VILP rows in flight
            }
        }
    }
    __syncthreads();

    // ---- H: all planes, results in registers ----
    constexpr int GROUPS = TW / 4;
    constexpr int NITEM = TH * GROUPS;
    constexpr int ROUNDS = (NITEM + COLS - 1) / COLS;
    constexpr int WL = 4 + 2 * HALO;
    constexpr int NPX = (TH * TW + COLS - 1) / COLS;  // the last round may be ragged (COLS = 512: 7.5 pixels per lane)
    constexpr bool RAGGED = (TH * TW) % COLS != 0;
    // QPRE: the R0 coefficients of the lane's S-phase pixels do not depend on the flow — fetch them now, so that a
    // third of the refresh's loads move under the horizontal arithmetic (+1 % pairs/s; 110 VGPRs, still 4 waves/SIMD)
    float qpre[QPRE ? NPX : 1][5];
    if constexpr (QPRE && FUSED) {
        if (a.update) {
            const float* __restrict__ R0p = a.R + (long long)(2 * z) * 5 * a.ps;
#pragma unroll
            for (int i = 0; i < NPX; i++) {
                int r, cx;
                if (partial) {
                    const int p = min(tid + i * COLS, TH * vw - 1);
                    r = (int)(((float)p + 0.5f) * inv_vw);
                    cx = cx0 + p - r * vw;
                } else {
                    const int p = RAGGED ? min(tid + i * COLS, TH * TW - 1) : tid + i * COLS;
                    r = p / TW;
                    const int c0 = p - r * TW + a.rot;
                    cx = c0 >= TW ? c0 - TW : c0;
                }
                const int xc = clampi(x0 + cx, 0, a.w - 1), yc = min(y0 + r, a.h - 1);
                // (round 6: 32-bit offsets behind a buffer resource unless the cache-policy A/B build wants its nontemporal loads)
                if constexpr (((TW_NT >> 1) & 1) == 0) {
                    const __amdgpu_buffer_rsrc_t rsR0 = make_rsrc(R0p);
                    const unsigned o = (unsigned)(yc * a.ld + xc) * 4u;
#pragma unroll
                    for (int cc = 0; cc < 5; cc++) qpre[i][cc] = bload(rsR0, o, (unsigned)(cc * a.ps * 4));
                } else {
                const long long o = (long long)yc * a.ld + xc;
#pragma unroll
                for (int cc = 0; cc < 5; cc++) qpre[i][cc] = ld_stream<1>(R0p + o + cc * a.ps);
                }
            }
        }
    }
    f32x4 res[ROUNDS][5];
#pragma unroll
    for (int rd = 0; rd < ROUNDS; rd++) {
        const int it = tid + rd * COLS;
        // (items wholly outside the image produce nothing that is stored: skipped, see V)
        int r, q;
        bool on;
        if (partial) {
            on = it < TH * gv;
            r = (int)(((float)it + 0.5f) * inv_gv);
            q = gq0 + it - r * gv;
        } else {
            r = it / GROUPS;
            q = it - r * GROUPS;
            on = it < NITEM && (a.nomask || (x0 + 4 * q < a.w && x0 + 4 * q + 3 >= 0));
        }
        if (on) {
#pragma unroll
            for (int ch = 0; ch < 5; ch++) {
                float v[WL];
#pragma unroll
                for (int u = 0; u < WL / 4; u++) {
                    const f32x4 A = *(const f32x4*)&sm[ch][r][4 * q + 4 * u];
                    v[4 * u] = A[0];
                    v[4 * u + 1] = A[1];
                    v[4 * u + 2] = A[2];
                    v[4 * u + 3] = A[3];
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int li = HALO + j;
                    float sum = v[li] * c.k[0];
#pragma unroll
                    for (int i = 1; i <= MH; i++) sum += c.k[i] * (v[li - i] + v[li + i]);
                    res[rd][ch][j] = sum;
                    if ((j & (HILP - 1)) == HILP - 1) __builtin_amdgcn_sched_barrier(0);  // HILP pixels in flight
                }
            }
        }
    }
    __syncthreads();  // every window has been read: the interiors may be overwritten
#pragma unroll
    for (int rd = 0; rd < ROUNDS; rd++) {
        const int it = tid + rd * COLS;
        int r, q;
        bool on;
        if (partial) {
            on = it < TH * gv;
            r = (int)(((float)it + 0.5f) * inv_gv);
            q = gq0 + it - r * gv;
        } else {
            r = it / GROUPS;
            q = it - r * GROUPS;
            on = it < NITEM && (a.nomask || (x0 + 4 * q < a.w && x0 + 4 * q + 3 >= 0));
        }
        if (on) {
#pragma unroll
            for (int ch = 0; ch < 5; ch++) *(f32x4*)&sm[ch][r][HALO + 4 * q] = res[rd][ch];
        }
    }
    __syncthreads();

    // ---- S: solve (+ refresh), lane-consecutive pixels ----
    float* __restrict__ Mout = a.Mout + (long long)z * 5 * a.ps;
    const float* __restrict__ R0 = a.R + (long long)(2 * z) * 5 * a.ps;
    const float* __restrict__ R1 = R0 + 5 * a.ps;
    const __amdgpu_buffer_rsrc_t rsF = make_rsrc(flow), rsM = make_rsrc(Mout);
    // Every lane always computes (on a clamped, valid pixel); only the stores are predicated, so the loop body
    // has no control flow and the gathers of SUNROLL pixels are in flight together.
#pragma unroll
    for (int i = 0; i < NPX; i++) {
        if (i % SUNROLL == 0) __builtin_amdgcn_sched_barrier(0);  // SUNROLL pixels in flight
        bool mine;
        int r, cx;
        if (partial) {
            if (i * COLS >= TH * vw) break;  // wave-uniform: the valid pixels of a partial tile take fewer rounds
            mine = tid + i * COLS < TH * vw;
            const int p = mine ? tid + i * COLS : TH * vw - 1;
            r = (int)(((float)p + 0.5f) * inv_vw);
            cx = cx0 + p - r * vw;
        } else {
            mine = !RAGGED || tid + i * COLS < TH * TW;
            const int p = mine ? tid + i * COLS : TH * TW - 1;
            // rotated by xsh within the row: the 64 consecutive pixels of a wave then start on a 128-byte boundary,
            // like the vertical phase's row segments (the tile itself starts xsh pixels left of one)
            r = p / TW;
            const int c0 = p - r * TW + a.rot;
            cx = c0 >= TW ? c0 - TW : c0;
        }
        const int x = x0 + cx, y = y0 + r;
        const bool valid = mine && x >= 0 && x < a.w && y < a.h;
        const int xc = clampi(x, 0, a.w - 1), yc = min(y, a.h - 1);
        const double g11 = sm[0][r][HALO + cx], g12 = sm[1][r][HALO + cx], g22 = sm[2][r][HALO + cx],
                     h1 = sm[3][r][HALO + cx], h2 = sm[4][r][HALO + cx];
        const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
        const float fxv = (float)((g11 * h2 - g12 * h1) * idet);
        const float fyv = (float)((g22 * h1 - g12 * h2) * idet);
        const long long o = (long long)yc * a.ld + xc;
        const unsigned ob = (unsigned)(yc * a.ld + xc) * 4u;
        if (valid && (!a.update || a.store_flow)) {
            bstore(rsF, ob, 0u, fxv);
            bstore(rsF, ob, (unsigned)(a.fps * 4), fyv);
            if (a.gspan > 0) {  // wave-uniform
                const unsigned qx = __umulhi((unsigned)xc, a.gmagic), qy = __umulhi((unsigned)yc, a.gmagic);
                if (qx * (unsigned)a.gspan == (unsigned)xc && qy * (unsigned)a.gspan == (unsigned)yc)
                    a.grid[((long long)z * a.gh + qy) * a.gw + qx] = float2{fxv, fyv};
            }
        }
        if (FUSED) {
            if (a.update) {  // wave-uniform
                float M[5];
                if constexpr (QPRE) update_matrices_core(qpre[i], R1, a.ps, a.ld, a.w, a.h, xc, yc, fxv, fyv, M);
                else update_matrices_px(R0, R1, a.ps, a.ld, a.w, a.h, xc, yc, fxv, fyv, M);
                if (valid) {
                    if constexpr ((TW_NT & 1) == 0) {
#pragma unroll
                        for (int cc = 0; cc < 5; cc++) bstore(rsM, ob, (unsigned)(cc * a.ps * 4), M[cc]);
                    } else {
#pragma unroll
                        for (int cc = 0; cc < 5; cc++) st_stream<0>(Mout + o + cc * a.ps, M[cc]);
                    }
                }
            }
        }
    }
}
template <int MH, int COLS, int HALO, int TH, bool FUSED, int VILP = 2, int HILP = 2, int SUNROLL = 2, bool QPRE = true, bool VPRE = true>
__global__ __launch_bounds__(COLS) __attribute__((amdgpu_waves_per_eu(4, 8))) void tw_blur_solve4(BlurArgs a)
{
    __shared__ __attribute__((aligned(16))) float sm[5][TH][COLS];
    tw_blur_solve4_body<MH, COLS, HALO, TH, FUSED, VILP, HILP, SUNROLL, QPRE, VPRE>(a, hw_grid(), hw_block(), sm);
}

// -----------------------------------------------------------------------------------------------------
// tw_flow_iter<MH, UPS> (round 5) : ONE WHOLE ITERATION of the flow update without M in HBM.
//   The reference's iteration is  M = FarnebackUpdateMatrices(R0, R1, flow)  ->  window average of M  ->  2x2 solve -> flow'
//   (optflowgf.cpp FarnebackUpdateFlow_GaussianBlur; the stripe-wise refresh is "whole new M from the whole new flow").
//   tw_blur_solve4 moves M across HBM twice per iteration (20 B/px in, 20 out, + R0 20 + R1 20 = 80 B/px).  M is a pure
//   function of R0(x), R1 around x + flow(x) and flow(x): this kernel reads the PREVIOUS FLOW (8 B/px) instead, recomputes
//   every M row exactly once and keeps the 2*MH + TH rows the window needs in an LDS ring — 56 B/px per iteration (flow 8 +
//   R0 20 + R1 20 in, flow' 8 out), no separate first-update launch per level (profiles/r04_m_free_iteration.md: the
//   memory shape replays at 29.9 us per 1080p pair against 42.1).
//   A 1024-thread workgroup (one per CU: 150 KB of LDS) marches down a strip of SC = 190 columns (OUT = 160 outputs + MH
//   halo columns either side) in steps of TH = 5 rows:
//     blk[8][5][TH][.]        eight equal LDS blocks; chunk k (TH = 5 rows of M, 5 planes) lives in block k % 8, the window of a
//                             step is 7 chunks (rows y0 - MH .. y0 + TH + MH - 1), the eighth block is the spare
//     phase A (15 waves)      V: thread (plane, column) reads its 35-row window (static LDS offsets: the block rotation is an
//                             8-way wave-uniform switch) and writes the TH vertical sums to the spare block X — float,
//                             centre-out pair order, exactly tw_blur_solve4's V.
//     barrier
//     phase B1 (1 000 thr.)   H: thread (plane, row, 4-pixel group) reads a 36-value window of X (ds_read_b128), horizontal
//                             sums in tw_blur_solve4's order, into block Y = the block of the chunk this step retires.
//     barrier
//     phase B2                C (950 threads): FarnebackUpdateMatrices of the chunk the next step needs (update_matrices_combine:
//                             the same function the other kernels use) into X, from loads issued a whole step earlier — and
//                             at once the loads of the chunk after it (R0, the 2x2 taps of R1 at x + flow, the flow one chunk
//                             further on) into the registers just freed: the HBM streams all the time, nothing waits for it;
//                             S (800 threads): one pixel from Y, 2x2 solve in double, flow store.
//     barrier
//   (A first version did H + S as 400 (row, 2-pixel, all planes) items next to C in one phase: two barriers per step, but
//   only two busy waves per SIMD for most of that phase — 44 % VALU utilisation, 49.8 us per 1080p pair; gpurun_out/r5d.)
//   Borders need no special case: M at a replicated row / column IS M at the clamped coordinate (the vertical sum of a
//   replicated column is the sum of the clamped column), so phase C clamps (x, y) and everything downstream is uniform.
//   UPS: the iteration's input flow is the bilinear upsample * (1/pyr_scale) of the coarser level's flow (first iteration of
//   a level: what tw_update_matrices<true> fuses), computed in place of the flow load; zero_flow: the coarsest level.
// -----------------------------------------------------------------------------------------------------
constexpr int FI_TH = 5, FI_SC = 190, FI_PITCH = 192;
// NT = 512 (TW_FI_NT=512, A/B): the same kernel as TWO 512-thread workgroups per CU on strips of 94 columns (64 outputs), 76.8 KB of
// LDS each — a workgroup's barrier phases then overlap with the other workgroup's, for 1.47 x instead of 1.19 x halo columns
constexpr int FI_SC_512 = 94, FI_PITCH_512 = 96;
struct FlowIterArgs {
    const float* R;        // pair z: R0 = R + (2z)*5ps, R1 = R0 + 5ps
    const float* flow_in;  // pair z: 2 planes at flow_in + z*2*fps_in
    float* flow_out;       // pair z: 2 planes at flow_out + z*2*fps_out
    int w, h, ld;
    long long ps, fps_in, fps_out;
    int nt;         // steps (of FI_TH rows) per workgroup; grid.y segments cover the rows
    int zero_flow;  // input flow = 0 (coarsest level), nothing is read
    // UPS: upsample source (the coarser level's final flow) and its resize tables, as UpdArgs
    const float* prev;
    int pw, ph, pld;
    long long pfps;
    const int* xofs;
    const float* alpha;
    const int* yofs;
    const float* beta;
    int xmax;
    float scale;
    unsigned long long* dbg;  // measurement builds (TW_VARIANTS): s_memtime stamps of steps 40-47 of the first 32 workgroups
    int dbg_skip;  // measurement builds (TW_VARIANTS, TW_FI_SKIP): 1 no C (combine), 4 no H, 8 no S — timing only
    WinCoef c;
};

#ifdef TW_VARIANTS
#define TW_FI_SKIP(bit) (a.dbg_skip & (bit))
// TW_FI_SKIP=16: no workgroup barrier in the step loop (wrong results; what the three barriers of a step cost: 1.5 %)
#define TW_FI_SYNC() do { if (!TW_FI_SKIP(16)) __syncthreads(); } while (0)
// stamp i (0..7) of step st, waves 0 and 9, workgroups 0..31 (linear block id), steps 40..47
#define TW_FI_STAMP(i)                                                                                                   \
    do {                                                                                                                 \
        if (a.dbg && (tid == 0 || tid == NT / 2 + 64) && fi_blin < 32 && st >= 40 && st < 48)                              \
            a.dbg[(((size_t)fi_blin * 2 + (tid ? 1 : 0)) * 8 + (st - 40)) * 8 + (i)] = __builtin_amdgcn_s_memtime();       \
    } while (0)
#else
#define TW_FI_SKIP(bit) false
#define TW_FI_SYNC() __syncthreads()
#define TW_FI_STAMP(i) do { } while (0)
#endif
// MODE 0: input flow from flow_in; 1: the coarser level's flow, upsampled in place of the load (UPS); 2: zero flow (no
// load at all).  A template parameter, not a runtime flag: with the three sources behind one join the compiler copied the
// freshly loaded flow into the loop-carried registers AFTER the join — an s_waitcnt vmcnt(0) on loads issued a few
// instructions earlier, in every step.
template <int MH, int MODE, int NT = 1024>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void tw_flow_iter(FlowIterArgs a)
{
    constexpr bool UPS = MODE == 1;
    // DEEP (round 6; TW_FI_DEEP, default on for MODE 0 / 2): the 2x2 taps of R1 are consumed ONE STEP LATER than they are
    // issued — two tap register sets that swap roles like the two R0 sets did.  A chunk's taps then have a whole step
    // (~9 000 cycles) to arrive and may be issued in ANY phase, phase B2 included; before, all of them had to go out in phases
    // A and H of the step whose B2 combines them (195 of a step's 255 wave-loads in phase A alone).  R0 is loaded in the step
    // that uses it.  The upsampling first iteration (MODE 1) carries eight coarse taps per pixel as well and keeps the old
    // schedule: a second tap set does not fit its registers.
#ifndef TW_FI_DEEP
#define TW_FI_DEEP 1
#endif
    constexpr bool DEEP = TW_FI_DEEP && MODE != 1 && NT == 1024;
    // where the five tap planes of the next chunk go out (DEEP): phase 0 = behind V's row `slot`, 1 = behind H's pixel `slot`,
    // 2 = phase B2 before the combine.  TW_FI_DIST picks a preset (A/B builds).
#ifndef TW_FI_DIST
#define TW_FI_DIST 0
#endif
    // ten tap loads (plane, half) -> phase * 8 + slot: phase 0 behind V's row `slot`, 1 behind H's pixel `slot`, 2 phase B2 (slot
    // 0 before the combine, 1 behind it).  Measured (gpurun_out/r6i, level 0, 64 pairs per launch, us per pair; the schedule
    // without DEEP 40.85 / 41.82 on the two leases): preset 0 40.15 / 41.12;  1 (one load per slot, every phase) 41.13;  all ten
    // in phase B2 44.8;  all ten in phase A 42.9;  four more spreadings within +-0.2 % of preset 0.  A burst of loads in one phase
    // blocks its waves at the issue; beyond avoiding that the placement does not matter.
#define TW_FI_P(ph, sl) ((ph) * 8 + (sl))
    constexpr int DIST[4][10] = {
        // plane 0 (row y1, row y1 + 1), plane 1, plane 2, plane 3, plane 4
        {TW_FI_P(0, 0), TW_FI_P(0, 0), TW_FI_P(1, 0), TW_FI_P(1, 0), TW_FI_P(1, 1), TW_FI_P(1, 1), TW_FI_P(2, 0), TW_FI_P(2, 0), TW_FI_P(2, 0), TW_FI_P(2, 0)},
        {TW_FI_P(0, 1), TW_FI_P(0, 3), TW_FI_P(1, 0), TW_FI_P(1, 1), TW_FI_P(1, 2), TW_FI_P(1, 3), TW_FI_P(2, 0), TW_FI_P(2, 0), TW_FI_P(2, 1), TW_FI_P(2, 1)},
        {TW_FI_P(2, 0), TW_FI_P(2, 0), TW_FI_P(2, 0), TW_FI_P(2, 0), TW_FI_P(2, 0), TW_FI_P(2, 0), TW_FI_P(2, 0), TW_FI_P(2, 0), TW_FI_P(2, 0), TW_FI_P(2, 0)},
        {TW_FI_P(0, 0), TW_FI_P(0, 0), TW_FI_P(0, 1), TW_FI_P(0, 1), TW_FI_P(0, 2), TW_FI_P(0, 2), TW_FI_P(0, 3), TW_FI_P(0, 3), TW_FI_P(0, 4), TW_FI_P(0, 4)},
    };
    constexpr int TH = FI_TH, SC = NT == 1024 ? FI_SC : FI_SC_512, P = NT == 1024 ? FI_PITCH : FI_PITCH_512, OUT = SC - 2 * MH,
                  RING = TH + 2 * MH, NCH = RING / TH, NB = NCH + 1;
    static_assert(RING % TH == 0 && SC <= P && TH * SC <= NT && TH == 5 && OUT % 4 == 0 && 5 * TH * (OUT / 4) <= NT && (NB & (NB - 1)) == 0, "geometry");
    // waves with a column in V / a pixel in C; a last wave without one (NT = 1024: wave 15) runs a loop of its own below
    constexpr int VWAVES = (TH * SC + 63) / 64;
    constexpr bool SPLIT = VWAVES * 64 < NT;
    // NB = 8 equal blocks of [plane][row][column]: chunk k of M lives in block k % 8; the eighth block is the spare that
    // takes the vertical sums of a step, and the block of the chunk a step retires takes its horizontal sums:
    //   step st:  V reads chunks st .. st+6 (blocks (st+j) % 8) and writes block X = (st+7) % 8;  H reads X and writes
    //   block Y = st % 8 (chunk st is dead);  S reads Y while C writes chunk st+7 into X — no two phases touch one block.
    __shared__ __attribute__((aligned(16))) float blk[NB][5][TH][P];
    const int tid = threadIdx.x;
    int bx, seg, z;
    xcd_remap(bx, seg, z);
    const unsigned fi_blin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    (void)fi_blin;
    const WinCoef& c = a.c;
    const float* __restrict__ R0 = a.R + (long long)(2 * z) * 5 * a.ps;
    const float* __restrict__ R1 = R0 + 5 * a.ps;
    const float* __restrict__ fin = a.flow_in + (long long)z * 2 * a.fps_in;
    float* __restrict__ fout = a.flow_out + (long long)z * 2 * a.fps_out;
    // thread = (row cr of a chunk, strip column cc) in phase C and = (plane cr, column cc) in phase A (TH == 5 planes)
    const bool cth = tid < TH * SC;
    const int cr = cth ? tid / SC : 0, cc = cth ? tid - cr * SC : 0;
    const int xc = clampi(bx * OUT - MH + cc, 0, a.w - 1);
    const int ys = seg * a.nt * TH;
    const int nsteps = min(a.nt, (a.h - ys + TH - 1) / TH);
    if (nsteps <= 0) return;

    // ---- the input flow of the pixel (xc, clamp(y)) of chunk k, row cr ----
    // UPS: resize(prevFlow, INTER_LINEAR) * scale as tw_update_matrices<true> computes it; the column's taps are fixed
    int sx = 0, sx1 = 0;
    bool two = false;
    float ua0 = 0.f, ua1 = 0.f;
    if (UPS) {
        sx = a.xofs[xc];
        sx1 = min(sx + 1, a.pw - 1);
        two = xc < a.xmax;
        ua0 = a.alpha[2 * xc];
        ua1 = a.alpha[2 * xc + 1];
    }
    const float* __restrict__ prev = UPS ? a.prev + (long long)z * 2 * a.pfps : nullptr;
    struct FlowIn {
        float p[UPS ? 8 : 2];
        float b0, b1;
    };
    auto row_of = [&](int k) { return clampi(ys - MH + TH * k + cr, 0, a.h - 1); };
    // UPS: the row's resize entry (yofs, beta) is a table lookup the coarse taps' addresses depend on: it is fetched one
    // chunk ahead of the taps (ysy / yb0 / yb1 hold the entry of the NEXT flow_issue's row), or every step would wait
    // for an L2 round trip behind the R0 loads it has just issued
    int ysy = 0;
    float yb0 = 0.f, yb1 = 0.f;
    auto ytab_issue = [&](int k) {
        if constexpr (UPS) {
            const int yc = row_of(k);
            ysy = a.yofs[yc];
            yb0 = a.beta[2 * yc];
            yb1 = a.beta[2 * yc + 1];
        }
    };
    const __amdgpu_buffer_rsrc_t rsp = make_rsrc(UPS ? (const void*)prev : (const void*)R0);  // UPS: the coarser level's flow
    const unsigned prev_p1 = UPS ? (unsigned)(a.pfps * 4) : 0u, sxb = (unsigned)sx * 4u, sx1b = (unsigned)sx1 * 4u;
    const __amdgpu_buffer_rsrc_t rsf = make_rsrc(MODE == 0 ? (const void*)fin : (const void*)R0);  // the input flow's two planes
    const unsigned fin_p1 = (unsigned)(a.fps_in * 4);
    auto flow_issue = [&](int k, FlowIn& f) {
        const int yc = row_of(k);
        if constexpr (UPS) {
            int sy;
            asm volatile("v_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %5"
                         : "=&v"(sy), "=&v"(f.b0), "=&v"(f.b1)
                         : "v"(ysy), "v"(yb0), "v"(yb1));
            const int r0 = clampi(sy, 0, a.ph - 1), r1 = clampi(sy + 1, 0, a.ph - 1);
            const unsigned b0 = (unsigned)(r0 * a.pld) * 4u, b1 = (unsigned)(r1 * a.pld) * 4u;
            f.p[0] = bload(rsp, b0 + sxb, 0u);
            f.p[1] = bload(rsp, b0 + sx1b, 0u);
            f.p[2] = bload(rsp, b1 + sxb, 0u);
            f.p[3] = bload(rsp, b1 + sx1b, 0u);
            f.p[4] = bload(rsp, b0 + sxb, prev_p1);
            f.p[5] = bload(rsp, b0 + sx1b, prev_p1);
            f.p[6] = bload(rsp, b1 + sxb, prev_p1);
            f.p[7] = bload(rsp, b1 + sx1b, prev_p1);
        } else if constexpr (MODE == 2) {
            f.p[0] = f.p[1] = 0.f;
        } else {
            const unsigned o = (unsigned)(yc * a.ld + xc) * 4u;  // (planes are below 4 GiB: check_dims)
            f.p[0] = bload(rsf, o, 0u);
            f.p[1] = bload(rsf, o, fin_p1);
        }
    };
    auto flow_value = [&](const FlowIn& f, float& dx, float& dy) {
        if constexpr (UPS) {
            // dx >= xmax: HResizeLinear's single-tap tail (S[sx] * 1)
            const float t0x = two ? f.p[0] * ua0 + f.p[1] * ua1 : f.p[0] * 1.f;
            const float t1x = two ? f.p[2] * ua0 + f.p[3] * ua1 : f.p[2] * 1.f;
            const float t0y = two ? f.p[4] * ua0 + f.p[5] * ua1 : f.p[4] * 1.f;
            const float t1y = two ? f.p[6] * ua0 + f.p[7] * ua1 : f.p[6] * 1.f;
            dx = (t0x * f.b0 + t1x * f.b1) * a.scale + 0.f;
            dy = (t0y * f.b0 + t1y * f.b1) * a.scale + 0.f;
        } else {
            // explicit moves, here: left to the register allocator the copies land AFTER the next flow load has been
            // issued, the load then needs registers of its own and its result is moved into the loop-carried pair behind
            // an s_waitcnt that stalls every wave on a load it issued a few instructions earlier
            asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(dx), "=&v"(dy) : "v"(f.p[0]), "v"(f.p[1]));
        }
    };
    // R0 and the 2x2 taps of R1 at (xc, yc) + flow: loads only (update_matrices_gather_s with the ten plane bases pinned
    // to SGPR pairs once, here, in uniform control flow: 27 loads share three 32-bit offset registers)
    gptr_cf r0b[5], r1b[5];
#pragma unroll
    for (int ch = 0; ch < 5; ch++) {
        r0b[ch] = sgpr_base(R0 + ch * a.ps);
        r1b[ch] = sgpr_base(R1 + ch * a.ps);
    }
    // In pieces, so that the steady loop can spread them over a step (see there): the addresses + R0, then one plane of taps each
    unsigned go0 = 0, go1 = 0;
    unsigned gq = 0;
    auto gather_r0 = [&](int ch, float q[5]) { q[ch] = gload(r0b[ch], gq); };
    auto gather_head = [&](int k, float dx, float dy, float q[5], UpdTaps& T, bool with_r0) {
        const int yc = row_of(k);
        gq = (unsigned)(yc * a.ld + xc) * 4u;
        if (with_r0) {
#pragma unroll
            for (int ch = 0; ch < 5; ch++) gather_r0(ch, q);
        }
        const float fx = (float)xc + dx, fy = (float)yc + dy;
        const float flx = floorf(fx), fly = floorf(fy);
        const bool inb = flx >= 0.f && flx < (float)(a.w - 1) && fly >= 0.f && fly < (float)(a.h - 1);
        const int x1 = inb ? (int)flx : 0, y1 = inb ? (int)fly : 0;
        T.fx = fx - (float)x1;
        T.fy = fy - (float)y1;
        T.inb = inb;
        go0 = (unsigned)(y1 * a.ld + x1) * 4u;
        go1 = go0 + (unsigned)a.ld * 4u;
    };
    // the two taps of a row are adjacent: ONE 8-byte load at 4-byte alignment each (10 instead of 20 tap instructions per
    // pixel: the memory pipe's cost is per wave-instruction — ~11.6 cycles each when it is the only thing running)
    // (round 6: raw buffer loads — one resource for R1, the plane as a scalar byte offset, the lane's 32-bit offset as it is:
    // the global form added the offset to the plane base with a 64-bit VALU add per load, ten per pixel)
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(R1);
    unsigned r1off[5];
#pragma unroll
    for (int ch = 0; ch < 5; ch++) r1off[ch] = (unsigned)((long long)ch * a.ps * 4);
    auto gather_plane = [&](int ch, UpdTaps& T) {
        const f32x2 t01 = bload2(rs1, go0, r1off[ch]);
        const f32x2 t23 = bload2(rs1, go1, r1off[ch]);
        T.t[ch][0] = t01.x;
        T.t[ch][1] = t01.y;
        T.t[ch][2] = t23.x;
        T.t[ch][3] = t23.y;
    };
    // one of a plane's two 8-byte loads (half 0: the row of y1, half 1: the row below)
    auto gather_half = [&](int ch, int half, UpdTaps& T) {
        if (TW_FI_SKIP(32) && ch >= 3) return;  // (timing experiment: 6 instead of 10 tap loads per pixel; wrong results)
        const f32x2 t = bload2(rs1, half ? go1 : go0, r1off[ch]);
        T.t[ch][2 * half] = t.x;
        T.t[ch][2 * half + 1] = t.y;
    };
    auto gather_issue = [&](int k, float dx, float dy, float q[5], UpdTaps& T) {
        gather_head(k, dx, dy, q, T, true);
#pragma unroll
        for (int ch = 0; ch < 5; ch++) gather_plane(ch, T);
    };
    auto combine_store = [&](int k, float dx, float dy, const float q[5], const UpdTaps& T) {
        float M[5];
        update_matrices_combine(q, T, a.w, a.h, xc, row_of(k), dx, dy, M);
#pragma unroll
        for (int ch = 0; ch < 5; ch++) blk[k % NB][ch][cr][cc] = M[ch];
    };

    // ---- prologue: chunks 0 .. NCH-1 (rows ys - MH .. ys + TH + MH - 1) ----
    FlowIn fcur, fnext;
    float q[5];
    UpdTaps T;
    float dx = 0.f, dy = 0.f;
    UpdTaps Tb;                  // DEEP: the second tap set
    float dxb = 0.f, dyb = 0.f;  // ... and its chunk's flow
    if (cth) {
        ytab_issue(0);
        flow_issue(0, fcur);
        ytab_issue(1);
#pragma unroll 1
        for (int k = 0; k < NCH; k++) {
            flow_issue(k + 1, fnext);
            ytab_issue(k + 2);
            flow_value(fcur, dx, dy);
            gather_issue(k, dx, dy, q, T);
            combine_store(k, dx, dy, q, T);
            fcur = fnext;
        }
        if constexpr (DEEP) {
            // chunk NCH: flow -> (dx, dy), head -> T, ALL of its taps go out now (step 0 combines them);
            // chunk NCH + 1: flow -> (dxb, dyb), head -> Tb and the tap offsets go0 / go1 (step 0 issues its taps)
            FlowIn f2;
            flow_issue(NCH + 1, f2);
            ytab_issue(NCH + 2);
            flow_value(fnext, dx, dy);
            gather_head(NCH, dx, dy, q, T, false);
#pragma unroll
            for (int ch = 0; ch < 5; ch++) gather_plane(ch, T);
            flow_value(f2, dxb, dyb);
            gather_head(NCH + 1, dxb, dyb, q, Tb, false);
        } else {
        flow_value(fnext, dx, dy);
        gather_head(NCH, dx, dy, q, T, true);
        }
        // chunk NCH's flow is in dx / dy, its tap addresses are set and its R0 is in flight into q; step 0 loads the flow of
        // chunk NCH + 1 before its V and turns it into addresses in its phase B2.  From here on ONE flow
        // variable is carried: its arrived value is consumed into dx / dy, then the next load is issued into the same
        // registers — with a second variable copied from it the compiler waited for the load it had just issued,
        // vmcnt(0), before the copy.
    }
    __syncthreads();

    // H thread: (plane hp, row hr, 4-pixel group hq) — 1 000 of the 1 024 threads; S thread: one output pixel (sr, sc)
    constexpr int NQ = OUT / 4;
    const bool hth = tid < 5 * TH * NQ;
    const int hp = hth ? tid / (TH * NQ) : 0, hrem = hth ? tid - hp * (TH * NQ) : 0, hr = hrem / NQ, hq = hrem - hr * NQ;
    // (S items 0 .. 767 on waves 0 .. 11, the last 32 on wave 15 — which has no column in V and no pixel in C: with items
    // 768 .. 799 on wave 12, SIMD 0 ran four S waves and the others three)
    constexpr int SMAIN = SPLIT ? (TH * OUT) / 64 * 64 : TH * OUT;
    const int sitem = tid < SMAIN ? tid : (SPLIT && tid >= VWAVES * 64 && tid - VWAVES * 64 < TH * OUT - SMAIN) ? SMAIN + tid - VWAVES * 64 : -1;
    const bool sth = sitem >= 0;
    const int sr = sth ? sitem / OUT : 0, sc = sth ? sitem - sr * OUT : 0;
    const int sxo = bx * OUT + sc;
    const __amdgpu_buffer_rsrc_t rso = make_rsrc(fout);
    const unsigned fout_p1 = (unsigned)(a.fps_out * 4);
    auto s_phase = [&](int st, int s0) {
        const int sy = ys + st * TH + sr;
        if (sth && sxo < a.w && sy < a.h && !TW_FI_SKIP(8)) {
            const double g11 = blk[s0][0][sr][sc], g12 = blk[s0][1][sr][sc], g22 = blk[s0][2][sr][sc],
                         h1 = blk[s0][3][sr][sc], h2 = blk[s0][4][sr][sc];
            const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
            const unsigned o = (unsigned)(sy * a.ld + sxo) * 4u;
            bstore(rso, o, 0u, (float)((g11 * h2 - g12 * h1) * idet));
            bstore(rso, o, fout_p1, (float)((g22 * h1 - g12 * h2) * idet));
        }
    };

    // H of one item: (plane hp, row hr, 4-pixel group hq): 36-value window of block bX, four sums into block s0; WITH_LOADS:
    // the taps of planes 3 and 4 of the thread's phase-C pixel behind the first two pixels (main waves only)
    auto h_phase = [&](int s0, int bX, auto with_loads_c, UpdTaps& Tl) {
        constexpr bool WITH_LOADS = decltype(with_loads_c)::value;
        constexpr int HPLANE0 = DEEP ? 1 : 3;  // the tap planes that go out behind H's first two pixels
        float v[4 + 2 * MH + 2];
        const f32x4* W4 = (const f32x4*)&blk[bX][hp][hr][4 * hq];
#pragma unroll
        for (int u = 0; u < (4 + 2 * MH + 2) / 4; u++) {
            const f32x4 t = W4[u];
            v[4 * u] = t[0];
            v[4 * u + 1] = t[1];
            v[4 * u + 2] = t[2];
            v[4 * u + 3] = t[3];
        }
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int li = MH + j;
            float sum = v[li] * c.k[0];
#pragma unroll
            for (int i = 1; i <= MH; i++) sum += c.k[i] * (v[li - i] + v[li + i]);
            // (the compiler sinks the four sums below the tap loads, towards the store that needs them: planes 3 and 4 then
            // go out at the start of H.  Pinning the sums in place with an opaque use measured 2.6 % slower: gpurun_out/r6b)
            o[j] = sum;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (WITH_LOADS) {
                if constexpr (DEEP) {
#pragma unroll
                    for (int u = 0; u < 10; u++)
                        if (DIST[TW_FI_DIST][u] == TW_FI_P(1, j)) gather_half(u >> 1, u & 1, Tl);
                } else {
                    if (j < 2) gather_plane(HPLANE0 + j, Tl);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        *(f32x4*)&blk[s0][hp][hr][4 * hq] = o;
    };

    if (SPLIT && tid >= VWAVES * 64) {
        // ---- wave 15: no column in V, no pixel in C; 40 of the 1 000 H items and the last 32 S pixels.  A loop of its own (three barriers per
        // step, like the others).  Tried here and dropped: an L2 PREFETCH — this otherwise idle wave touching one dword
        // of every 128-byte line the next chunk's loads will hit (504 lanes, 9 wave-loads per step) — made the launch
        // 10 % SLOWER (46.7 against 42.3 us per 1080p pair, end to end +1.6 % instead of +7.5 %; gpurun_out/r5s): the
        // touches take the miss slots the pixels' own loads are short of.
#pragma unroll 1
        for (int st = 0; st < nsteps; st++) {
            const int s0 = st & (NB - 1), bX = (st + NCH) & (NB - 1);
            TW_FI_SYNC();  // (V)
            if (hth && !TW_FI_SKIP(4)) h_phase(s0, bX, std::false_type(), T);
            TW_FI_SYNC();  // (H)
            s_phase(st, s0);
            TW_FI_SYNC();  // (C, S)
        }
        return;
    }

    // One step.  qc = the R0 coefficients of the chunk this step turns into M (st + NCH; loaded during the PREVIOUS step), qn =
    // those of the chunk after it, loaded during this step: two register sets that swap roles from step to step — the loop
    // is unrolled by two so that the swap is a renaming, never a copy (a copy of a register a load is still in flight for
    // costs an s_waitcnt in place).  With it NO load is left in phase B2 (stamps: seven head loads issued there by all 15
    // waves at once blocked every wave for ~1 000 cycles): the next chunk's flow before V's first row, its R0 behind V's
    // rows, this chunk's taps of planes 0-2 behind rows 0-2 and of planes 3-4 behind H's first pixels.
    auto step = [&](int st, float (&qc)[5], float (&qn)[5]) {
        const bool more = st + 1 < nsteps;  // V(st + 1) will run: it needs chunk st + NCH
        const int s0 = st & (NB - 1), bX = (st + NCH) & (NB - 1);
        TW_FI_STAMP(0);
        // ---- phase A: V(st) -> block X, with loads spread through it ----
        // The memory side of a step (56 B/px + the halo columns: 22.6 us per 1080p pair alone) and its arithmetic (V + H + S:
        // 26.7 us alone) only overlap if requests are QUEUED all through the arithmetic.  Issued in one piece the 405
        // wave-loads of a step block their waves at the issue (the memory pipe buffers a fraction of them) until most are
        // served, and the arithmetic phases then run with an idle memory pipe: 49 us, the plain sum (gpurun_out/r5h).
        // All loads are UNCONDITIONAL for the threads that run V (in the last steps of a segment they fetch chunks nobody
        // needs — row_of clamps them into the image): a conditional load joins with "no load" in a copy of the loaded
        // register, and the compiler waits for the load, vmcnt(0), right where it issued it.
        if (cth) {
            // the flow of chunk st + NCH + 1 (fnext's previous value — chunk st + NCH — went into dx / dy and the tap addresses
            // in the previous step's phase B2; this one is consumed in this step's) and that chunk's R0 offset
            flow_issue(st + NCH + 1, fnext);
            ytab_issue(st + NCH + 2);
            gq = (unsigned)(row_of(st + NCH + 1) * a.ld + xc) * 4u;
            __builtin_amdgcn_sched_barrier(0);
            // The window's chunk j (rows TH*j .. TH*j + TH - 1) sits in block (s0 + j) % NB: the block offsets are wave-uniform
            // (scalar), so a chunk costs one vector add for its address and its TH rows are immediate offsets — no code per
            // rotation.  (An 8-way switch over s0 with the loads AND sums per case compiled without spills, but once global
            // loads were interleaved with the rows the compiler hoisted / joined them across the cases and waited for them
            // in place.)
            float wv[RING];
#pragma unroll
            for (int j = 0; j < NCH; j++) {
                const float* bj = &blk[(s0 + j) & (NB - 1)][cr][0][cc];
#pragma unroll
                for (int r = 0; r < TH; r++) wv[TH * j + r] = bj[r * P];
            }
            float* vout = &blk[bX][cr][0][cc];
#pragma unroll
            for (int r = 0; r < TH; r++) {
                float sv = wv[r + MH] * c.k[0];
#pragma unroll
                for (int i = 1; i <= MH; i++) sv += (wv[r + MH + i] + wv[r + MH - i]) * c.k[i];
                vout[r * P] = sv;
                __builtin_amdgcn_sched_barrier(0);
                if (r < 3) gather_plane(r, T);  // (the taps of planes 0-2 of chunk st + NCH behind rows 0-2; planes 3, 4 inside H)
                gather_r0(r, qn);               // (TH == 5 planes: R0 plane r of chunk st + NCH + 1 behind row r)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        TW_FI_STAMP(1);
        TW_FI_SYNC();
        TW_FI_STAMP(2);

        // ---- phase B1: H — (plane, row, 4-pixel group): 36-value window from block X, four sums into block Y ----
        // (every H thread of these waves issues the tap loads: the few that have no pixel in phase C load from offset 0 of
        // the plane — valid, unused — rather than make the load conditional: see above)
        if (!TW_FI_SKIP(4)) h_phase(s0, bX, std::true_type(), T);
        TW_FI_STAMP(3);
        TW_FI_SYNC();
        TW_FI_STAMP(4);

        // ---- phase B2: S — one pixel per thread from block Y; C — chunk st + NCH into block X; the next chunk's addresses ----
        // (S first: the taps of planes 3 and 4 went out in the middle of H and get S's time to arrive before the combine
        // waits for them)
        s_phase(st, s0);
        __builtin_amdgcn_sched_barrier(0);
        TW_FI_STAMP(5);
        if (cth) {
            if (more && !TW_FI_SKIP(1)) combine_store(st + NCH, dx, dy, qc, T);
            __builtin_amdgcn_sched_barrier(0);
            TW_FI_STAMP(6);
            // the next chunk's flow value and tap addresses (no load: its R0 is on its way into qn, its taps follow in V / H)
            flow_value(fnext, dx, dy);
            gather_head(st + NCH + 1, dx, dy, qn, T, false);
            __builtin_amdgcn_sched_barrier(0);
        }
        TW_FI_STAMP(7);
        TW_FI_SYNC();
    };
    // DEEP: one step.  (dxc, dyc, Tc): the chunk this step turns into M (st + NCH) — flow, fractions and ALL taps, issued a
    // step ago;  (dxn, dyn, Tn): the chunk after it — flow and fractions known, tap offsets in go0 / go1, its taps go out
    // during this step: plane 0 behind V's first row, planes 1, 2 behind H's first pixels, planes 3, 4 in phase B2.  The R0
    // coefficients of chunk st + NCH go out behind V's rows, the flow of chunk st + NCH + 2 before them; at the end of the step
    // that flow becomes (dxc, dyc, Tc) — the registers chunk st + NCH has just left — and go0 / go1.
    auto step_deep = [&](int st, float& dxc, float& dyc, UpdTaps& Tc, float& dxn, float& dyn, UpdTaps& Tn) {
        const bool more = st + 1 < nsteps;
        const int s0 = st & (NB - 1), bX = (st + NCH) & (NB - 1);
        TW_FI_STAMP(0);
        if (cth) {
            flow_issue(st + NCH + 2, fnext);
            ytab_issue(st + NCH + 3);
            gq = (unsigned)(row_of(st + NCH) * a.ld + xc) * 4u;
            __builtin_amdgcn_sched_barrier(0);
            float wv[RING];
#pragma unroll
            for (int j = 0; j < NCH; j++) {
                const float* bj = &blk[(s0 + j) & (NB - 1)][cr][0][cc];
#pragma unroll
                for (int r = 0; r < TH; r++) wv[TH * j + r] = bj[r * P];
            }
            float* vout = &blk[bX][cr][0][cc];
#pragma unroll
            for (int r = 0; r < TH; r++) {
                float sv = wv[r + MH] * c.k[0];
#pragma unroll
                for (int i = 1; i <= MH; i++) sv += (wv[r + MH + i] + wv[r + MH - i]) * c.k[i];
                vout[r * P] = sv;
                __builtin_amdgcn_sched_barrier(0);
                gather_r0(r, q);                  // (R0 plane r of chunk st + NCH: this step's phase B2 combines it)
#pragma unroll
                for (int u = 0; u < 10; u++)
                    if (DIST[TW_FI_DIST][u] == TW_FI_P(0, r)) gather_half(u >> 1, u & 1, Tn);  // (taps of chunk st + NCH + 1)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        TW_FI_STAMP(1);
        TW_FI_SYNC();
        TW_FI_STAMP(2);
        if (!TW_FI_SKIP(4)) h_phase(s0, bX, std::true_type(), Tn);
        TW_FI_STAMP(3);
        TW_FI_SYNC();
        TW_FI_STAMP(4);
        s_phase(st, s0);
        __builtin_amdgcn_sched_barrier(0);
        TW_FI_STAMP(5);
        if (cth) {
#pragma unroll
            for (int u = 0; u < 10; u++)
                if (DIST[TW_FI_DIST][u] == TW_FI_P(2, 0)) gather_half(u >> 1, u & 1, Tn);
            __builtin_amdgcn_sched_barrier(0);
            if (more && !TW_FI_SKIP(1)) combine_store(st + NCH, dxc, dyc, q, Tc);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 10; u++)
                if (DIST[TW_FI_DIST][u] == TW_FI_P(2, 1)) gather_half(u >> 1, u & 1, Tn);
            __builtin_amdgcn_sched_barrier(0);
            TW_FI_STAMP(6);
            // chunk st + NCH + 2: flow value, fractions and tap offsets — into the registers chunk st + NCH has left
            flow_value(fnext, dxc, dyc);
            gather_head(st + NCH + 2, dxc, dyc, q, Tc, false);
            __builtin_amdgcn_sched_barrier(0);
        }
        TW_FI_STAMP(7);
        TW_FI_SYNC();
    };
    if constexpr (DEEP) {
        int st = 0;
#pragma unroll 1
        for (; st + 1 < nsteps; st += 2) {
            step_deep(st, dx, dy, T, dxb, dyb, Tb);
            step_deep(st + 1, dxb, dyb, Tb, dx, dy, T);
        }
        if (st < nsteps) step_deep(st, dx, dy, T, dxb, dyb, Tb);
        return;
    }
    float q1[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    {
        int st = 0;
#pragma unroll 1
        for (; st + 1 < nsteps; st += 2) {
            step(st, q, q1);
            step(st + 1, q1, q);
        }
        if (st < nsteps) step(st, q, q1);
    }
}

// -----------------------------------------------------------------------------------------------------
// tw_blur_solve4q<MH,COLS,HALO,TH,FUSED> (round 4): tw_blur_solve4 with the solve + refresh done BY THE HORIZONTAL ITEM'S
//   OWNER, four pixels of one row per lane.  The horizontal pass already leaves the window averages of a 4-pixel group in
//   its lane's registers (all five planes); tw_blur_solve4 writes them back to the LDS tile, barriers twice and reads them
//   lane-consecutively (one pixel per lane) so that the refresh's accesses are dwords.  Here they never leave the
//   registers: R0 comes in and M goes out as 16-byte accesses per lane (the refresh replay of round 4: 16-byte R0 / M
//   accesses + per-pixel dword taps move the same bytes 4-6 % faster, profiles/r04_refresh_shape.txt), two of the three
//   workgroup barriers and 55 LDS accesses per lane and tile go away.  Same values, same order per value.
//   MEASURED (profiles/r04_blur_quads.md): the last launch of a level -4 %, the refreshing launch +12 % (the 2x2 taps of
//   lanes that sit 16 bytes apart touch four times the cache lines per instruction) — in a BATCH, where the launch is
//   throughput-bound; the variants library keeps that A/B (TW_BLUR_VARIANT=9).
//   Round 6: a SINGLE pair's level-0 launches are bound by the workgroup's serial V -> H -> S chain (1 215 workgroups on 1 024
//   slots: one round and a tail), and there the two barriers and the LDS round trip this kernel leaves out are worth more than
//   the taps' extra cache lines: 51.8 / 47.9 / 35.6 -> 48.6 / 44.5 / 33.3 us for the three launches of a 1080p pair
//   (profiles/r06_single_pair.md) — the single-pair schedule's level-0 kernel since then, span-grid samples included (BlurArgs::grid).
// -----------------------------------------------------------------------------------------------------
template <int MH, int COLS, int HALO, int TH, bool FUSED, int VILP = 2, int HILP = 2>
__global__ __launch_bounds__(COLS) __attribute__((amdgpu_waves_per_eu(4, 8))) void tw_blur_solve4q(BlurArgs a)
{
    constexpr int TW = COLS - 2 * HALO;
    constexpr int NW = TH + 2 * MH;
    __shared__ __attribute__((aligned(16))) float sm[5][TH][COLS];
    const int tid = threadIdx.x;
    int bx, by, z;
    xcd_remap(bx, by, z);
    const int x0 = bx * TW - a.xsh, y0 = by * TH;  // xsh is 0 or 16: x0 is a multiple of 4
    const WinCoef& c = a.c;
    const float* __restrict__ Min = a.Min + (long long)z * 5 * a.ps;
    float* __restrict__ flow = a.flow + (long long)z * 2 * a.fps;
    const int cx0 = max(0, -x0), vw = min(TW, a.w - x0) - cx0;
    const bool partial = !a.nomask && vw <= TW - 32;
    const int gq0 = cx0 >> 2, gv = ((cx0 + vw + 3) >> 2) - gq0;  // 4-pixel groups with a valid pixel
    const float inv_gv = 1.f / (float)gv;

    // ---- V (as tw_blur_solve4) ----
    if (a.nomask || (x0 - HALO + tid >= -MH && x0 - HALO + tid <= a.w - 1 + MH)) {
        const unsigned xb = (unsigned)clampi(x0 - HALO + tid, 0, a.w - 1) * 4u;
        unsigned ro[NW];
#pragma unroll
        for (int i = 0; i < NW; i++) ro[i] = (unsigned)clampi(y0 - MH + i, 0, a.h - 1) * ((unsigned)a.ld * 4u);
        float wa[NW], wb[NW];
        {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(Min);
#pragma unroll
            for (int i = 0; i < NW; i++) wa[i] = bload(rs, xb, ro[i]);
        }
#pragma unroll
        for (int ch = 0; ch < 5; ch++) {
            float* cur = (ch & 1) ? wb : wa;
            float* nxt = (ch & 1) ? wa : wb;
            if (ch < 4) {
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(Min + (long long)(ch + 1) * a.ps);
#pragma unroll
                for (int i = 0; i < NW; i++) nxt[i] = bload(rs, xb, ro[i]);
            }
#pragma unroll
            for (int r = 0; r < TH; r++) {
                float s0 = cur[r + MH] * c.k[0];
#pragma unroll
                for (int i = 1; i <= MH; i++) s0 += (cur[r + MH + i] + cur[r + MH - i]) * c.k[i];
                sm[ch][r][tid] = s0;
                if ((r & (VILP - 1)) == VILP - 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __syncthreads();

    // ---- H + S: one 4-pixel item at a time, its owner solves and refreshes it ----
    constexpr int GROUPS = TW / 4;
    constexpr int NITEM = TH * GROUPS;
    constexpr int ROUNDS = (NITEM + COLS - 1) / COLS;
    constexpr int WL = 4 + 2 * HALO;
    float* __restrict__ Mout = a.Mout + (long long)z * 5 * a.ps;
    const float* __restrict__ R0 = a.R + (long long)(2 * z) * 5 * a.ps;
    const float* __restrict__ R1 = R0 + 5 * a.ps;
#pragma unroll
    for (int rd = 0; rd < ROUNDS; rd++) {
        const int it = tid + rd * COLS;
        int r, q;
        bool on;
        if (partial) {
            on = it < TH * gv;
            r = (int)(((float)it + 0.5f) * inv_gv);
            q = gq0 + it - r * gv;
        } else {
            r = it / GROUPS;
            q = it - r * GROUPS;
            on = it < NITEM && (a.nomask || (x0 + 4 * q < a.w && x0 + 4 * q + 3 >= 0));
        }
        if (!on) continue;
        const int x = x0 + 4 * q, y = y0 + r;
        const int yc = min(y, a.h - 1);
        const bool full = y < a.h && x >= 0 && x + 3 < a.w;  // the whole group lies inside the image
        const long long o = (long long)yc * a.ld + x;        // (used when full)
        // the R0 coefficients do not depend on the flow: fetched now, they fly under the horizontal arithmetic
        f32x4 q4[5];
        if (FUSED && a.update) {
            if (full) {
#pragma unroll
                for (int cc = 0; cc < 5; cc++) q4[cc] = *(const f32x4*)(R0 + o + cc * a.ps);
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const long long oj = (long long)yc * a.ld + clampi(x + j, 0, a.w - 1);
#pragma unroll
                    for (int cc = 0; cc < 5; cc++) q4[cc][j] = R0[oj + cc * a.ps];
                }
            }
        }
        f32x4 res[5];
#pragma unroll
        for (int ch = 0; ch < 5; ch++) {
            float v[WL];
#pragma unroll
            for (int u = 0; u < WL / 4; u++) {
                const f32x4 A = *(const f32x4*)&sm[ch][r][4 * q + 4 * u];
                v[4 * u] = A[0];
                v[4 * u + 1] = A[1];
                v[4 * u + 2] = A[2];
                v[4 * u + 3] = A[3];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int li = HALO + j;
                float sum = v[li] * c.k[0];
#pragma unroll
                for (int i = 1; i <= MH; i++) sum += c.k[i] * (v[li - i] + v[li + i]);
                res[ch][j] = sum;
                if ((j & (HILP - 1)) == HILP - 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
        f32x4 fx4, fy4;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const double g11 = res[0][j], g12 = res[1][j], g22 = res[2][j], h1 = res[3][j], h2 = res[4][j];
            const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
            fx4[j] = (float)((g11 * h2 - g12 * h1) * idet);
            fy4[j] = (float)((g22 * h1 - g12 * h2) * idet);
        }
        if (!a.update || a.store_flow) {
            if (full) {
                *(f32x4*)(flow + o) = fx4;
                *(f32x4*)(flow + o + a.fps) = fy4;
            } else if (y < a.h) {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (x + j >= 0 && x + j < a.w) {
                        flow[(long long)yc * a.ld + x + j] = fx4[j];
                        flow[(long long)yc * a.ld + x + j + a.fps] = fy4[j];
                    }
            }
            if (a.gspan > 0 && y < a.h) {  // (wave-uniform) the span-grid samples of the stored flow, as tw_blur_solve4 writes them
                const unsigned qy = __umulhi((unsigned)y, a.gmagic);
                if (qy * (unsigned)a.gspan == (unsigned)y) {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int xj = x + j;
                        if (xj >= 0 && xj < a.w) {
                            const unsigned qx = __umulhi((unsigned)xj, a.gmagic);
                            if (qx * (unsigned)a.gspan == (unsigned)xj) a.grid[((long long)z * a.gh + qy) * a.gw + qx] = float2{fx4[j], fy4[j]};
                        }
                    }
                }
            }
        }
        if (FUSED) {
            if (a.update) {  // wave-uniform
                f32x4 M4[5];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if ((j & 1) == 0) __builtin_amdgcn_sched_barrier(0);  // two pixels' gathers in flight
                    const float qj[5] = {q4[0][j], q4[1][j], q4[2][j], q4[3][j], q4[4][j]};
                    float M[5];
                    update_matrices_core(qj, R1, a.ps, a.ld, a.w, a.h, clampi(x + j, 0, a.w - 1), yc, fx4[j], fy4[j], M);
#pragma unroll
                    for (int cc = 0; cc < 5; cc++) M4[cc][j] = M[cc];
                }
                if (full) {
#pragma unroll
                    for (int cc = 0; cc < 5; cc++) st_stream<0>((f32x4*)(Mout + o + cc * a.ps), M4[cc]);
                } else if (y < a.h) {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (x + j >= 0 && x + j < a.w) {
#pragma unroll
                            for (int cc = 0; cc < 5; cc++) Mout[(long long)yc * a.ld + x + j + cc * a.ps] = M4[cc][j];
                        }
                }
            }
        }
    }
}

#ifdef TW_VARIANTS  // round-3 trial (TW_BLUR_PIPE), measured 55 % slower: VARIANTS=1 builds only (profiles/r03_blur_pipeline_negative.md)
// -----------------------------------------------------------------------------------------------------
// tw_blur_solve4p<MH,COLS,HALO,TH,NT> : tw_blur_solve4's refreshing launch as a CROSS-TILE PIPELINE.
//   A workgroup walks NT vertically adjacent tiles.  The FarnebackUpdateMatrices refresh of tile t-1 — R0 loads, the
//   flow-dependent R1 gathers, the combine, the M stores: traffic that tw_blur_solve4 leaves exposed at the end of every
//   tile — is interleaved with the VERTICAL pass of tile t, one round of two pixels per M plane:
//       per plane ch:  V arithmetic(ch) | issue window loads(ch+1) | combine + store round ch-1 | issue gathers round ch
//   so the gathers fly under ~370 VALU instructions of window arithmetic and the window loads under the combine.
//   What makes it fit the 128-VGPR budget of four waves per SIMD: one register window (no next-plane double buffer), two
//   pixels of taps in flight, R0 fetched with the gathers (no 35-register prefetch), and the flows of the previous
//   tile's seven pixels carried in 14 registers.  Same values, same order per value as tw_blur_solve4; used for
//   refreshing launches only (the last iteration of a level has no refresh to hide).
// -----------------------------------------------------------------------------------------------------
template <int MH, int COLS, int HALO, int TH, int NT, int WPE = 4>
__global__ __launch_bounds__(COLS) __attribute__((amdgpu_waves_per_eu(WPE, 8))) void tw_blur_solve4p(BlurArgs a)
{
    constexpr int TW = COLS - 2 * HALO;
    constexpr int NW = TH + 2 * MH;
    __shared__ __attribute__((aligned(16))) float sm[5][TH][COLS];
    const int tid = threadIdx.x;
    int bx, by, z;
    xcd_remap(bx, by, z);
    const int x0 = bx * TW - a.xsh;
    const WinCoef& c = a.c;
    const float* __restrict__ Min = a.Min + (long long)z * 5 * a.ps;
    float* __restrict__ flow = a.flow + (long long)z * 2 * a.fps;
    float* __restrict__ Mout = a.Mout + (long long)z * 5 * a.ps;
    const float* __restrict__ R0 = a.R + (long long)(2 * z) * 5 * a.ps;
    const float* __restrict__ R1 = R0 + 5 * a.ps;
    // partial tiles (the last tile column): as in tw_blur_solve4 — the same for every tile of the column
    const int cx0 = max(0, -x0), vw = min(TW, a.w - x0) - cx0;
    const bool partial = !a.nomask && vw <= TW - 32;
    const int gq0 = cx0 >> 2, gv = ((cx0 + vw + 3) >> 2) - gq0;
    const float inv_gv = 1.f / (float)gv, inv_vw = 1.f / (float)vw;
    constexpr int GROUPS = TW / 4;
    constexpr int NITEM = TH * GROUPS;
    constexpr int ROUNDS = (NITEM + COLS - 1) / COLS;
    constexpr int WL = 4 + 2 * HALO;
    constexpr int NPX = (TH * TW + COLS - 1) / COLS;
    constexpr bool RAGGED = (TH * TW) % COLS != 0;
    constexpr int NRND = (NPX + 1) / 2;  // refresh rounds of two pixels
    static_assert(NRND <= 4, "a refresh round per M plane: issue at planes 0..3, combine at planes 1..4");
    // S-phase pixel i of this lane within a tile: row r, column cx; false when the tile has no such pixel for any lane
    // of the wave (wave-uniform: partial tiles take fewer rounds)
    // (`lane` is the thread index, handed in as an opaque copy by every phase: otherwise the coordinates and 64-bit
    // offsets of all seven pixels are hoisted out of the tile loop and live in ~40 registers for the whole kernel)
    auto pixel = [&](int lane, int i, bool& mine, int& r, int& cx) -> bool {
        if (partial) {
            if (i * COLS >= TH * vw) return false;
            mine = lane + i * COLS < TH * vw;
            const int p = mine ? lane + i * COLS : TH * vw - 1;
            r = (int)(((float)p + 0.5f) * inv_vw);
            cx = cx0 + p - r * vw;
        } else {
            mine = !RAGGED || lane + i * COLS < TH * TW;
            const int p = mine ? lane + i * COLS : TH * TW - 1;
            r = p / TW;
            const int c0 = p - r * TW + a.rot;
            cx = c0 >= TW ? c0 - TW : c0;
        }
        return true;
    };
    const bool vact = a.nomask || (x0 - HALO + tid >= -MH && x0 - HALO + tid <= a.w - 1 + MH);
    const unsigned xb = (unsigned)clampi(x0 - HALO + tid, 0, a.w - 1) * 4u;

    float pfx[NPX], pfy[NPX];  // flows of the previous tile's pixels (the refresh that is still owed)
    bool have_prev = false;
    int py0 = 0;
#pragma unroll 1
    for (int t = 0; t <= NT; t++) {
        const int y0 = (by * NT + t) * TH;
        const bool have = t < NT && y0 < a.h;  // workgroup-uniform
        if (!have && !have_prev) break;
        const bool vrun = have && vact;

        // ---- V(t) interleaved with the refresh of tile t-1 ----
        UpdTaps T[2];
        float q[2][5];
        auto issue_round = [&](int g) __attribute__((always_inline)) {
            int lane = tid;
            asm volatile("" : "+v"(lane));
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int i = 2 * g + j;
                if (i >= NPX) break;
                bool mine;
                int r, cx;
                if (!pixel(lane, i, mine, r, cx)) break;
                const int xc = clampi(x0 + cx, 0, a.w - 1), yc = min(py0 + r, a.h - 1);
                const unsigned o = (unsigned)(yc * a.ld + xc) * 4u;
#pragma unroll
                for (int cc = 0; cc < 5; cc++) q[j][cc] = gload(sgpr_base(R0 + cc * a.ps), o);
                update_matrices_gather_s(R1, a.ps, a.ld, a.w, a.h, xc, yc, pfx[i], pfy[i], T[j]);
            }
        };
        auto finish_round = [&](int g) __attribute__((always_inline)) {
            int lane = tid;
            asm volatile("" : "+v"(lane));
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int i = 2 * g + j;
                if (i >= NPX) break;
                bool mine;
                int r, cx;
                if (!pixel(lane, i, mine, r, cx)) break;
                const int x = x0 + cx, y = py0 + r;
                const bool valid = mine && x >= 0 && x < a.w && y < a.h;
                const int xc = clampi(x, 0, a.w - 1), yc = min(y, a.h - 1);
                float M[5];
                update_matrices_combine(q[j], T[j], a.w, a.h, xc, yc, pfx[i], pfy[i], M);
                if (valid) {
                    const unsigned o = (unsigned)(yc * a.ld + xc) * 4u;
#pragma unroll
                    for (int cc = 0; cc < 5; cc++) gstore(sgpr_base_w(Mout + cc * a.ps), o, M[cc]);
                }
            }
        };
        {
            float wa[NW];
            const unsigned pitch = (unsigned)a.ld * 4u;
            // the NW clamped row offsets are recomputed per plane on the scalar unit (an opaque zero keeps them from
            // being hoisted into 38 SGPRs for the whole tile: with the refresh's pointers beside them they would spill)
            auto load_window = [&](int ch) __attribute__((always_inline)) {
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(Min + (long long)ch * a.ps);
                int zero;
                asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
#pragma unroll
                for (int i = 0; i < NW; i++)
                    wa[i] = bload(rs, xb, (unsigned)clampi(y0 - MH + i + zero, 0, a.h - 1) * pitch);
            };
            if (vrun) load_window(0);
            if (have_prev) issue_round(0);
#pragma unroll
            for (int ch = 0; ch < 5; ch++) {
                __builtin_amdgcn_sched_barrier(0);
                if (vrun) {
#pragma unroll
                    for (int r = 0; r < TH; r++) {
                        float s0 = wa[r + MH] * c.k[0];
#pragma unroll
                        for (int i = 1; i <= MH; i++) s0 += (wa[r + MH + i] + wa[r + MH - i]) * c.k[i];
                        sm[ch][r][tid] = s0;
                        if (r & 1) __builtin_amdgcn_sched_barrier(0);  // two rows in flight
                    }
                    if (ch < 4) load_window(ch + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (have_prev) {
                    if (ch < NRND) finish_round(ch);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ch + 1 < NRND) issue_round(ch + 1);
                }
            }
        }
        if (!have) break;  // the drain step: nothing but the last tile's refresh
        __syncthreads();

        // ---- H: all planes, results in registers (as tw_blur_solve4) ----
        f32x4 res[ROUNDS][5];
#pragma unroll
        for (int rd = 0; rd < ROUNDS; rd++) {
            const int it = tid + rd * COLS;
            int r, qd;
            bool on;
            if (partial) {
                on = it < TH * gv;
                r = (int)(((float)it + 0.5f) * inv_gv);
                qd = gq0 + it - r * gv;
            } else {
                r = it / GROUPS;
                qd = it - r * GROUPS;
                on = it < NITEM && (a.nomask || (x0 + 4 * qd < a.w && x0 + 4 * qd + 3 >= 0));
            }
            if (on) {
#pragma unroll
                for (int ch = 0; ch < 5; ch++) {
                    float v[WL];
#pragma unroll
                    for (int u = 0; u < WL / 4; u++) {
                        const f32x4 A = *(const f32x4*)&sm[ch][r][4 * qd + 4 * u];
                        v[4 * u] = A[0];
                        v[4 * u + 1] = A[1];
                        v[4 * u + 2] = A[2];
                        v[4 * u + 3] = A[3];
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int li = HALO + j;
                        float sum = v[li] * c.k[0];
#pragma unroll
                        for (int i = 1; i <= MH; i++) sum += c.k[i] * (v[li - i] + v[li + i]);
                        res[rd][ch][j] = sum;
                        if (j & 1) __builtin_amdgcn_sched_barrier(0);  // two pixels in flight
                    }
                }
            }
        }
        __syncthreads();  // every window has been read: the interiors may be overwritten
#pragma unroll
        for (int rd = 0; rd < ROUNDS; rd++) {
            const int it = tid + rd * COLS;
            int r, qd;
            bool on;
            if (partial) {
                on = it < TH * gv;
                r = (int)(((float)it + 0.5f) * inv_gv);
                qd = gq0 + it - r * gv;
            } else {
                r = it / GROUPS;
                qd = it - r * GROUPS;
                on = it < NITEM && (a.nomask || (x0 + 4 * qd < a.w && x0 + 4 * qd + 3 >= 0));
            }
            if (on) {
#pragma unroll
                for (int ch = 0; ch < 5; ch++) *(f32x4*)&sm[ch][r][HALO + 4 * qd] = res[rd][ch];
            }
        }
        __syncthreads();

        // ---- solve: the flows of this tile's pixels stay in registers; their refresh rides on the next tile's V ----
        int slane = tid;
        asm volatile("" : "+v"(slane));
#pragma unroll
        for (int i = 0; i < NPX; i++) {
            bool mine;
            int r, cx;
            if (!pixel(slane, i, mine, r, cx)) break;
            const double g11 = sm[0][r][HALO + cx], g12 = sm[1][r][HALO + cx], g22 = sm[2][r][HALO + cx],
                         h1 = sm[3][r][HALO + cx], h2 = sm[4][r][HALO + cx];
            const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
            pfx[i] = (float)((g11 * h2 - g12 * h1) * idet);
            pfy[i] = (float)((g22 * h1 - g12 * h2) * idet);
            if (a.store_flow) {
                const int x = x0 + cx, y = y0 + r;
                if (mine && x >= 0 && x < a.w && y < a.h) {
                    const long long o = (long long)y * a.ld + x;
                    flow[o] = pfx[i];
                    flow[o + a.fps] = pfy[i];
                }
            }
        }
        have_prev = true;
        py0 = y0;
        __syncthreads();  // the next tile's V overwrites the LDS tile
    }
}

#endif  // TW_VARIANTS (tw_blur_solve4p)

#ifdef TW_VARIANTS  // A/B kernel (TW_BLUR_VARIANT=60/61, measured 18-68 % slower): VARIANTS=1 builds only
// -----------------------------------------------------------------------------------------------------
// tw_blur_solve6<MH,COLS,HALO,TH> : tw_blur_solve4's arithmetic as a PLANE PIPELINE (VERDICT r1 #5 / DESIGN §9 (a)):
//   the five M planes go one after the other through a two-plane LDS ring (16 KB instead of 40 KB), the horizontal
//   results of all planes stay in registers, the item owner solves its 2 x 4 pixels there, and only the two flow
//   planes are staged in LDS (in the ring, free by then) for the lane-consecutive refresh.  One barrier per plane:
//       V0 | H0 V1 | H1 V2 | H2 V3 | H3 V4 | H4 solve | refresh
//   LDS no longer limits the residency; registers do (no next-plane window prefetch, R0 fetched after the last H).
//   Same values, same order per value.
// -----------------------------------------------------------------------------------------------------
template <int MH, int COLS, int HALO, int TH, int WPE = 5>
__global__ __launch_bounds__(COLS) __attribute__((amdgpu_waves_per_eu(WPE, 8))) void tw_blur_solve6(BlurArgs a)
{
    constexpr int TW = COLS - 2 * HALO;
    constexpr int NW = TH + 2 * MH;
    __shared__ __attribute__((aligned(16))) float ring[2][TH][COLS];
    const int tid = threadIdx.x;
    int bx, by, z;
    xcd_remap(bx, by, z);
    const int x0 = bx * TW - a.xsh, y0 = by * TH;
    const WinCoef& c = a.c;
    const float* __restrict__ Min = a.Min + (long long)z * 5 * a.ps;
    float* __restrict__ flow = a.flow + (long long)z * 2 * a.fps;

    const bool vact = a.nomask || (x0 - HALO + tid >= -MH && x0 - HALO + tid <= a.w - 1 + MH);
    const unsigned xb = (unsigned)clampi(x0 - HALO + tid, 0, a.w - 1) * 4u;
    // vertical pass of one plane into one ring slot
    auto vpass = [&](int ch) {
        if (!vact) return;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(Min + (long long)ch * a.ps);
        float wa[NW];
#pragma unroll
        for (int i = 0; i < NW; i++)
            wa[i] = bload(rs, xb, (unsigned)clampi(y0 - MH + i, 0, a.h - 1) * ((unsigned)a.ld * 4u));
#pragma unroll
        for (int r = 0; r < TH; r++) {
            float s0 = wa[r + MH] * c.k[0];
#pragma unroll
            for (int i = 1; i <= MH; i++) s0 += (wa[r + MH + i] + wa[r + MH - i]) * c.k[i];
            ring[ch & 1][r][tid] = s0;
            if (r & 1) __builtin_amdgcn_sched_barrier(0);  // two rows in flight
        }
    };

    constexpr int GROUPS = TW / 4;
    constexpr int NITEM = TH * GROUPS;
    constexpr int ROUNDS = (NITEM + COLS - 1) / COLS;
    constexpr int WL = 4 + 2 * HALO;
    f32x4 res[5][ROUNDS];
    bool item_on[ROUNDS];
    int item_r[ROUNDS], item_q[ROUNDS];
#pragma unroll
    for (int rd = 0; rd < ROUNDS; rd++) {
        const int it = tid + rd * COLS;
        item_r[rd] = it / GROUPS;
        item_q[rd] = it - item_r[rd] * GROUPS;
        item_on[rd] = it < NITEM && (a.nomask || (x0 + 4 * item_q[rd] < a.w && x0 + 4 * item_q[rd] + 3 >= 0));
    }
    auto hpass = [&](int ch) {
#pragma unroll
        for (int rd = 0; rd < ROUNDS; rd++) {
            if (item_on[rd]) {
                float v[WL];
#pragma unroll
                for (int u = 0; u < WL / 4; u++) {
                    const f32x4 A = *(const f32x4*)&ring[ch & 1][item_r[rd]][4 * item_q[rd] + 4 * u];
                    v[4 * u] = A[0];
                    v[4 * u + 1] = A[1];
                    v[4 * u + 2] = A[2];
                    v[4 * u + 3] = A[3];
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int li = HALO + j;
                    float sum = v[li] * c.k[0];
#pragma unroll
                    for (int i = 1; i <= MH; i++) sum += c.k[i] * (v[li - i] + v[li + i]);
                    res[ch][rd][j] = sum;
                    if (j & 1) __builtin_amdgcn_sched_barrier(0);  // two pixels in flight
                }
                asm volatile("" : "+v"(res[ch][rd]));  // a finished round stays where it is (no sinking below the next one)
            }
        }
    };

    vpass(0);
    __syncthreads();
#pragma unroll
    for (int ch = 0; ch < 5; ch++) {
        hpass(ch);                    // reads slot ch & 1
        __builtin_amdgcn_sched_barrier(0);
        if (ch < 4) vpass(ch + 1);    // writes slot (ch + 1) & 1, which H(ch - 1) finished with before the last barrier
        if (ch < 4) __syncthreads();
    }

    // ---- solve where the horizontal results are (item owner, 2 x 4 pixels); the flows go to the ring, which is free
    // once every wave has finished H(4) ----
    constexpr int NPX = (TH * TW + COLS - 1) / COLS;
    constexpr bool RAGGED = (TH * TW) % COLS != 0;
    // R0 of the lane's refresh pixels: independent of the flow, fetched now (they fly under the solve)
    float qpre[NPX][5];
    if (a.update) {
        const float* __restrict__ R0p = a.R + (long long)(2 * z) * 5 * a.ps;
#pragma unroll
        for (int i = 0; i < NPX; i++) {
            const int p = RAGGED ? min(tid + i * COLS, TH * TW - 1) : tid + i * COLS;
            const int r = p / TW, c0 = p - r * TW + a.rot, cx = c0 >= TW ? c0 - TW : c0;
            const int xc = clampi(x0 + cx, 0, a.w - 1), yc = min(y0 + r, a.h - 1);
            const long long o = (long long)yc * a.ld + xc;
#pragma unroll
            for (int cc = 0; cc < 5; cc++) qpre[i][cc] = R0p[o + cc * a.ps];
        }
    }
    __syncthreads();  // H(4) is done everywhere: both slots are free
#pragma unroll
    for (int rd = 0; rd < ROUNDS; rd++) {
        if (item_on[rd]) {
            f32x4 fxv, fyv;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const double g11 = res[0][rd][j], g12 = res[1][rd][j], g22 = res[2][rd][j], h1 = res[3][rd][j],
                             h2 = res[4][rd][j];
                const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
                fxv[j] = (float)((g11 * h2 - g12 * h1) * idet);
                fyv[j] = (float)((g22 * h1 - g12 * h2) * idet);
            }
            *(f32x4*)&ring[0][item_r[rd]][HALO + 4 * item_q[rd]] = fxv;
            *(f32x4*)&ring[1][item_r[rd]][HALO + 4 * item_q[rd]] = fyv;
        }
    }
    __syncthreads();

    // ---- store the flow / refresh M, lane-consecutive pixels ----
    float* __restrict__ Mout = a.Mout + (long long)z * 5 * a.ps;
    const float* __restrict__ R1 = a.R + (long long)(2 * z) * 5 * a.ps + 5 * a.ps;
#pragma unroll
    for (int i = 0; i < NPX; i++) {
        if (i % 2 == 0) __builtin_amdgcn_sched_barrier(0);  // two pixels in flight
        const bool mine = !RAGGED || tid + i * COLS < TH * TW;
        const int p = mine ? tid + i * COLS : TH * TW - 1;
        const int r = p / TW, c0 = p - r * TW + a.rot, cx = c0 >= TW ? c0 - TW : c0;
        const int x = x0 + cx, y = y0 + r;
        const bool valid = mine && x >= 0 && x < a.w && y < a.h;
        const int xc = clampi(x, 0, a.w - 1), yc = min(y, a.h - 1);
        const float fxv = ring[0][r][HALO + cx], fyv = ring[1][r][HALO + cx];
        const long long o = (long long)yc * a.ld + xc;
        if (valid && (!a.update || a.store_flow)) {
            flow[o] = fxv;
            flow[o + a.fps] = fyv;
        }
        if (a.update) {  // wave-uniform
            float M[5];
            update_matrices_core(qpre[i], R1, a.ps, a.ld, a.w, a.h, xc, yc, fxv, fyv, M);
            if (valid) {
#pragma unroll
                for (int cc = 0; cc < 5; cc++) Mout[o + cc * a.ps] = M[cc];
            }
        }
    }
}

#endif  // TW_VARIANTS

// -----------------------------------------------------------------------------------------------------
// tw_blur_solve_pp<MH,COLS,HALO,TH> : the same arithmetic for SMALL grids (one pair, coarse levels), where a launch
//   is a single round of workgroups and its duration is one workgroup's serial V -> H -> S chain, not throughput.
//   Plane-parallel: a workgroup has 5 x COLS threads, thread group g = tid / COLS owns plane g of M in the vertical
//   and in the horizontal pass (a fifth of tw_blur_solve4's chain per wave), all groups share the solve / refresh.
//   Same LDS layout, same operation order per value; twice the halo overhead at COLS = 64 is irrelevant here.
// -----------------------------------------------------------------------------------------------------
template <int MH, int COLS, int HALO, int TH>
__device__ __forceinline__ void tw_blur_solve_pp_body(const BlurArgs& a, const VGrid& vg, const unsigned vb)
{
    constexpr int TW = COLS - 2 * HALO;
    constexpr int NW = TH + 2 * MH;
    constexpr int NT = 5 * COLS;
    __shared__ __attribute__((aligned(16))) float sm[5][TH][COLS];
    const int tid = threadIdx.x, ch = tid / COLS, ct = tid - ch * COLS;
    int bx, by, z;
    xcd_remap_v(vg, vb, bx, by, z);
    const int x0 = bx * TW - a.xsh, y0 = by * TH;
    const WinCoef& c = a.c;
    const float* __restrict__ Min = a.Min + (long long)z * 5 * a.ps;
    float* __restrict__ flow = a.flow + (long long)z * 2 * a.fps;

    // ---- V: plane ch, column ct ----
    {
        const unsigned xb = (unsigned)clampi(x0 - HALO + ct, 0, a.w - 1) * 4u;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(Min + (long long)ch * a.ps);
        float wa[NW];
#pragma unroll
        for (int i = 0; i < NW; i++)
            wa[i] = bload(rs, xb, (unsigned)clampi(y0 - MH + i, 0, a.h - 1) * ((unsigned)a.ld * 4u));
#pragma unroll
        for (int r = 0; r < TH; r++) {
            float s0 = wa[r + MH] * c.k[0];
#pragma unroll
            for (int i = 1; i <= MH; i++) s0 += (wa[r + MH + i] + wa[r + MH - i]) * c.k[i];
            sm[ch][r][ct] = s0;
            if (r & 1) __builtin_amdgcn_sched_barrier(0);  // two rows in flight
        }
    }
    // the R0 coefficients of the lane's S-phase pixels do not depend on the flow: fetch them under the H pass
    constexpr int NPX = (TH * TW + NT - 1) / NT;
    float qpre[NPX][5];
    if (a.update) {
        const float* __restrict__ R0p = a.R + (long long)(2 * z) * 5 * a.ps;
#pragma unroll
        for (int i = 0; i < NPX; i++) {
            const int p = min(tid + i * NT, TH * TW - 1);
            const int r = p / TW, c0 = p - r * TW + a.rot, cx = c0 >= TW ? c0 - TW : c0;
            const int xc = clampi(x0 + cx, 0, a.w - 1), yc = min(y0 + r, a.h - 1);
            const long long o = (long long)yc * a.ld + xc;
#pragma unroll
            for (int cc = 0; cc < 5; cc++) qpre[i][cc] = R0p[o + cc * a.ps];
        }
    }
    __syncthreads();

    // ---- H: plane ch, 4 pixels per item ----
    constexpr int GROUPS = TW / 4;
    constexpr int NITEM = TH * GROUPS;
    constexpr int ROUNDS = (NITEM + COLS - 1) / COLS;
    constexpr int WL = 4 + 2 * HALO;
    f32x4 res[ROUNDS];
#pragma unroll
    for (int rd = 0; rd < ROUNDS; rd++) {
        const int it = ct + rd * COLS;
        if (it < NITEM) {
            const int r = it / GROUPS, q = it - r * GROUPS;
            float v[WL];
#pragma unroll
            for (int u = 0; u < WL / 4; u++) {
                const f32x4 A = *(const f32x4*)&sm[ch][r][4 * q + 4 * u];
                v[4 * u] = A[0];
                v[4 * u + 1] = A[1];
                v[4 * u + 2] = A[2];
                v[4 * u + 3] = A[3];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int li = HALO + j;
                float sum = v[li] * c.k[0];
#pragma unroll
                for (int i = 1; i <= MH; i++) sum += c.k[i] * (v[li - i] + v[li + i]);
                res[rd][j] = sum;
                if (j & 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __syncthreads();  // every window has been read: the interiors may be overwritten
#pragma unroll
    for (int rd = 0; rd < ROUNDS; rd++) {
        const int it = ct + rd * COLS;
        if (it < NITEM) {
            const int r = it / GROUPS, q = it - r * GROUPS;
            *(f32x4*)&sm[ch][r][HALO + 4 * q] = res[rd];
        }
    }
    __syncthreads();

    // ---- S: solve (+ refresh), lane-consecutive pixels over all 5*COLS threads ----
    float* __restrict__ Mout = a.Mout + (long long)z * 5 * a.ps;
    const float* __restrict__ R1 = a.R + (long long)(2 * z) * 5 * a.ps + 5 * a.ps;
#pragma unroll
    for (int i = 0; i < NPX; i++) {
        const int p = tid + i * NT;
        const bool mine = p < TH * TW;
        const int pc = mine ? p : TH * TW - 1;
        const int r = pc / TW, c0 = pc - r * TW + a.rot, cx = c0 >= TW ? c0 - TW : c0;
        const int x = x0 + cx, y = y0 + r;
        const bool valid = mine && x >= 0 && x < a.w && y < a.h;
        const int xc = clampi(x, 0, a.w - 1), yc = min(y, a.h - 1);
        const double g11 = sm[0][r][HALO + cx], g12 = sm[1][r][HALO + cx], g22 = sm[2][r][HALO + cx],
                     h1 = sm[3][r][HALO + cx], h2 = sm[4][r][HALO + cx];
        const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
        const float fxv = (float)((g11 * h2 - g12 * h1) * idet);
        const float fyv = (float)((g22 * h1 - g12 * h2) * idet);
        const long long o = (long long)yc * a.ld + xc;
        if (valid && (!a.update || a.store_flow)) {
            flow[o] = fxv;
            flow[o + a.fps] = fyv;
        }
        if (a.update) {  // wave-uniform
            float M[5];
            update_matrices_core(qpre[i], R1, a.ps, a.ld, a.w, a.h, xc, yc, fxv, fyv, M);
            if (valid) {
#pragma unroll
                for (int cc = 0; cc < 5; cc++) Mout[o + cc * a.ps] = M[cc];
            }
        }
    }
}
template <int MH, int COLS, int HALO, int TH>
__global__ __launch_bounds__(5 * COLS) void tw_blur_solve_pp(BlurArgs a)
{
    tw_blur_solve_pp_body<MH, COLS, HALO, TH>(a, hw_grid(), hw_block());
}

// -----------------------------------------------------------------------------------------------------
// Twin launches (round 5, the single-pair schedule of BASELINE config 2): TWO kernel bodies in one launch.  A single pair's
//   coarse-level flow chain is a dozen dependent launches of 5-10 us that use a twentieth of the chip; the image-only work of
//   the finer levels (their polynomial expansions, the level 0 / 1 images) used to run beside it from a second stream, whose
//   cross-queue hand-offs cost tens of microseconds on this runtime and whose full-chip launches took the chain's workgroup
//   slots (profiles/r03_latency.md).  Kernels of one queue never overlap (hipExtAnyOrderLaunch is not honoured on gfx9:
//   tools/ubench/anyorder.hip), so the overlap is made inside the launch: workgroups [0, nA) run the chain body (dispatched
//   first), the rest a BAND of the image-only body.  Part A is padded to a multiple of 8 workgroups (see xcd_remap_v).  The
//   bodies are the plain kernels' bodies: same values.  Threads past a body's own workgroup size exit at once (whole waves).
// -----------------------------------------------------------------------------------------------------
// a band of the polynomial expansion by itself (a side job no chain kernel was left to carry)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void tw_polyexp_band(PolyArgs pa, VGrid g)
{
    tw_polyexp_pk_body<7, 8, 0>(pa, g, blockIdx.x);
}
struct TwinGrid {
    VGrid a, b;
    unsigned nA, nA8;  // workgroups of part A, rounded up to a multiple of 8
};
// chain: FarnebackUpdateMatrices (256 threads) | side: a band of the polynomial expansion (256 threads)
template <bool UPSAMPLE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void tw_twin_upd_poly(UpdArgs ua, PolyArgs pa, TwinGrid t)
{
    if (blockIdx.x < t.nA8) {
        if (blockIdx.x < t.nA) tw_update_matrices_body<UPSAMPLE, 2>(ua, t.a, blockIdx.x);
    } else {
        tw_polyexp_pk_body<7, 8, 0>(pa, t.b, blockIdx.x - t.nA8);
    }
}
// chain: plane-parallel window average + solve (320 threads) | side: a band of the polynomial expansion (256 threads)
__global__ __launch_bounds__(320) __attribute__((amdgpu_waves_per_eu(3, 8))) void tw_twin_pp_poly(BlurArgs ba, PolyArgs pa, TwinGrid t)
{
    if (blockIdx.x < t.nA8) {
        if (blockIdx.x < t.nA) tw_blur_solve_pp_body<15, 64, 16, 8>(ba, t.a, blockIdx.x);
    } else {
        if (threadIdx.x >= 256) return;
        tw_polyexp_pk_body<7, 8, 0>(pa, t.b, blockIdx.x - t.nA8);
    }
}
// chain: the 96 x 8-tile window kernel of a level too small to fill the chip (128 threads; BASELINE config 2's level 1) | side: a band
// of the polynomial expansion (256 threads).  The two bodies' LDS tiles (20 KB / 24 KB) share one block: a workgroup runs one body.
// (round 6: level 1's window launches are 680 two-wave workgroups, 1.3 waves per SIMD — the chip has room beside them)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void tw_twin_s4_poly(BlurArgs ba, PolyArgs pa, TwinGrid t)
{
    constexpr size_t LDS_A = sizeof(float) * 5 * BS_TH * 128, LDS_B = sizeof(f32x2) * 3 * (PE_TH / 2) * PE_COLS;
    __shared__ __attribute__((aligned(16))) char lds[LDS_A > LDS_B ? LDS_A : LDS_B];
    if (blockIdx.x < t.nA8) {
        if (blockIdx.x >= t.nA || threadIdx.x >= 128) return;
        tw_blur_solve4_body<15, 128, 16, BS_TH, true>(ba, t.a, blockIdx.x, (float (*)[BS_TH][128])lds);
    } else {
        tw_polyexp_pk_body<7, PE_TH, 0, true>(pa, t.b, blockIdx.x - t.nA8, lds);
    }
}
// chain: the coarsest level's polynomial expansion | side: levels 0 and 1 of the images (tw_pyr_k3f, 256 threads, no remap)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void tw_twin_poly_k3f(PolyArgs pa, PyrK3fArgs ka, TwinGrid t)
{
    if (blockIdx.x < t.nA8) {
        if (blockIdx.x < t.nA) tw_polyexp_pk_body<7, 8, 0>(pa, t.a, blockIdx.x);
    } else {
        const unsigned b = blockIdx.x - t.nA8;
        const unsigned bx = b % t.b.gx, r = b / t.b.gx;
        tw_pyr_k3f_body(ka, (int)bx, (int)(r % t.b.gy), (int)(r / t.b.gy));
    }
}

// (round 3: ships for the 51-tap window of winSize 50 — 58 window rows per 8 output rows become 66 per 16, -7 % on the
//  launch, config 5 0.367 -> 0.385 of its roofline; the 31-tap instantiation stays an A/B kernel of the variants library,
//  TW_BLUR_VARIANT=2: equal within the noise there)
// -----------------------------------------------------------------------------------------------------
// tw_blur_solve4y<MH,COLS,HALO,TH,NSUB> : tw_blur_solve4 with NSUB vertically adjacent TH-row sub-tiles per workgroup.
//   The vertical pass runs once over a TH*NSUB + 2*MH row register window (the window rows of adjacent sub-tiles
//   overlap in all but TH rows, so the L2 -> L1 window traffic per output row drops from (TH+2MH)/TH to
//   (NSUB*TH+2MH)/(NSUB*TH)); the first sub-tile's results go to LDS, the others wait in registers until the
//   LDS tile is free again.  Horizontal pass, solve and refresh are tw_blur_solve4's, once per sub-tile.
//   Same values, same order.  One register window (no next-plane prefetch), R0 prefetched for the last sub-tile.
// -----------------------------------------------------------------------------------------------------
template <int MH, int COLS, int HALO, int TH, int NSUB, int VILP = 2, int HILP = 2, int SUNROLL = 2>
__global__ __launch_bounds__(COLS) __attribute__((amdgpu_waves_per_eu(4, 8))) void tw_blur_solve4y(BlurArgs a)
{
    constexpr int TW = COLS - 2 * HALO;
    constexpr int NR = TH * NSUB;
    constexpr int NW = NR + 2 * MH;
    __shared__ __attribute__((aligned(16))) float sm[5][TH][COLS];
    const int tid = threadIdx.x;
    int bx, by, z;
    if (a.cm) xcd_remap_cm(bx, by, z);
    else xcd_remap(bx, by, z);
    const int x0 = bx * TW - a.xsh, y00 = by * NR;
    const WinCoef& c = a.c;
    const float* __restrict__ Min = a.Min + (long long)z * 5 * a.ps;
    float* __restrict__ flow = a.flow + (long long)z * 2 * a.fps;
    float* __restrict__ Mout = a.Mout + (long long)z * 5 * a.ps;
    const float* __restrict__ R0 = a.R + (long long)(2 * z) * 5 * a.ps;
    const float* __restrict__ R1 = R0 + 5 * a.ps;
    // (round 6: 32-bit offsets behind buffer resources for the solve phase's R0 loads and flow / M stores too)
    const __amdgpu_buffer_rsrc_t rsR0 = make_rsrc(R0), rsF = make_rsrc(flow), rsM = make_rsrc(Mout);

    // ---- V: all sub-tiles ----
    float vlate[5][NR - TH];  // vertical results of the sub-tiles after the first
    {
        unsigned xb = (unsigned)clampi(x0 - HALO + tid, 0, a.w - 1) * 4u;
        const unsigned pitch = (unsigned)a.ld * 4u;
        float wa[NW];
#pragma unroll
        for (int ch = 0; ch < 5; ch++) {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(Min + (long long)ch * a.ps);
            // the NW clamped row offsets are recomputed per plane on the scalar unit (an opaque zero keeps the
            // compiler from hoisting all NW of them into SGPRs for the whole kernel: 46 + the 16 coefficients + the
            // arguments do not fit, and the overflow would land in VGPRs)
            int zero;
            asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
#pragma unroll
            for (int i = 0; i < NW; i++)
                wa[i] = bload(rs, xb, (unsigned)clampi(y00 - MH + i + zero, 0, a.h - 1) * pitch);
#pragma unroll
            for (int r = 0; r < NR; r++) {
                float s0 = wa[r + MH] * c.k[0];
#pragma unroll
                for (int i = 1; i <= MH; i++) s0 += (wa[r + MH + i] + wa[r + MH - i]) * c.k[i];
                if (r < TH) sm[ch][r][tid] = s0;
                else {
                    vlate[ch][r - TH] = s0;
                    asm volatile("" : "+v"(vlate[ch][r - TH]));  // a scalar in a VGPR, not a lane of a wide vector value
                }
                if ((r & (VILP - 1)) == VILP - 1) __builtin_amdgcn_sched_barrier(0);
                // the next plane's loads use xb: tying it to this plane's last result keeps them (and their 46
                // registers) below this plane's arithmetic — one window at a time
                if (r == NR - 1) asm volatile("" : "+v"(xb) : "v"(s0));
            }
        }
    }

    constexpr int GROUPS = TW / 4;
    constexpr int NITEM = TH * GROUPS;
    constexpr int ROUNDS = (NITEM + COLS - 1) / COLS;
    constexpr int WL = 4 + 2 * HALO;
    constexpr int NPX = TH * TW / COLS;
    static_assert((TH * TW) % COLS == 0, "pixels per lane must be whole");
    static_assert(NSUB == 2, "two sub-tiles");
    auto subtile = [&](auto last_c, const int y0) __attribute__((always_inline)) {
        constexpr bool last = decltype(last_c)::value;
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        // the lane index is made opaque here: everything the H and S phases derive from it (LDS addresses, pixel
        // coordinates, 64-bit global offsets of 7 pixels) would otherwise be hoisted above the vertical phase and
        // sit in ~40 VGPRs while the register window and the late results need them
        int t = tid;
        asm volatile("" : "+v"(t));
        float qpre[last ? NPX : 1][5];
        if (last && a.update) {  // `last` is a compile-time constant here
#pragma unroll
            for (int i = 0; i < NPX; i++) {
                const int p = t + i * COLS;
                const int r = p / TW, c0 = p - r * TW + a.rot, cx = c0 >= TW ? c0 - TW : c0;
                const int xc = clampi(x0 + cx, 0, a.w - 1), yc = min(y0 + r, a.h - 1);
                const unsigned o = (unsigned)(yc * a.ld + xc) * 4u;
#pragma unroll
                for (int cc = 0; cc < 5; cc++) qpre[i][cc] = bload(rsR0, o, (unsigned)(cc * a.ps * 4));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 res[ROUNDS][5];
#pragma unroll
        for (int rd = 0; rd < ROUNDS; rd++) {
            const int it = t + rd * COLS;
            if (it < NITEM) {
                const int r = it / GROUPS, q = it - r * GROUPS;
#pragma unroll
                for (int ch = 0; ch < 5; ch++) {
                    float v[WL];
#pragma unroll
                    for (int u = 0; u < WL / 4; u++) {
                        const f32x4 A = *(const f32x4*)&sm[ch][r][4 * q + 4 * u];
                        v[4 * u] = A[0];
                        v[4 * u + 1] = A[1];
                        v[4 * u + 2] = A[2];
                        v[4 * u + 3] = A[3];
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int li = HALO + j;
                        float sum = v[li] * c.k[0];
#pragma unroll
                        for (int i = 1; i <= MH; i++) sum += c.k[i] * (v[li - i] + v[li + i]);
                        res[rd][ch][j] = sum;
                        if ((j & (HILP - 1)) == HILP - 1) __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int rd = 0; rd < ROUNDS; rd++) {
            const int it = t + rd * COLS;
            if (it < NITEM) {
                const int r = it / GROUPS, q = it - r * GROUPS;
#pragma unroll
                for (int ch = 0; ch < 5; ch++) *(f32x4*)&sm[ch][r][HALO + 4 * q] = res[rd][ch];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NPX; i++) {
            if (i % SUNROLL == 0) __builtin_amdgcn_sched_barrier(0);
            const int p = t + i * COLS;
            const int r = p / TW, c0 = p - r * TW + a.rot, cx = c0 >= TW ? c0 - TW : c0;
            const int x = x0 + cx, y = y0 + r;
            const bool valid = x >= 0 && x < a.w && y < a.h;
            const int xc = clampi(x, 0, a.w - 1), yc = min(y, a.h - 1);
            const double g11 = sm[0][r][HALO + cx], g12 = sm[1][r][HALO + cx], g22 = sm[2][r][HALO + cx],
                         h1 = sm[3][r][HALO + cx], h2 = sm[4][r][HALO + cx];
            const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
            const float fxv = (float)((g11 * h2 - g12 * h1) * idet);
            const float fyv = (float)((g22 * h1 - g12 * h2) * idet);
            const unsigned o = (unsigned)(yc * a.ld + xc) * 4u;
            if (valid && (!a.update || a.store_flow)) {
                bstore(rsF, o, 0u, fxv);
                bstore(rsF, o, (unsigned)(a.fps * 4), fyv);
            }
            if (a.update) {  // wave-uniform
                float M[5];
                if constexpr (last) update_matrices_core(qpre[i], R1, a.ps, a.ld, a.w, a.h, xc, yc, fxv, fyv, M);
                else update_matrices_px(R0, R1, a.ps, a.ld, a.w, a.h, xc, yc, fxv, fyv, M);
                if (valid) {
#pragma unroll
                    for (int cc = 0; cc < 5; cc++) bstore(rsM, o, (unsigned)(cc * a.ps * 4), M[cc]);
                }
            }
        }
    };
    subtile(std::false_type{}, y00);
    if (y00 + TH < a.h) {  // workgroup-uniform: otherwise the image ends inside the first sub-tile
        __syncthreads();   // the first sub-tile's S phase has read the tile
#pragma unroll
        for (int ch = 0; ch < 5; ch++)
#pragma unroll
            for (int r = 0; r < TH; r++) sm[ch][r][tid] = vlate[ch][r];
        subtile(std::true_type{}, y00 + TH);
    }
}

// -----------------------------------------------------------------------------------------------------
// tw_blur_solve8<MH,COLS,HALO,TH,FUSED,PREFETCH> : v4's tiling and occupancy (COLS threads, 40 KB LDS, 4
//   workgroups = 16 waves per CU) with packed f32 arithmetic (DESIGN.md §5 has the measurements):
//     V : one column per lane; output rows are produced in PAIRS (r, r+1): the window is held as even-aligned
//         pairs {w[2j], w[2j+1]} plus a copy shifted by one row {w[2j+1], w[2j+2]}, so every tap of a row pair
//         is one aligned register pair;  {s_r, s_r+1} += ({w[a], w[a+1]} + {w[b], w[b+1]}) * k
//     H : an item owns 4 pixels of a row pair; LDS holds float2 {row 2p, row 2p+1} per column
//     S : a lane solves the two vertically adjacent pixels of a column (lane-consecutive x), refresh fused
// -----------------------------------------------------------------------------------------------------
template <int MH, int COLS, int HALO, int TH, bool FUSED, bool PREFETCH>
__global__ __launch_bounds__(COLS) void tw_blur_solve8(BlurArgs a)
{
    constexpr int TW = COLS - 2 * HALO;
    constexpr int NW = TH + 2 * MH;      // window rows (even)
    constexpr int NP = NW / 2;           // even-aligned pairs
    constexpr int RP = TH / 2;
    static_assert(NW % 2 == 0 && TH % 2 == 0 && TW % 4 == 0 && HALO % 2 == 0, "tile shape");
    __shared__ __attribute__((aligned(16))) f32x2 sm[5][RP][COLS];  // {row 2p, row 2p+1} per column
    const int tid = threadIdx.x;
    int bx, by, z;
    xcd_remap(bx, by, z);
    const int x0 = bx * TW, y0 = by * TH;
    const WinCoef& c = a.c;
    const float* __restrict__ Min = a.Min + (long long)z * 5 * a.ps;
    float* __restrict__ flow = a.flow + (long long)z * 2 * a.fps;

    // ---- V ----
    {
        const unsigned xb = (unsigned)clampi(x0 - HALO + tid, 0, a.w - 1) * 4u;
        unsigned ro[NW];
#pragma unroll
        for (int i = 0; i < NW; i++) ro[i] = (unsigned)clampi(y0 - MH + i, 0, a.h - 1) * ((unsigned)a.ld * 4u);
        f32x2 wa[NP], wb[NP];
        auto load_plane = [&](int ch, f32x2* w) {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(Min + (long long)ch * a.ps);
#pragma unroll
            for (int i = 0; i < NP; i++) {
                w[i].x = bload(rs, xb, ro[2 * i]);
                w[i].y = bload(rs, xb, ro[2 * i + 1]);
            }
        };
        load_plane(0, wa);
#pragma unroll
        for (int ch = 0; ch < 5; ch++) {
            f32x2* we = (!PREFETCH || !(ch & 1)) ? wa : wb;
            if (PREFETCH) {
                if (ch < 4) load_plane(ch + 1, (ch & 1) ? wa : wb);
            }
            // window shifted by one row: wo[j] = {w[2j+1], w[2j+2]}
            f32x2 wo[NP - 1];
#pragma unroll
            for (int j = 0; j < NP - 1; j++) wo[j] = f32x2{we[j].y, we[j + 1].x};
            // pair starting at window row t (any parity)
            auto P = [&](int t) -> f32x2 { return (t & 1) ? wo[t >> 1] : we[t >> 1]; };
#pragma unroll
            for (int rp = 0; rp < RP; rp += 2) {
                // two row pairs (4 output rows) advance in lockstep
                f32x2 s0 = P(2 * rp + MH) * c.k[0], s1 = P(2 * rp + 2 + MH) * c.k[0];
#pragma unroll
                for (int i = 1; i <= MH; i++) {
                    f32x2 t0 = P(2 * rp + MH + i) + P(2 * rp + MH - i);
                    f32x2 t1 = P(2 * rp + 2 + MH + i) + P(2 * rp + 2 + MH - i);
                    t0 = t0 * c.k[i];
                    t1 = t1 * c.k[i];
                    s0 += t0;
                    s1 += t1;
                }
                sm[ch][rp][tid] = s0;
                sm[ch][rp + 1][tid] = s1;
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!PREFETCH) {
                if (ch < 4) load_plane(ch + 1, wa);
            }
        }
    }
    __syncthreads();

    // ---- H: 4 pixels x 2 rows per item, all planes, results in registers ----
    constexpr int GROUPS = TW / 4;
    constexpr int NITEM = RP * GROUPS;
    constexpr int ROUNDS = (NITEM + COLS - 1) / COLS;
    constexpr int WL = 4 + 2 * HALO;
    f32x2 res[ROUNDS][5][4];
#pragma unroll
    for (int rd = 0; rd < ROUNDS; rd++) {
        const int it = tid + rd * COLS;
        if (it < NITEM) {
            const int rp = it / GROUPS, q = it - rp * GROUPS;
#pragma unroll
            for (int ch = 0; ch < 5; ch++) {
                f32x2 v[WL];
#pragma unroll
                for (int u = 0; u < WL / 2; u++) {
                    const f32x4 A = *(const f32x4*)&sm[ch][rp][4 * q + 2 * u];
                    v[2 * u] = f32x2{A[0], A[1]};
                    v[2 * u + 1] = f32x2{A[2], A[3]};
                }
                {
                    f32x2 sum[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) sum[j] = v[HALO + j] * c.k[0];
#pragma unroll
                    for (int i = 1; i <= MH; i++) {
                        f32x2 t[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) t[j] = v[HALO + j - i] + v[HALO + j + i];
#pragma unroll
                        for (int j = 0; j < 4; j++) t[j] = c.k[i] * t[j];
#pragma unroll
                        for (int j = 0; j < 4; j++) sum[j] += t[j];
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++) res[rd][ch][j] = sum[j];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int ch = 0; ch < 5; ch++)
#pragma unroll
                for (int j = 0; j < 4; j++) asm volatile("" : "+v"(res[rd][ch][j]));
        }
    }
    __syncthreads();  // every window has been read: the interiors may be overwritten
#pragma unroll
    for (int rd = 0; rd < ROUNDS; rd++) {
        const int it = tid + rd * COLS;
        if (it < NITEM) {
            const int rp = it / GROUPS, q = it - rp * GROUPS;
#pragma unroll
            for (int ch = 0; ch < 5; ch++) {
                *(f32x4*)&sm[ch][rp][HALO + 4 * q] =
                    f32x4{res[rd][ch][0].x, res[rd][ch][0].y, res[rd][ch][1].x, res[rd][ch][1].y};
                *(f32x4*)&sm[ch][rp][HALO + 4 * q + 2] =
                    f32x4{res[rd][ch][2].x, res[rd][ch][2].y, res[rd][ch][3].x, res[rd][ch][3].y};
            }
        }
    }
    __syncthreads();

    // ---- S: solve (+ refresh): a lane takes the two vertically adjacent pixels of a column ----
    float* __restrict__ Mout = a.Mout + (long long)z * 5 * a.ps;
    const float* __restrict__ R0 = a.R + (long long)(2 * z) * 5 * a.ps;
    const float* __restrict__ R1 = R0 + 5 * a.ps;
#pragma unroll 1
    for (int p = tid; p < RP * TW; p += COLS) {
        const int rp = p / TW, cx = p - rp * TW;
        const int x = x0 + cx;
        if (x >= a.w) continue;
        f32x2 b[5];
#pragma unroll
        for (int ch = 0; ch < 5; ch++) b[ch] = sm[ch][rp][HALO + cx];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int y = y0 + 2 * rp + q;
            if (y >= a.h) break;
            const double g11 = b[0][q], g12 = b[1][q], g22 = b[2][q], h1 = b[3][q], h2 = b[4][q];
            const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
            const float fxv = (float)((g11 * h2 - g12 * h1) * idet);
            const float fyv = (float)((g22 * h1 - g12 * h2) * idet);
            const long long o = (long long)y * a.ld + x;
            if (!a.update || a.store_flow) {
                flow[o] = fxv;
                flow[o + a.fps] = fyv;
            }
            if (FUSED) {
                if (a.update) {
                    float M[5];
                    update_matrices_px(R0, R1, a.ps, a.ld, a.w, a.h, x, y, fxv, fyv, M);
#pragma unroll
                    for (int cc = 0; cc < 5; cc++) Mout[o + cc * a.ps] = M[cc];
                }
            }
        }
    }
}

// Generic window size (any m <= 32): same arithmetic, runtime loops, one pixel per thread, no register
// window.  Slow path for non-default winSize.
__global__ __launch_bounds__(256) void tw_blur_solve_generic(BlurArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float gsm[];  // [5][8][64+2m]
    const int m = a.m;
    const int CW = 64 + 2 * m;
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * BS_TH;
    const int z = blockIdx.z;
    const float* __restrict__ Min = a.Min + (long long)z * 5 * a.ps;
    float* __restrict__ Mout = a.Mout + (long long)z * 5 * a.ps;
    float* __restrict__ flow = a.flow + (long long)z * 2 * a.fps;
    const float* __restrict__ R0 = a.R + (long long)(2 * z) * 5 * a.ps;
    const float* __restrict__ R1 = R0 + 5 * a.ps;
    for (int it = tid; it < 5 * BS_TH * CW; it += 256) {
        const int ch = it / (BS_TH * CW);
        const int rem = it - ch * BS_TH * CW;
        const int r = rem / CW, cix = rem - r * CW;
        const int x = clampi(x0 - m + cix, 0, a.w - 1);
        const int y = y0 + r;
        const float* __restrict__ Mp = Min + ch * a.ps + x;
        float s0 = Mp[(long long)clampi(y, 0, a.h - 1) * a.ld] * a.c.k[0];
        for (int i = 1; i <= m; i++)
            s0 += (Mp[(long long)clampi(y + i, 0, a.h - 1) * a.ld] + Mp[(long long)clampi(y - i, 0, a.h - 1) * a.ld]) *
                  a.c.k[i];
        gsm[it] = s0;
    }
    __syncthreads();
    for (int it = tid; it < BS_TH * 64; it += 256) {
        const int r = it >> 6, cx = it & 63;
        const int x = x0 + cx, y = y0 + r;
        if (x >= a.w || y >= a.h) continue;
        float hs[5];
        for (int ch = 0; ch < 5; ch++) {
            const float* v = gsm + (ch * BS_TH + r) * CW + cx + m;
            float sum = v[0] * a.c.k[0];
            for (int i = 1; i <= m; i++) sum += a.c.k[i] * (v[-i] + v[i]);
            hs[ch] = sum;
        }
        const double g11 = hs[0], g12 = hs[1], g22 = hs[2], h1 = hs[3], h2 = hs[4];
        const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
        const float fxv = (float)((g11 * h2 - g12 * h1) * idet);
        const float fyv = (float)((g22 * h1 - g12 * h2) * idet);
        const long long o = (long long)y * a.ld + x;
        if (!a.update || a.store_flow) {
            flow[o] = fxv;
            flow[o + a.fps] = fyv;
        }
        if (a.update) {
            float M[5];
            update_matrices_px(R0, R1, a.ps, a.ld, a.w, a.h, x, y, fxv, fyv, M);
            for (int cc = 0; cc < 5; cc++) Mout[o + cc * a.ps] = M[cc];
        }
    }
}

// =====================================================================================================
// FarnebackUpdateFlow_Blur (flags without OPTFLOW_FARNEBACK_GAUSSIAN): box window with DOUBLE running sums.
//   The CPU code carries one running sum per column down the image and one per row across it; every step
//   rounds, so bit-identical results need the same sequential order.  Two scan kernels (slow path, correct
//   first): tw_box_vscan — one lane per column and plane walks the rows; tw_box_hscan_solve — five lanes per
//   row (one per plane) walk the columns, the lane of plane 0 gathers the five sums with shuffles and solves.
// =====================================================================================================
struct BoxArgs {
    const float* Min;  // pair z at + z*5*ps
    double* V;         // pair z: 5 planes of running column sums at + z*5*ps
    float* flow;       // pair z at + z*2*fps
    int w, h, ld;
    long long ps, fps;
    int m;         // winSize / 2
    double scale;  // 1 / winSize^2
};

__global__ __launch_bounds__(64) void tw_box_vscan(BoxArgs a)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    if (x >= a.w) return;
    const int ch = blockIdx.y, z = blockIdx.z;
    const float* __restrict__ M = a.Min + ((long long)z * 5 + ch) * a.ps + x;
    double* __restrict__ V = a.V + ((long long)z * 5 + ch) * a.ps + x;
    const int m = a.m, h = a.h;
    double vs = (double)(M[0] * (float)(m + 2));  // srow0[x]*(m+2): float product
    for (int y = 1; y < m; y++) vs += (double)M[(long long)min(y, h - 1) * a.ld];
    for (int y = 0; y < h; y++) {
        const float s1 = M[(long long)min(y + m, h - 1) * a.ld], s0 = M[(long long)max(y - m - 1, 0) * a.ld];
        vs += (double)(s1 - s0);
        V[(long long)y * a.ld] = vs;
    }
}

__global__ __launch_bounds__(64) void tw_box_hscan_solve(BoxArgs a)
{
    const int lane = threadIdx.x;
    const int rl = lane / 5, ch = lane - rl * 5;  // 12 rows x 5 planes per wave (lanes 60..63 idle)
    const int y = blockIdx.x * 12 + rl;
    const int z = blockIdx.z;
    const bool on = (lane < 60) && (y < a.h);
    const int yy = min(y, a.h - 1);
    const double* __restrict__ V = a.V + ((long long)z * 5 + ch) * a.ps + (long long)yy * a.ld;
    float* __restrict__ flow = a.flow + (long long)z * 2 * a.fps + (long long)yy * a.ld;
    const int m = a.m, w = a.w;
    // vsum has its first / last element replicated (m+1) times on either side
    double g = V[0] * (double)(m + 2);
    for (int x = 1; x < m; x++) g += V[min(x, w - 1)];
    for (int x = 0; x < w; x++) {
        g += V[min(x + m, w - 1)] - V[max(x - m - 1, 0)];
        const double gs = g * a.scale;
        // lanes 5r .. 5r+4 hold g11, g12, g22, h1, h2 of row r
        const double g12 = __shfl(gs, lane + 1), g22 = __shfl(gs, lane + 2), h1 = __shfl(gs, lane + 3),
                     h2 = __shfl(gs, lane + 4);
        if (on && ch == 0) {
            const double g11 = gs;
            const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
            flow[x] = (float)((g11 * h2 - g12 * h1) * idet);
            flow[x + a.fps] = (float)((g22 * h1 - g12 * h2) * idet);
        }
    }
}

// =====================================================================================================
// K12  the span-grid threshold scan of /root/reference/src/consumer.cpp:60-76, in two kernels:
//   tw_span_gather : one thread per grid point copies (dx,dy) at (x*span, y*span) into a dense per-pair
//                    buffer — fully parallel, launched per level-0 chunk while the flow is still in cache;
//   tw_span_scan   : one 1024-thread workgroup per pair (launched once per batch): len = dx*dx + dy*dy in
//                    float, compared in double against threshold*threshold (strict >), hits compacted in the
//                    reference's row-major order (y, then x) with wave ballots + one LDS scan of the per-wave
//                    counts per round of 32 x 1024 points, so that only the hits cross PCIe.
// =====================================================================================================
struct ScanRec {
    int x, y;
    float dx, dy;
};
// -----------------------------------------------------------------------------------------------------
// tw_blur_grid<MH,SPAN,GR> : the LAST window average + solve of level 0 when nothing but the span grid is read
//   afterwards (tw_submit / tw_diff with the scan-fused option): evaluated only at the grid points
//   (x, y multiples of SPAN) and written straight into the dense grid buffer the ordered scan reads.
//   Same arithmetic and order as tw_blur_solve4 for every value it produces: vertical centre-out sums over
//   replicate-clamped rows, horizontal centre-out sums, 2x2 solve in double.
//   Workgroup: 256 columns (22 grid columns + 15-column halos) x GR grid rows; one (GR-1)*SPAN + 2*MH + 1 row
//   register window per column and plane.
// -----------------------------------------------------------------------------------------------------
struct BlurGridArgs {
    const float* Min;  // pair z: 5 planes at Min + z*5*ps
    float2* g;         // pair z: gw*gh points at g + z*gzs
    int w, h, ld;
    long long ps, gzs;
    int gw, gh;
    WinCoef c;
};

template <int MH, int SPAN, int GR>
__global__ __launch_bounds__(256) void tw_blur_grid(BlurGridArgs a)
{
    constexpr int COLS = 256, XL = 16;       // the tile's first loaded column is XL left of its first grid column
    constexpr int NGC = (COLS - XL - MH) / SPAN;  // grid columns per tile (the last one needs MH columns to its right)
    constexpr int NW = (GR - 1) * SPAN + 2 * MH + 1;
    static_assert(XL >= MH && GR * NGC <= COLS, "tile shape");
    __shared__ float sm[5][GR][COLS];
    __shared__ float res[5][GR][NGC];
    const int tid = threadIdx.x;
    const int bx = blockIdx.x, by = blockIdx.y, z = blockIdx.z;
    const int x0 = bx * NGC * SPAN, gy0 = by * GR;
    const WinCoef& c = a.c;
    const float* __restrict__ Min = a.Min + (long long)z * 5 * a.ps;
    {
        const unsigned xb = (unsigned)clampi(x0 - XL + tid, 0, a.w - 1) * 4u;
        unsigned ro[NW];
#pragma unroll
        for (int i = 0; i < NW; i++) ro[i] = (unsigned)clampi(gy0 * SPAN - MH + i, 0, a.h - 1) * ((unsigned)a.ld * 4u);
        constexpr bool PRE = NW <= 48;  // two windows only while they fit the register budget
        float wa[NW], wb[PRE ? NW : 1];
        {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(Min);
#pragma unroll
            for (int i = 0; i < NW; i++) wa[i] = bload(rs, xb, ro[i]);
        }
#pragma unroll
        for (int ch = 0; ch < 5; ch++) {
            float* cur = (PRE && (ch & 1)) ? wb : wa;
            float* nxt = (PRE && (ch & 1)) ? wa : wb;
            if (PRE && ch < 4) {
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(Min + (long long)(ch + 1) * a.ps);
#pragma unroll
                for (int i = 0; i < NW; i++) nxt[i] = bload(rs, xb, ro[i]);
            }
            if (!PRE && ch > 0) {
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(Min + (long long)ch * a.ps);
#pragma unroll
                for (int i = 0; i < NW; i++) wa[i] = bload(rs, xb, ro[i]);
            }
#pragma unroll
            for (int j = 0; j < GR; j++) {
                const int cc = j * SPAN + MH;
                float s0 = cur[cc] * c.k[0];
#pragma unroll
                for (int i = 1; i <= MH; i++) s0 += (cur[cc + i] + cur[cc - i]) * c.k[i];
                sm[ch][j][tid] = s0;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __syncthreads();
    for (int it = tid; it < 5 * GR * NGC; it += COLS) {
        const int ch = it / (GR * NGC), rem = it - ch * (GR * NGC);
        const int j = rem / NGC, gc = rem - j * NGC;
        const float* v = &sm[ch][j][XL + gc * SPAN];
        float sum = v[0] * c.k[0];
#pragma unroll
        for (int i = 1; i <= MH; i++) sum += c.k[i] * (v[-i] + v[i]);
        res[ch][j][gc] = sum;
    }
    __syncthreads();
    if (tid < GR * NGC) {
        const int j = tid / NGC, gc = tid - j * NGC;
        const int gx = bx * NGC + gc, gy = gy0 + j;
        if (gx < a.gw && gy < a.gh) {
            const double g11 = res[0][j][gc], g12 = res[1][j][gc], g22 = res[2][j][gc], h1 = res[3][j][gc],
                         h2 = res[4][j][gc];
            const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
            float2 o;
            o.x = (float)((g11 * h2 - g12 * h1) * idet);
            o.y = (float)((g22 * h1 - g12 * h2) * idet);
            a.g[(long long)z * a.gzs + (long long)gy * a.gw + gx] = o;
        }
    }
}

struct GatherArgs {
    const float* flow;  // pair z: 2 planes at flow + z*fzs
    long long fzs, fps;
    int ld, span, gw, gh;
    float2* g;  // pair z: gw*gh points at g + z*G
};

__global__ __launch_bounds__(256) void tw_span_gather(GatherArgs a)
{
    const int gx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int gy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (gx >= a.gw || gy >= a.gh) return;
    const float* __restrict__ fxp = a.flow + (long long)blockIdx.z * a.fzs;
    const long long o = (long long)(gy * a.span) * a.ld + gx * a.span;
    float2 v;
    v.x = fxp[o];
    v.y = fxp[o + a.fps];
    a.g[(long long)blockIdx.z * a.gw * a.gh + gy * a.gw + gx] = v;
}

struct ScanArgs {
    const float2* g;  // pair z: dense grid samples at g + z*G
    int span, gw, gh;
    double thr2;
    int* count;    // [pairs]
    ScanRec* rec;  // pair z at rec + z*rec_zs
    long long rec_zs;
    // DIRECT (a single pair: no tw_span_gather launch): the samples come straight from the level-0 flow planes
    const float* flow;  // pair z: 2 planes at flow + z*2*fps
    long long fps;
    int ld;
};

constexpr int SCAN_IT = 32;  // iterations of 1024 points per round

template <bool DIRECT>
__global__ __launch_bounds__(1024) void tw_span_scan(ScanArgs a)
{
    __shared__ int wsum[SCAN_IT * 16];
    __shared__ int wtot[16];
    __shared__ int round_total;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int z = blockIdx.x;
    const int G = a.gw * a.gh;
    const float2* __restrict__ gdense = a.g + (long long)z * G;
    const float* __restrict__ fx = DIRECT ? a.flow + (long long)z * 2 * a.fps : nullptr;
    auto sample = [&](int idx) -> float2 {
        if (!DIRECT) return gdense[idx];
        const int gy = idx / a.gw, gx = idx - gy * a.gw;
        const long long o = (long long)(gy * a.span) * a.ld + gx * a.span;
        return float2{fx[o], fx[o + a.fps]};
    };
    ScanRec* __restrict__ rec = a.rec + (long long)z * a.rec_zs;
    int base_out = 0;
    for (int base = 0; base < G; base += SCAN_IT * 1024) {
        const int nit = min(SCAN_IT, (G - base + 1023) / 1024);
        unsigned mask = 0;
        for (int i0 = 0; i0 < nit; i0 += 8) {
            float2 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int idx = base + (i0 + u) * 1024 + tid;
                v[u] = float2{0.f, 0.f};
                if (i0 + u < nit && idx < G) v[u] = sample(idx);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = i0 + u;
                if (i < nit) {  // wave-uniform
                    const int idx = base + i * 1024 + tid;
                    const float len = (v[u].x * v[u].x) + (v[u].y * v[u].y);
                    const bool f = idx < G && (double)len > a.thr2;
                    const unsigned long long b = __ballot(f);
                    if (lane == 0) wsum[i * 16 + wave] = __popcll(b);
                    mask |= f ? (1u << i) : 0u;
                }
            }
        }
        __syncthreads();
        // exclusive scan of the nit*16 per-wave counts (<= 512 entries) in place: a shuffle scan inside each wave, then the
        // totals of the waves before it (two barriers; the log-step scan through LDS this replaces took 27)
        const int ne = nit * 16;
        const int mine = (tid < ne) ? wsum[tid] : 0;
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        for (int w2 = 0; w2 < wave; w2++) incl += wtot[w2];
        if (tid < ne) wsum[tid] = incl - mine;
        if (tid == ne - 1) round_total = incl;
        __syncthreads();
        for (int i = 0; i < nit; i++) {
            const bool f = (mask >> i) & 1u;
            const unsigned long long b = __ballot(f);
            if (f) {
                const int idx = base + i * 1024 + tid;
                const int gy = idx / a.gw, gx = idx - gy * a.gw;
                const int pos = base_out + wsum[i * 16 + wave] + __popcll(b & ((1ull << lane) - 1ull));
                const float2 vv = sample(idx);
                ScanRec rr;
                rr.x = gx * a.span;
                rr.y = gy * a.span;
                rr.dx = vv.x;
                rr.dy = vv.y;
                rec[pos] = rr;
            }
        }
        base_out += round_total;
        __syncthreads();
    }
    if (tid == 0) a.count[z] = base_out;
}

// tw_span_scan_seg (round 6, the single-pair schedule): the same ordered compaction by NSEG workgroups with no exchange between
//   them.  One workgroup's 16 waves spend ~5 us of VALU issue on a 1080p pair's 20 736 points (tw_span_scan: 8 us of a 350 us pair);
//   here workgroup g owns the points [g*S, (g+1)*S) and finds its output offset by COUNTING the hits before its segment itself
//   (5 instructions per point, all loads in flight) — the last workgroup re-reads 15/16 of the grid from L2, nothing waits on
//   anything.  Same compare (float len, double threshold), same row-major order of the records, same count.
__global__ __launch_bounds__(1024) void tw_span_scan_seg(ScanArgs a, int S)
{
    __shared__ int wsum[SCAN_IT * 16];
    __shared__ int wtot[16];
    __shared__ int wpre[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = a.gw * a.gh;
    const int lo = (int)blockIdx.x * S, hi = min(G, lo + S);
    const float2* __restrict__ gdense = a.g;
    // hits before the segment
    int c = 0;
    for (int i0 = 0; i0 < lo; i0 += 8 * 1024) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int idx = i0 + u * 1024 + tid;
            v[u] = float2{0.f, 0.f};
            if (idx < lo) v[u] = gdense[idx];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int idx = i0 + u * 1024 + tid;
            const float len = (v[u].x * v[u].x) + (v[u].y * v[u].y);
            c += (idx < lo && (double)len > a.thr2) ? 1 : 0;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
    if (lane == 0) wpre[wave] = c;
    // the segment itself: one round of tw_span_scan (S <= SCAN_IT * 1024)
    const int nit = (hi - lo + 1023) / 1024;
    unsigned mask = 0;
    for (int i = 0; i < nit; i++) {
        const int idx = lo + i * 1024 + tid;
        float2 v = float2{0.f, 0.f};
        if (idx < hi) v = gdense[idx];
        const float len = (v.x * v.x) + (v.y * v.y);
        const bool f = idx < hi && (double)len > a.thr2;
        const unsigned long long b = __ballot(f);
        if (lane == 0) wsum[i * 16 + wave] = __popcll(b);
        mask |= f ? (1u << i) : 0u;
    }
    __syncthreads();
    int base_out = 0;
#pragma unroll
    for (int w2 = 0; w2 < 16; w2++) base_out += wpre[w2];
    const int ne = nit * 16;
    const int mine = (tid < ne) ? wsum[tid] : 0;
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    for (int w2 = 0; w2 < wave; w2++) incl += wtot[w2];
    if (tid < ne) wsum[tid] = incl - mine;
    if (hi == G && tid == max(ne, 1) - 1) a.count[0] = base_out + (ne > 0 ? incl : 0);  // the workgroup of the last segment
    __syncthreads();
    for (int i = 0; i < nit; i++) {
        const bool f = (mask >> i) & 1u;
        const unsigned long long b = __ballot(f);
        if (f) {
            const int idx = lo + i * 1024 + tid;
            const int gy = idx / a.gw, gx = idx - gy * a.gw;
            const int pos = base_out + wsum[i * 16 + wave] + __popcll(b & ((1ull << lane) - 1ull));
            const float2 vv = gdense[idx];
            ScanRec rr;
            rr.x = gx * a.span;
            rr.y = gy * a.span;
            rr.dx = vv.x;
            rr.dy = vv.y;
            a.rec[pos] = rr;
        }
    }
}

// =====================================================================================================
// K0  tw_png_unfilter<NWV> : PNG scanline reconstruction (ISO/IEC 15948 §9: None / Sub / Up / Average / Paeth) of 8-bit
//   gray, gray+alpha, RGB or RGBA rows + libpng-1.5 gray conversion, on the device — the part of cv::imread
//   (src/opticalflow.cpp:37-48) that is byte arithmetic with a left / up / up-left dependence, taken off the host's
//   decode threads.  The host inflates; the filtered rows go up as they are.
//   One workgroup per image, one LANE per ROW, rows skewed by one pixel: at step t lane r of wave v reconstructs pixel
//   x = t - r - (64 + S) v of its row.  Then
//     a (left)    = the lane's own previous result
//     b (up)      = lane r-1's previous result (DPP wave_shr:1); lane 0 reads it from the row the previous wave's lane 63
//                   published in LDS S + 1 steps — at least one workgroup barrier — earlier; the first wave of a band
//                   reads the last row of the previous band there
//     c (up-left) = the lane's previous b
//   so a wave advances 64 rows at once and NWV waves 64 NWV rows; a band of 64 NWV rows takes w + 64 NWV + S (NWV - 1)
//   steps, one barrier every S steps.  A pixel's channels ride in the bytes of one dword.  A band whose rows fill only
//   some of the waves takes only the steps those waves need.
// =====================================================================================================
struct PngJob {
    const uint8_t* src;  // filtered rows: h x (1 + w * ch) bytes (filter type byte first)
    uint8_t* dst;        // gray, w x h, dense
    int ch;              // 1 gray, 2 gray + alpha, 3 RGB, 4 RGBA; 0: nothing to do (the image came as gray)
    int pad;
};
struct PngArgs {
    const PngJob* jobs;
    int w, h;
};
constexpr int PNG_S = 64;            // steps between two workgroup barriers
constexpr int PNG_LDS_PIXELS = 32768;  // NWV x WMAX: 128 KB of row buffers per workgroup

// (selects only: the predictor of every filter type is computed for every pixel and picked with per-row masks, so the
// step loop has no branches — rows of one wave have different filter types)
__device__ __forceinline__ unsigned png_paeth(unsigned a, unsigned b, unsigned c)
{
    const int pa = abs((int)b - (int)c), pb = abs((int)a - (int)c), pc = abs((int)a + (int)b - 2 * (int)c);
    const unsigned bc = pb <= pc ? b : c;
    return ((pa <= pb) & (pa <= pc)) ? a : bc;
}

// One band of rows for a compile-time channel count.  Four steps at a time: the 4 * CH raw bytes a lane needs for its
// next four pixels arrive with ONE (unaligned) load issued a whole group earlier, so no step waits for memory — with a
// byte load per step the kernel ran at the L2's latency, ~1 100 cycles per pixel column (4.2 ms per batch of 64 images).
// The rows may be read up to 16 bytes before their first and after their last byte: the staging buffers are padded.
template <int NWV, int CH>
__device__ __forceinline__ void png_unfilter_band(const PngJob& job, int w, int h, int band, unsigned (*edge)[PNG_LDS_PIXELS / NWV + 4])
{
    constexpr int ROWS = 64 * NWV, S = PNG_S, ND = (4 * CH + 3) / 4;  // ND dwords hold a group's 4 * CH bytes
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long stride = (long long)w * CH + 1;
    const int off = lane + (64 + S) * wv;
    const int rows_left = h - band * ROWS;
    const int nact = rows_left >= ROWS ? NWV : (rows_left + 63) / 64;  // waves of this band that own a row
    const int steps = (w + 64 * nact + S * (nact - 1) + S - 1) / S * S;
    const bool dword_out = (w & 3) == 0;  // dst rows then start on a dword (the image slot is 256-byte aligned)
    const int prev_wave = (wv + NWV - 1) % NWV;
    const int row = band * ROWS + tid;
    const bool active = row < h;
    const uint8_t* __restrict__ in = job.src + (long long)(active ? row : 0) * stride;
    uint8_t* __restrict__ out = job.dst + (long long)(active ? row : 0) * w;
    unsigned ft = active ? in[0] : 0u;
    if (ft > 4u) ft = 0u;  // (the host refuses such files before they get here)
    const unsigned m_sub = ft == 1u ? 255u : 0u, m_up = ft == 2u ? 255u : 0u, m_avg = ft == 3u ? 255u : 0u,
                   m_paeth = ft == 4u ? 255u : 0u;
    const bool has_up = row > 0;
    unsigned cur = 0, bprev = 0, acc = 0;
    unsigned nxt[ND];
    auto fetch = [&](int x0) {  // the raw bytes of pixels x0 .. x0 + 3 of the lane's row
        const int xl = min(max(x0, -4), w);
        const uint8_t* q = in + 1 + (long long)xl * CH;
#pragma unroll
        for (int i = 0; i < ND; i++) __builtin_memcpy(&nxt[i], q + 4 * i, 4);
    };
    fetch(-off);
    for (int t0 = 0; t0 < steps; t0 += S) {
#pragma unroll 1
        for (int g = t0; g < t0 + S; g += 4) {
            const int x0 = g - off;
            unsigned raw4[ND];
#pragma unroll
            for (int i = 0; i < ND; i++) raw4[i] = nxt[i];
            fetch(x0 + 4);
            // lane 0: the four pixels above, published by the previous wave (or the previous band) at least one barrier ago
            unsigned up4[4] = {0u, 0u, 0u, 0u};
            if (lane == 0 && has_up && active && x0 >= 0 && x0 < w) {
                const uint4 e = *(const uint4*)&edge[prev_wave][x0];
                up4[0] = e.x;
                up4[1] = e.y;
                up4[2] = e.z;
                up4[3] = e.w;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int x = x0 + k;
                const bool on = active && x >= 0 && x < w;
                unsigned b = dpp_from_prev(cur);  // the row above finished pixel x one step ago
                if (lane == 0) b = up4[k];
                if (!has_up) b = 0u;
                const unsigned left = x > 0 ? cur : 0u, c = x > 0 ? bprev : 0u;
                // pixel k of the group: bytes CH * k .. CH * k + CH - 1 of raw4
                constexpr int dummy = 0;
                (void)dummy;
                const int bit = 8 * CH * k, lo = bit >> 5, sh = bit & 31;
                unsigned raw = raw4[lo] >> sh;
                if (sh + 8 * CH > 32) raw |= raw4[(lo + 1) < ND ? lo + 1 : lo] << (32 - sh);
                unsigned rec = 0;
#pragma unroll
                for (int ci = 0; ci < CH; ci++) {
                    const unsigned av = (left >> (8 * ci)) & 255u, bv = (b >> (8 * ci)) & 255u, cv = (c >> (8 * ci)) & 255u;
                    const unsigned pred = (av & m_sub) | (bv & m_up) | (((av + bv) >> 1) & m_avg) | (png_paeth(av, bv, cv) & m_paeth);
                    rec |= ((((raw >> (8 * ci)) & 255u) + pred) & 255u) << (8 * ci);
                }
                if (on) {
                    cur = rec;
                    bprev = b;
                    if (lane == 63) edge[wv][x] = rec;
                    // libpng 1.5.12 png_do_rgb_to_gray as OpenCV 2.4.9 configures it: truncated 15-bit coefficients
                    unsigned gr = rec & 255u;
                    if (CH >= 3) {
                        const unsigned gg = (rec >> 8) & 255u, bb = (rec >> 16) & 255u;
                        if (!(gr == gg && gg == bb)) gr = (9797u * gr + 19234u * gg + 3737u * bb) >> 15;
                    }
                    if (dword_out) {
                        acc |= gr << (8 * (x & 3));
                        if ((x & 3) == 3) {
                            *(unsigned*)(out + x - 3) = acc;
                            acc = 0;
                        }
                    } else {
                        out[x] = (uint8_t)gr;
                    }
                }
            }
        }
        __syncthreads();
    }
}

template <int NWV>
__global__ __launch_bounds__(64 * NWV) void tw_png_unfilter(PngArgs a)
{
    constexpr int ROWS = 64 * NWV;
    // edge[v][x]: pixel x of the row lane 63 of wave v reconstructed last (+ 4: lane 0 reads whole groups of four)
    __shared__ __attribute__((aligned(16))) unsigned edge[NWV][PNG_LDS_PIXELS / NWV + 4];
    const PngJob job = a.jobs[blockIdx.x];
    if (job.ch == 0) return;  // (workgroup-uniform)
    for (int band = 0; band * ROWS < a.h; band++) {
        if (job.ch == 1) png_unfilter_band<NWV, 1>(job, a.w, a.h, band, edge);
        else if (job.ch == 2) png_unfilter_band<NWV, 2>(job, a.w, a.h, band, edge);
        else if (job.ch == 3) png_unfilter_band<NWV, 3>(job, a.w, a.h, band, edge);
        else png_unfilter_band<NWV, 4>(job, a.w, a.h, band, edge);
    }
}

}  // namespace twk
