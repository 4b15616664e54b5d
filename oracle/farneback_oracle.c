/*
 * farneback_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C99, scalar, -O2 -ffp-contract=off) of the arithmetic on tidal-wave's
 * hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (libtwflow.so) never links, loads or calls it.
 *
 * What it restates
 *   - cv::calcOpticalFlowFarneback as called at /root/reference/src/opticalflow.cpp:83-85 with the
 *     parameters of src/opticalflow.h:28-36, followed by the plane split of :88-91;
 *   - the span-grid threshold scan of /root/reference/src/consumer.cpp:60-76.
 *
 * The Farneback arithmetic itself is NOT in /root/reference: it lives in the un-vendored
 * third-party dependency OpenCV 2.4.x (pinned 2.4.9: README.md:20,158, .travis.yml:8;
 * linked through `pkg-config opencv`, binding.gyp:13-15).  This file restates the published
 * algorithm of that release (modules/video/src/optflowgf.cpp, modules/imgproc/src/smooth.cpp,
 * filter.cpp, imgwarp.cpp, modules/core/src/lapack.cpp — CPU, SSE2 build, no IPP) operation by
 * operation: same float/double types, same accumulation order, no FMA contraction.
 *
 * Parity pin: tests/test_oracle_golden.py checks this code against every golden vector the
 * reference's own test-suite holds for the path — the 24 {x,y,dx,dy} vectors of
 * test/index.coffee:67-91 (float32-exact), the three `vector: []` cases (:17-37, :49-57),
 * and the dims/status fields.  Intermediates (pyramid, polyexp, M) are pinned only
 * transitively through those vectors.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "farneback_oracle.h"

/* ---- OpenCV scalar helpers ------------------------------------------------------------- */

/* cvRound: SSE2 cvtsd2si under the default rounding mode = round-half-to-even. */
static int cv_round(double v) { return (int)lrint(v); }

/* cvFloor; out-of-int-range / NaN inputs give INT_MIN like cvtsd2si does. */
static int cv_floor(double v)
{
    if (!(v > -2147483648.0 && v < 2147483648.0)) return INT32_MIN;
    int i = (int)lrint(v);
    return i - (v < (double)i);
}

/* borderInterpolate(p, len, BORDER_REFLECT_101) */
static int reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

/* ---- cv::getGaussianKernel(n, sigma, CV_32F)  (imgproc/smooth.cpp) ------------------------ */
void orc_gaussian_kernel(int n, double sigma, float* k)
{
    static const float tab[4][7] = {
        {1.f},
        {0.25f, 0.5f, 0.25f},
        {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f},
        {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f}};
    const float* fixed = (n % 2 == 1 && n <= 7 && sigma <= 0) ? tab[n >> 1] : 0;
    double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    double scale2X = -0.5 / (sigmaX * sigmaX);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double t = fixed ? (double)fixed[i] : exp(scale2X * x * x);
        k[i] = (float)t;
        sum += k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) k[i] = (float)(k[i] * sum);
}

/* ---- cv::GaussianBlur on a CV_32FC1 image, BORDER_REFLECT_101 ---------------------------
 * createSeparableLinearFilter: row filter first, then column filter, both float.
 *   row:    ksize<=5  SymmRowSmallFilter  s = S0*k0 + (S-1+S+1)*k1 [+ (S-2+S+2)*k2]
 *           ksize>5   RowFilter           s = k[0]*S[0]; s += k[j]*S[j]   (left to right)
 *   column: ksize==3  SymmColumnSmallFilter  s = (S-1+S+1)*k1 + S0*k0
 *           else      SymmColumnFilter       s = k0*S0; s += k[j]*(S[+j] + S[-j])
 * src is the u8 image (convertTo CV_32F is exact).                                           */
static void gaussian_blur_u8(const uint8_t* src, int w, int h, int ksize, const float* k, float* dst)
{
    int r = ksize / 2;
    const float* kc = k + r; /* centre */
    float* tmp = (float*)malloc((size_t)w * h * sizeof(float));
    float* prow = (float*)malloc((size_t)(w + 2 * r) * sizeof(float));
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * w;
        for (int x = -r; x < w + r; x++) prow[x + r] = (float)s[reflect101(x, w)];
        float* t = tmp + (size_t)y * w;
        const float* S = prow + r;
        if (ksize == 1) {
            for (int x = 0; x < w; x++) t[x] = S[x] * kc[0];
        } else if (ksize == 3) {
            float k0 = kc[0], k1 = kc[1];
            for (int x = 0; x < w; x++) t[x] = S[x] * k0 + (S[x - 1] + S[x + 1]) * k1;
        } else if (ksize == 5) {
            float k0 = kc[0], k1 = kc[1], k2 = kc[2];
            for (int x = 0; x < w; x++)
                t[x] = S[x] * k0 + (S[x - 1] + S[x + 1]) * k1 + (S[x - 2] + S[x + 2]) * k2;
        } else {
            for (int x = 0; x < w; x++) {
                const float* p = S + x - r;
                float s0 = k[0] * p[0];
                for (int j = 1; j < ksize; j++) s0 += k[j] * p[j];
                t[x] = s0;
            }
        }
    }
    const float** rows = (const float**)malloc((size_t)ksize * sizeof(float*));
    for (int y = 0; y < h; y++) {
        for (int j = -r; j <= r; j++) rows[j + r] = tmp + (size_t)reflect101(y + j, h) * w;
        float* d = dst + (size_t)y * w;
        if (ksize == 1) {
            for (int x = 0; x < w; x++) d[x] = kc[0] * rows[r][x];
        } else if (ksize == 3) {
            float f0 = kc[0], f1 = kc[1];
            const float *S0 = rows[0], *S1 = rows[1], *S2 = rows[2];
            for (int x = 0; x < w; x++) d[x] = (S0[x] + S2[x]) * f1 + S1[x] * f0;
        } else {
            for (int x = 0; x < w; x++) {
                float s0 = kc[0] * rows[r][x];
                for (int j = 1; j <= r; j++) s0 += kc[j] * (rows[r + j][x] + rows[r - j][x]);
                d[x] = s0;
            }
        }
    }
    free(rows);
    free(prow);
    free(tmp);
}

/* ---- cv::resize on CV_32F, cn channels  (imgproc/imgwarp.cpp) --------------------------------
 * INTER_LINEAR; when both scale factors are exactly 2 the 2.4 branch switches to the
 * INTER_AREA fast path (sum of the 2x2 block in row-major order, times 1/4).               */
static void resize_linear_f32(const float* src, int sw, int sh, float* dst, int dw, int dh, int cn)
{
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int iscale_x = cv_round(scale_x), iscale_y = cv_round(scale_y);
    int is_area_fast = fabs(scale_x - iscale_x) < DBL_EPSILON && fabs(scale_y - iscale_y) < DBL_EPSILON;
    if (is_area_fast && iscale_x == 2 && iscale_y == 2) {
        /* ResizeAreaFast_Invoker<float,float>: sum += S[0]+S[cn]+S[step]+S[step+cn]; D = sum*0.25f */
        const float scale = 1.f / 4;
        for (int dy = 0; dy < dh; dy++) {
            const float* S0 = src + (size_t)(2 * dy) * sw * cn;
            const float* S1 = S0 + (size_t)sw * cn;
            float* D = dst + (size_t)dy * dw * cn;
            for (int dx = 0; dx < dw; dx++)
                for (int c = 0; c < cn; c++) {
                    int sx = 2 * dx * cn + c;
                    float sum = 0;
                    sum += S0[sx] + S0[sx + cn] + S1[sx] + S1[sx + cn];
                    D[dx * cn + c] = sum * scale;
                }
        }
        return;
    }
    int* xofs = (int*)malloc((size_t)dw * sizeof(int));
    float* alpha = (float*)malloc((size_t)dw * 2 * sizeof(float));
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            if (dx < xmax) xmax = dx;
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        xofs[dx] = sx;
        alpha[dx * 2] = 1.f - fx;
        alpha[dx * 2 + 1] = fx;
    }
    float* hbuf[2];
    int hrow[2] = {-1, -1};
    hbuf[0] = (float*)malloc((size_t)dw * cn * sizeof(float));
    hbuf[1] = (float*)malloc((size_t)dw * cn * sizeof(float));
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        float b0 = 1.f - fy, b1 = fy;
        /* rows are clipped, the weights are not (resizeGeneric_Invoker) */
        int sy0 = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);
        int sy1 = sy + 1 < 0 ? 0 : (sy + 1 >= sh ? sh - 1 : sy + 1);
        int want[2] = {sy0, sy1};
        const float* R[2] = {0, 0};
        int used0 = -1;
        for (int k = 0; k < 2; k++) {
            int slot = -1;
            for (int q = 0; q < 2; q++)
                if (hrow[q] == want[k]) slot = q;
            if (slot < 0) {
                /* horizontally resize the row into a slot the other tap does not hold */
                if (k == 0) slot = (hrow[0] == want[1]) ? 1 : 0;
                else slot = 1 - used0;
                const float* S = src + (size_t)want[k] * sw * cn;
                float* D = hbuf[slot];
                for (int dx = 0; dx < dw; dx++) {
                    int sx = xofs[dx] * cn;
                    if (dx < xmax) {
                        float a0 = alpha[dx * 2], a1 = alpha[dx * 2 + 1];
                        for (int c = 0; c < cn; c++) D[dx * cn + c] = S[sx + c] * a0 + S[sx + cn + c] * a1;
                    } else {
                        for (int c = 0; c < cn; c++) D[dx * cn + c] = S[sx + c] * 1;
                    }
                }
                hrow[slot] = want[k];
            }
            R[k] = hbuf[slot];
            if (k == 0) used0 = slot;
        }
        float* D = dst + (size_t)dy * dw * cn;
        for (int x = 0; x < dw * cn; x++) D[x] = R[0][x] * b0 + R[1][x] * b1;
    }
    free(hbuf[0]);
    free(hbuf[1]);
    free(alpha);
    free(xofs);
}

/* ---- level plan of cv::calcOpticalFlowFarneback (video/optflowgf.cpp) -------------------- */
int orc_level_plan(int w0, int h0, double pyr_scale, int levels, orc_level* out)
{
    const int min_size = 32;
    int k;
    double scale;
    for (k = 0, scale = 1; k < levels; k++) {
        scale *= pyr_scale;
        if (w0 * scale < min_size || h0 * scale < min_size) break;
    }
    levels = k;
    if (out) {
        for (k = levels; k >= 0; k--) {
            int i;
            for (i = 0, scale = 1; i < k; i++) scale *= pyr_scale;
            double sigma = (1. / scale - 1) * 0.5;
            int smooth_sz = cv_round(sigma * 5) | 1;
            if (smooth_sz < 3) smooth_sz = 3;
            out[k].width = cv_round(w0 * scale);
            out[k].height = cv_round(h0 * scale);
            out[k].sigma = sigma;
            out[k].smooth_sz = smooth_sz;
            out[k].scale = scale;
        }
    }
    return levels;
}

/* pyramid level k of one u8 image: convertTo(CV_32F) -> GaussianBlur(full-res) -> resize */
void orc_pyr_level(const uint8_t* img, int w0, int h0, const orc_level* lv, float* I)
{
    float* kern = (float*)malloc((size_t)lv->smooth_sz * sizeof(float));
    float* blurred = (float*)malloc((size_t)w0 * h0 * sizeof(float));
    orc_gaussian_kernel(lv->smooth_sz, lv->sigma, kern);
    gaussian_blur_u8(img, w0, h0, lv->smooth_sz, kern, blurred);
    if (lv->width == w0 && lv->height == h0)
        memcpy(I, blurred, (size_t)w0 * h0 * sizeof(float));
    else
        resize_linear_f32(blurred, w0, h0, I, lv->width, lv->height, 1);
    free(blurred);
    free(kern);
}

/* ---- FarnebackPolyExp setup: g, xg, xxg and the four inverse-Gram entries ------------------
 * invG = G.inv(DECOMP_CHOLESKY): core/lapack.cpp Cholesky<double> on the 6x6 system.         */
void orc_polyexp_setup(int n, double sigma, float* g, float* xg, float* xxg, double ig[4])
{
    /* g, xg, xxg are centred arrays: index [-n..n] -> caller passes base + n */
    if (sigma < FLT_EPSILON) sigma = n * 0.3;
    double s = 0.;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)exp(-x * x / (2 * sigma * sigma));
        s += g[x];
    }
    s = 1. / s;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)(g[x] * s);
        xg[x] = (float)(x * g[x]);
        xxg[x] = (float)(x * x * g[x]);
    }
    double G[6][6];
    memset(G, 0, sizeof(G));
    for (int y = -n; y <= n; y++)
        for (int x = -n; x <= n; x++) {
            G[0][0] += g[y] * g[x];
            G[1][1] += g[y] * g[x] * x * x;
            G[3][3] += g[y] * g[x] * x * x * x * x;
            G[5][5] += g[y] * g[x] * x * x * y * y;
        }
    G[2][2] = G[0][3] = G[0][4] = G[3][0] = G[4][0] = G[1][1];
    G[4][4] = G[3][3];
    G[3][4] = G[4][3] = G[5][5];

    /* Cholesky: A -> L (diagonal stored inverted), then solve L L^T X = I */
    double A[6][6], B[6][6];
    memcpy(A, G, sizeof(A));
    memset(B, 0, sizeof(B));
    for (int i = 0; i < 6; i++) B[i][i] = 1;
    const int m = 6;
    for (int i = 0; i < m; i++) {
        int j, k;
        double t;
        for (j = 0; j < i; j++) {
            t = A[i][j];
            for (k = 0; k < j; k++) t -= A[i][k] * A[j][k];
            A[i][j] = t * A[j][j];
        }
        t = A[i][i];
        for (k = 0; k < j; k++) {
            double u = A[i][k];
            t -= u * u;
        }
        A[i][i] = 1. / sqrt(t);
    }
    for (int i = 0; i < m; i++)
        for (int j = 0; j < m; j++) {
            double t = B[i][j];
            for (int k = 0; k < i; k++) t -= A[i][k] * B[k][j];
            B[i][j] = t * A[i][i];
        }
    for (int i = m - 1; i >= 0; i--)
        for (int j = 0; j < m; j++) {
            double t = B[i][j];
            for (int k = m - 1; k > i; k--) t -= A[k][i] * B[k][j];
            B[i][j] = t * A[i][i];
        }
    ig[0] = B[1][1]; /* ig11 */
    ig[1] = B[0][3]; /* ig03 */
    ig[2] = B[3][3]; /* ig33 */
    ig[3] = B[5][5]; /* ig55 */
}

/* ---- FarnebackPolyExp: src f32 HxW -> dst f32 HxWx5 interleaved -------------------------- */
void orc_polyexp(const float* src, int width, int height, int n, double sigma, float* dst)
{
    float* kbuf = (float*)malloc((size_t)(n * 6 + 3) * sizeof(float));
    float* _row = (float*)malloc((size_t)(width + n * 2) * 3 * sizeof(float));
    float* g = kbuf + n;
    float* xg = g + n * 2 + 1;
    float* xxg = xg + n * 2 + 1;
    float* row = _row + n * 3;
    double ig[4];
    orc_polyexp_setup(n, sigma, g, xg, xxg, ig);
    double ig11 = ig[0], ig03 = ig[1], ig33 = ig[2], ig55 = ig[3];

    for (int y = 0; y < height; y++) {
        float g0 = g[0], g1, g2;
        const float* srow0 = src + (size_t)width * y;
        const float* srow1 = 0;
        float* drow = dst + (size_t)width * 5 * y;
        int x, k;

        /* vertical part of convolution (float) */
        for (x = 0; x < width; x++) {
            row[x * 3] = srow0[x] * g0;
            row[x * 3 + 1] = row[x * 3 + 2] = 0.f;
        }
        for (k = 1; k <= n; k++) {
            g0 = g[k];
            g1 = xg[k];
            g2 = xxg[k];
            srow0 = src + (size_t)width * (y - k > 0 ? y - k : 0);
            srow1 = src + (size_t)width * (y + k < height - 1 ? y + k : height - 1);
            for (x = 0; x < width; x++) {
                float p = srow0[x] + srow1[x];
                float t0 = row[x * 3] + g0 * p;
                float t1 = row[x * 3 + 1] + g1 * (srow1[x] - srow0[x]);
                float t2 = row[x * 3 + 2] + g2 * p;
                row[x * 3] = t0;
                row[x * 3 + 1] = t1;
                row[x * 3 + 2] = t2;
            }
        }

        /* horizontal part of convolution: replicate the first / last triple n times */
        for (x = 0; x < n * 3; x++) {
            row[-1 - x] = row[2 - x];
            row[width * 3 + x] = row[width * 3 + x - 3];
        }

        for (x = 0; x < width; x++) {
            g0 = g[0];
            /* float products widened to double where both factors are float, as in the source */
            double b1 = row[x * 3] * g0, b2 = 0, b3 = row[x * 3 + 1] * g0, b4 = 0, b5 = row[x * 3 + 2] * g0, b6 = 0;
            for (k = 1; k <= n; k++) {
                double tg = row[(x + k) * 3] + row[(x - k) * 3];
                g0 = g[k];
                b1 += tg * g0;
                b4 += tg * xxg[k];
                b2 += (row[(x + k) * 3] - row[(x - k) * 3]) * xg[k];
                b3 += (row[(x + k) * 3 + 1] + row[(x - k) * 3 + 1]) * g0;
                b6 += (row[(x + k) * 3 + 1] - row[(x - k) * 3 + 1]) * xg[k];
                b5 += (row[(x + k) * 3 + 2] + row[(x - k) * 3 + 2]) * g0;
            }
            /* do not store r1 */
            drow[x * 5 + 1] = (float)(b2 * ig11);
            drow[x * 5] = (float)(b3 * ig11);
            drow[x * 5 + 3] = (float)(b1 * ig03 + b4 * ig33);
            drow[x * 5 + 2] = (float)(b1 * ig03 + b5 * ig33);
            drow[x * 5 + 4] = (float)(b6 * ig55);
        }
    }
    free(_row);
    free(kbuf);
}

/* ---- FarnebackUpdateMatrices: rows [y0,y1) ------------------------------------------------ */
void orc_update_matrices(const float* R0a, const float* R1, const float* flowa, float* Ma, int width, int height,
                         int _y0, int _y1)
{
    enum { BORDER = 5 };
    static const float border[BORDER] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
    size_t step1 = (size_t)width * 5;
    for (int y = _y0; y < _y1; y++) {
        const float* flow = flowa + (size_t)y * width * 2;
        const float* R0 = R0a + (size_t)y * step1;
        float* M = Ma + (size_t)y * step1;
        for (int x = 0; x < width; x++) {
            float dx = flow[x * 2], dy = flow[x * 2 + 1];
            float fx = x + dx, fy = y + dy;
            int x1 = cv_floor(fx), y1 = cv_floor(fy);
            float r2, r3, r4, r5, r6;
            fx -= x1;
            fy -= y1;
            if ((unsigned)x1 < (unsigned)(width - 1) && (unsigned)y1 < (unsigned)(height - 1)) {
                const float* ptr = R1 + (size_t)y1 * step1 + (size_t)x1 * 5;
                float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
                r2 = a00 * ptr[0] + a01 * ptr[5] + a10 * ptr[step1] + a11 * ptr[step1 + 5];
                r3 = a00 * ptr[1] + a01 * ptr[6] + a10 * ptr[step1 + 1] + a11 * ptr[step1 + 6];
                r4 = a00 * ptr[2] + a01 * ptr[7] + a10 * ptr[step1 + 2] + a11 * ptr[step1 + 7];
                r5 = a00 * ptr[3] + a01 * ptr[8] + a10 * ptr[step1 + 3] + a11 * ptr[step1 + 8];
                r6 = a00 * ptr[4] + a01 * ptr[9] + a10 * ptr[step1 + 4] + a11 * ptr[step1 + 9];
                r4 = (R0[x * 5 + 2] + r4) * 0.5f;
                r5 = (R0[x * 5 + 3] + r5) * 0.5f;
                r6 = (R0[x * 5 + 4] + r6) * 0.25f;
            } else {
                r2 = r3 = 0.f;
                r4 = R0[x * 5 + 2];
                r5 = R0[x * 5 + 3];
                r6 = R0[x * 5 + 4] * 0.5f;
            }
            r2 = (R0[x * 5] - r2) * 0.5f;
            r3 = (R0[x * 5 + 1] - r3) * 0.5f;
            r2 += r4 * dy + r6 * dx;
            r3 += r6 * dy + r5 * dx;
            if ((unsigned)(x - BORDER) >= (unsigned)(width - BORDER * 2) ||
                (unsigned)(y - BORDER) >= (unsigned)(height - BORDER * 2)) {
                float scale = (x < BORDER ? border[x] : 1.f) * (x >= width - BORDER ? border[width - x - 1] : 1.f) *
                              (y < BORDER ? border[y] : 1.f) * (y >= height - BORDER ? border[height - y - 1] : 1.f);
                r2 *= scale;
                r3 *= scale;
                r4 *= scale;
                r5 *= scale;
                r6 *= scale;
            }
            M[x * 5] = r4 * r4 + r6 * r6;     /* G(1,1) */
            M[x * 5 + 1] = (r4 + r5) * r6;    /* G(1,2) */
            M[x * 5 + 2] = r5 * r5 + r6 * r6; /* G(2,2) */
            M[x * 5 + 3] = r4 * r2 + r6 * r3; /* h(1)   */
            M[x * 5 + 4] = r6 * r2 + r5 * r3; /* h(2)   */
        }
    }
}

/* window kernel of FarnebackUpdateFlow_GaussianBlur: kernel[0..m] */
void orc_window_kernel(int block_size, float* kernel)
{
    int m = block_size / 2;
    double sigma = m * 0.3, s = 1;
    kernel[0] = (float)s;
    for (int i = 1; i <= m; i++) {
        float t = (float)exp(-i * i / (2 * sigma * sigma));
        kernel[i] = t;
        s += t * 2;
    }
    s = 1. / s;
    for (int i = 0; i <= m; i++) kernel[i] = (float)(kernel[i] * s);
}

/* ---- FarnebackUpdateFlow_GaussianBlur ---------------------------------------------------- */
void orc_update_flow_gaussian(const float* R0, const float* R1, float* flowa, float* matM, int width, int height,
                              int block_size, int update_matrices)
{
    int x, y, i;
    int m = block_size / 2;
    int y0 = 0, y1;
    int min_update_stripe = (1 << 10) / width > block_size ? (1 << 10) / width : block_size;
    float* _vsum = (float*)malloc((size_t)((width + m * 2 + 2) * 5) * sizeof(float));
    float* hsum = (float*)malloc((size_t)(width * 5) * sizeof(float));
    float* kernel = (float*)malloc((size_t)(m + 1) * sizeof(float));
    const float** srow = (const float**)malloc((size_t)(m * 2 + 1) * sizeof(float*));
    float* vsum = _vsum + (m + 1) * 5;
    size_t step = (size_t)width * 5;
    orc_window_kernel(block_size, kernel);

    for (y = 0; y < height; y++) {
        double g11, g12, g22, h1, h2;
        float* flow = flowa + (size_t)y * width * 2;
        /* vertical blur */
        for (i = 0; i <= m; i++) {
            srow[m - i] = matM + step * (y - i > 0 ? y - i : 0);
            srow[m + i] = matM + step * (y + i < height - 1 ? y + i : height - 1);
        }
        for (x = 0; x < width * 5; x++) {
            float s0 = srow[m][x] * kernel[0];
            for (i = 1; i <= m; i++) s0 += (srow[m + i][x] + srow[m - i][x]) * kernel[i];
            vsum[x] = s0;
        }
        /* update borders */
        for (x = 0; x < m * 5; x++) {
            vsum[-1 - x] = vsum[4 - x];
            vsum[width * 5 + x] = vsum[width * 5 + x - 5];
        }
        /* horizontal blur */
        for (x = 0; x < width * 5; x++) {
            float sum = vsum[x] * kernel[0];
            for (i = 1; i <= m; i++) sum += kernel[i] * (vsum[x - i * 5] + vsum[x + i * 5]);
            hsum[x] = sum;
        }
        for (x = 0; x < width; x++) {
            g11 = hsum[x * 5];
            g12 = hsum[x * 5 + 1];
            g22 = hsum[x * 5 + 2];
            h1 = hsum[x * 5 + 3];
            h2 = hsum[x * 5 + 4];
            double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
            flow[x * 2] = (float)((g11 * h2 - g12 * h1) * idet);
            flow[x * 2 + 1] = (float)((g22 * h1 - g12 * h2) * idet);
        }
        y1 = y == height - 1 ? height : y - block_size;
        if (update_matrices && (y1 == height || y1 >= y0 + min_update_stripe)) {
            orc_update_matrices(R0, R1, flowa, matM, width, height, y0, y1);
            y0 = y1;
        }
    }
    free(srow);
    free(kernel);
    free(hsum);
    free(_vsum);
}

/* ---- FarnebackUpdateFlow_Blur (box window, double running sums) ------------------------- */
void orc_update_flow_box(const float* R0, const float* R1, float* flowa, float* matM, int width, int height,
                         int block_size, int update_matrices)
{
    int x, y;
    int m = block_size / 2;
    int y0 = 0, y1;
    int min_update_stripe = (1 << 10) / width > block_size ? (1 << 10) / width : block_size;
    double scale = 1. / (block_size * block_size);
    double* _vsum = (double*)malloc((size_t)((width + m * 2 + 2) * 5) * sizeof(double));
    double* vsum = _vsum + (m + 1) * 5;
    size_t step = (size_t)width * 5;

    const float* srow0 = matM;
    for (x = 0; x < width * 5; x++) vsum[x] = srow0[x] * (m + 2);
    for (y = 1; y < m; y++) {
        srow0 = matM + step * (y < height - 1 ? y : height - 1);
        for (x = 0; x < width * 5; x++) vsum[x] += srow0[x];
    }
    for (y = 0; y < height; y++) {
        double g11, g12, g22, h1, h2;
        float* flow = flowa + (size_t)y * width * 2;
        srow0 = matM + step * (y - m - 1 > 0 ? y - m - 1 : 0);
        const float* srow1 = matM + step * (y + m < height - 1 ? y + m : height - 1);
        for (x = 0; x < width * 5; x++) vsum[x] += srow1[x] - srow0[x];
        for (x = 0; x < (m + 1) * 5; x++) {
            vsum[-1 - x] = vsum[4 - x];
            vsum[width * 5 + x] = vsum[width * 5 + x - 5];
        }
        g11 = vsum[0] * (m + 2);
        g12 = vsum[1] * (m + 2);
        g22 = vsum[2] * (m + 2);
        h1 = vsum[3] * (m + 2);
        h2 = vsum[4] * (m + 2);
        for (x = 1; x < m; x++) {
            g11 += vsum[x * 5];
            g12 += vsum[x * 5 + 1];
            g22 += vsum[x * 5 + 2];
            h1 += vsum[x * 5 + 3];
            h2 += vsum[x * 5 + 4];
        }
        for (x = 0; x < width; x++) {
            g11 += vsum[(x + m) * 5] - vsum[(x - m) * 5 - 5];
            g12 += vsum[(x + m) * 5 + 1] - vsum[(x - m) * 5 - 4];
            g22 += vsum[(x + m) * 5 + 2] - vsum[(x - m) * 5 - 3];
            h1 += vsum[(x + m) * 5 + 3] - vsum[(x - m) * 5 - 2];
            h2 += vsum[(x + m) * 5 + 4] - vsum[(x - m) * 5 - 1];
            double g11_ = g11 * scale, g12_ = g12 * scale, g22_ = g22 * scale, h1_ = h1 * scale, h2_ = h2 * scale;
            double idet = 1. / (g11_ * g22_ - g12_ * g12_ + 1e-3);
            flow[x * 2] = (float)((g11_ * h2_ - g12_ * h1_) * idet);
            flow[x * 2 + 1] = (float)((g22_ * h1_ - g12_ * h2_) * idet);
        }
        y1 = y == height - 1 ? height : y - block_size;
        if (update_matrices && (y1 == height || y1 >= y0 + min_update_stripe)) {
            orc_update_matrices(R0, R1, flowa, matM, width, height, y0, y1);
            y0 = y1;
        }
    }
    free(_vsum);
}

/* resize(prevFlow -> flow, INTER_LINEAR) on CV_32FC2, then flow *= 1/pyr_scale (float scale) */
void orc_flow_upsample(const float* prev, int pw, int ph, float* flow, int w, int h, double pyr_scale)
{
    resize_linear_f32(prev, pw, ph, flow, w, h, 2);
    float s = (float)(1. / pyr_scale);
    for (size_t i = 0; i < (size_t)w * h * 2; i++) flow[i] = flow[i] * s + 0.f;
}

/* ---- cv::calcOpticalFlowFarneback + split (src/opticalflow.cpp:83-91) -------------------- */
int orc_farneback(const uint8_t* prev0, const uint8_t* next0, int w0, int h0, const orc_params* p, float* flowx,
                  float* flowy)
{
    if (!(p->pyrScale < 1) || w0 <= 0 || h0 <= 0) return -1;
    orc_level lv[64];
    int req = p->pyrLevels > 60 ? 60 : p->pyrLevels;
    int levels = orc_level_plan(w0, h0, p->pyrScale, req, lv);
    const uint8_t* img[2] = {prev0, next0};
    float* prevFlow = 0;
    int pw = 0, ph = 0;
    float* flow = 0;
    for (int k = levels; k >= 0; k--) {
        int width = lv[k].width, height = lv[k].height;
        size_t N = (size_t)width * height;
        flow = (float*)malloc(N * 2 * sizeof(float));
        if (!prevFlow) {
            /* OPTFLOW_USE_INITIAL_FLOW (flags&4) would read the caller's flow0, which the
             * reference never initialises (src/opticalflow.cpp:80) — defined here as zero. */
            memset(flow, 0, N * 2 * sizeof(float));
        } else {
            orc_flow_upsample(prevFlow, pw, ph, flow, width, height, p->pyrScale);
            free(prevFlow);
        }
        float* R[2];
        float* I = (float*)malloc(N * sizeof(float));
        for (int i = 0; i < 2; i++) {
            orc_pyr_level(img[i], w0, h0, &lv[k], I);
            R[i] = (float*)malloc(N * 5 * sizeof(float));
            orc_polyexp(I, width, height, p->polyN, p->polySigma, R[i]);
        }
        free(I);
        float* M = (float*)malloc(N * 5 * sizeof(float));
        orc_update_matrices(R[0], R[1], flow, M, width, height, 0, height);
        for (int i = 0; i < p->pyrIterations; i++) {
            if (p->flags & 256)
                orc_update_flow_gaussian(R[0], R[1], flow, M, width, height, p->winSize, i < p->pyrIterations - 1);
            else
                orc_update_flow_box(R[0], R[1], flow, M, width, height, p->winSize, i < p->pyrIterations - 1);
        }
        free(M);
        free(R[0]);
        free(R[1]);
        prevFlow = flow;
        pw = width;
        ph = height;
    }
    for (size_t i = 0; i < (size_t)w0 * h0; i++) {
        flowx[i] = flow[i * 2];
        flowy[i] = flow[i * 2 + 1];
    }
    free(flow);
    return 0;
}

/* ---- cv::resize on CV_8UC1 as called at /root/reference/src/opticalflow.cpp:64-68 -----------------
 * `cv::resize(targetImg, resized, resized.size(), cv::INTER_NEAREST)`: the constant 0 lands in the
 * `fx` parameter; dsize is non-empty, so the scale comes from dsize and the interpolation is the
 * default INTER_LINEAR.  OpenCV 2.4.9 imgproc/imgwarp.cpp, depth CV_8U
 * ("fixpt"), restated per OUTPUT PIXEL (no row cache, no coefficient tables shared with any other
 * implementation in this repository):
 *   - cv::resize: scale = 1/((double)dsize/ssize); an exact 2 x 2 reduction is switched to
 *     INTER_AREA (ResizeAreaFast_Invoker<uchar,int>: (S00+S01+S10+S11+2)>>2);
 *   - coordinate rule: f = (float)((d+0.5)*scale-0.5); s = cvFloor(f); f -= s; in x only: s<0 ->
 *     (0, f=0), s>=ssize-1 -> (ssize-1, f=0); in y the rows are clipped later, the weights are not;
 *   - coefficients: saturate_cast<short>(w * INTER_RESIZE_COEF_SCALE) with INTER_RESIZE_COEF_SCALE =
 *     1<<11, w in {1.f-f, f} (float product, cvRound = round-half-even);
 *   - HResizeLinear<uchar,int,short,2048>: D = S[sx]*a0 + S[sx+1]*a1 while sx+1 < ssize.width
 *     (dx < xmax), D = S[sx]*2048 beyond;
 *   - VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>:
 *     dst = uchar((((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2), rows sy and sy+1 clipped to
 *     [0, ssize.height) (resizeGeneric_Invoker's clip()).
 * Parity unpinned: no fixture of the reference has a pair of unequal sizes.                        */
static short orc_sat_short(float v)
{
    int r = cv_round((double)v);
    return (short)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
}
static int orc_clip(int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; }
/* one horizontally interpolated sample of source row `row` (already clipped), in 11-bit fixed point */
static int orc_hsample_u8(const uint8_t* src, int sw, int row, int dx, double scale_x)
{
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    const uint8_t* S = src + (size_t)row * sw;
    if (sx + 1 >= sw) return S[sx] * 2048; /* dx >= xmax */
    float c0 = 1.f - fx, c1 = fx;
    return S[sx] * orc_sat_short(c0 * 2048) + S[sx + 1] * orc_sat_short(c1 * 2048);
}
void orc_resize_u8_linear(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh)
{
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int iscale_x = (int)lrint(scale_x), iscale_y = (int)lrint(scale_y); /* saturate_cast<int>(double) = cvRound */
    int is_area_fast = fabs(scale_x - iscale_x) < DBL_EPSILON && fabs(scale_y - iscale_y) < DBL_EPSILON;
    if (is_area_fast && iscale_x == 2 && iscale_y == 2) {
        for (int dy = 0; dy < dh; dy++)
            for (int dx = 0; dx < dw; dx++) {
                const uint8_t* S = src + (size_t)(2 * dy) * sw + 2 * dx;
                dst[(size_t)dy * dw + dx] = (uint8_t)((S[0] + S[1] + S[sw] + S[sw + 1] + 2) >> 2);
            }
        return;
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        short b0 = orc_sat_short((1.f - fy) * 2048), b1 = orc_sat_short(fy * 2048);
        int r0 = orc_clip(sy, 0, sh), r1 = orc_clip(sy + 1, 0, sh);
        for (int dx = 0; dx < dw; dx++) {
            int S0 = orc_hsample_u8(src, sw, r0, dx, scale_x);
            int S1 = orc_hsample_u8(src, sw, r1, dx, scale_x);
            dst[(size_t)dy * dw + dx] = (uint8_t)((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
        }
    }
}

/* OpticalFlow::calculate's size rule (/root/reference/src/opticalflow.cpp:52-68) on decoded gray images:
 * returns 3 (DontMatchSize) when either dimension differs by more than 5, else 0 with `target_out`
 * (ew x eh) = the target as calculateInternal receives it (a copy when the sizes are equal).       */
int orc_reconcile_target(const uint8_t* target, int tw, int th, int ew, int eh, uint8_t* target_out)
{
    if (abs(eh - th) > 5 || abs(ew - tw) > 5) return 3;
    if (eh != th || ew != tw) orc_resize_u8_linear(target, tw, th, target_out, ew, eh);
    else memcpy(target_out, target, (size_t)ew * eh);
    return 0;
}

/* ---- span-grid threshold scan (src/consumer.cpp:60-76) ------------------------------------ */
int orc_span_scan(const float* flowx, const float* flowy, int w, int h, int span, double threshold, orc_vector* out,
                  int cap)
{
    int n = 0;
    for (int y = 0; y < h; ++y) {
        if (y % span != 0) continue;
        for (int x = 0; x < w; ++x) {
            if (x % span != 0) continue;
            float dx = flowx[(size_t)y * w + x];
            float dy = flowy[(size_t)y * w + x];
            float len = (dx * dx) + (dy * dy);
            if (len > (threshold * threshold)) {
                if (n < cap) {
                    out[n].x = x;
                    out[n].y = y;
                    out[n].dx = dx;
                    out[n].dy = dy;
                }
                n++;
            }
        }
    }
    return n;
}
