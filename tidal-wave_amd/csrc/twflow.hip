// twflow.hip — engine + C ABI of libtwflow.so (see include/twflow.h).
//
// One tw_engine = one worker's view of one MI355X: `slots` independent in-flight image pairs, each with
// its own HIP stream and device workspace (no allocation per job in steady state).  The host side only
// builds the small per-level tables (Gaussian taps, bilinear coordinates, polynomial-expansion constants)
// in double precision exactly as OpenCV 2.4.9 does on the CPU, and enqueues the kernels of
// twflow_kernels.hip.h.  There is no CPU compute path in this library.
#include "twflow_kernels.hip.h"

#include <hip/hip_runtime.h>
#include <unistd.h>

#include <dlfcn.h>
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/twflow.h"
#include "../../include/twflow_debug.h"

using namespace twk;

namespace {

// ---------------------------------------------------------------------------------------------------
// Host-side scalar helpers (OpenCV 2.4.9 semantics)
// ---------------------------------------------------------------------------------------------------
inline int cv_round(double v) { return (int)lrint(v); }  // cvtsd2si: round-half-even
inline int cv_floor(double v)
{
    if (!(v > -2147483648.0 && v < 2147483648.0)) return INT32_MIN;
    int i = (int)lrint(v);
    return i - (v < (double)i);
}
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// cv::getGaussianKernel(n, sigma, CV_32F) — imgproc/smooth.cpp
void gaussian_kernel(int n, double sigma, std::vector<float>& k)
{
    static const float tab[4][7] = {{1.f},
                                    {0.25f, 0.5f, 0.25f},
                                    {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f},
                                    {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f}};
    const float* fixed = (n % 2 == 1 && n <= 7 && sigma <= 0) ? tab[n >> 1] : nullptr;
    k.resize(n);
    const double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    const double scale2X = -0.5 / (sigmaX * sigmaX);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        const double x = i - (n - 1) * 0.5;
        const double t = fixed ? (double)fixed[i] : exp(scale2X * x * x);
        k[i] = (float)t;
        sum += k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) k[i] = (float)(k[i] * sum);
}

// Coordinate tables of cv::resize(INTER_LINEAR) for float data — imgproc/imgwarp.cpp
struct ResizeTab {
    std::vector<int> xofs, yofs;
    std::vector<float> alpha, beta;
    int xmax = 0;
    int mode = 1;  // 0 same size, 1 linear, 2 area-fast 2x2
};

void make_resize_tab(int sw, int sh, int dw, int dh, ResizeTab& t)
{
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    const int iscale_x = cv_round(scale_x), iscale_y = cv_round(scale_y);
    const bool is_area_fast = fabs(scale_x - iscale_x) < DBL_EPSILON && fabs(scale_y - iscale_y) < DBL_EPSILON;
    t.xofs.resize(dw);
    t.yofs.resize(dh);
    t.alpha.resize(2 * (size_t)dw);
    t.beta.resize(2 * (size_t)dh);
    t.xmax = dw;
    if (sw == dw && sh == dh) {
        // scale 1: fx = fy = 0 everywhere, the generic bilinear code copies exactly
        t.mode = 0;
        for (int x = 0; x < dw; x++) { t.xofs[x] = x; t.alpha[2 * x] = 1.f; t.alpha[2 * x + 1] = 0.f; }
        for (int y = 0; y < dh; y++) { t.yofs[y] = y; t.beta[2 * y] = 1.f; t.beta[2 * y + 1] = 0.f; }
        return;
    }
    if (is_area_fast && iscale_x == 2 && iscale_y == 2) {
        // "INTER_AREA (fast) also is equal to INTER_LINEAR" branch: sum of the 2x2 block * 0.25f
        t.mode = 2;
        for (int x = 0; x < dw; x++) { t.xofs[x] = 2 * x; t.alpha[2 * x] = 0.5f; t.alpha[2 * x + 1] = 0.5f; }
        for (int y = 0; y < dh; y++) { t.yofs[y] = 2 * y; t.beta[2 * y] = 0.5f; t.beta[2 * y + 1] = 0.5f; }
        return;
    }
    t.mode = 1;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            if (dx < t.xmax) t.xmax = dx;
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        t.xofs[dx] = sx;
        t.alpha[2 * dx] = 1.f - fx;
        t.alpha[2 * dx + 1] = fx;
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        t.yofs[dy] = sy;  // rows are clipped at use, weights are not (resizeGeneric_Invoker)
        t.beta[2 * dy] = 1.f - fy;
        t.beta[2 * dy + 1] = fy;
    }
}

// FarnebackPolyExp constants — video/optflowgf.cpp; invG = G.inv(DECOMP_CHOLESKY) — core/lapack.cpp
void polyexp_setup(int n, double sigma, PolyCoef& pc)
{
    std::vector<float> gb(2 * n + 1), xgb(2 * n + 1), xxgb(2 * n + 1);
    float *g = gb.data() + n, *xg = xgb.data() + n, *xxg = xxgb.data() + n;
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
    double G[6][6] = {};
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
    double L[6][6], B[6][6] = {};
    memcpy(L, G, sizeof(L));
    for (int i = 0; i < 6; i++) B[i][i] = 1;
    for (int i = 0; i < 6; i++) {
        int j, k;
        double t;
        for (j = 0; j < i; j++) {
            t = L[i][j];
            for (k = 0; k < j; k++) t -= L[i][k] * L[j][k];
            L[i][j] = t * L[j][j];
        }
        t = L[i][i];
        for (k = 0; k < j; k++) {
            const double u = L[i][k];
            t -= u * u;
        }
        L[i][i] = 1. / sqrt(t);
    }
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            double t = B[i][j];
            for (int k = 0; k < i; k++) t -= L[i][k] * B[k][j];
            B[i][j] = t * L[i][i];
        }
    for (int i = 5; i >= 0; i--)
        for (int j = 0; j < 6; j++) {
            double t = B[i][j];
            for (int k = 5; k > i; k--) t -= L[k][i] * B[k][j];
            B[i][j] = t * L[i][i];
        }
    memset(&pc, 0, sizeof(pc));
    for (int k = 0; k <= n && k < 8; k++) {
        pc.g[k] = g[k];
        pc.xg[k] = xg[k];
        pc.xxg[k] = xxg[k];
        pc.gd[k] = (double)g[k];
        pc.xxgd[k] = (double)xxg[k];
    }
    pc.ig11 = B[1][1];
    pc.ig03 = B[0][3];
    pc.ig33 = B[3][3];
    pc.ig55 = B[5][5];
}

// window taps of FarnebackUpdateFlow_GaussianBlur
void window_kernel(int block_size, WinCoef& wc)
{
    const int m = block_size / 2;
    const double sigma = m * 0.3;
    double s = 1;
    memset(&wc, 0, sizeof(wc));
    wc.k[0] = (float)s;
    for (int i = 1; i <= m; i++) {
        const float t = (float)exp(-i * i / (2 * sigma * sigma));
        wc.k[i] = t;
        s += t * 2;
    }
    s = 1. / s;
    for (int i = 0; i <= m; i++) wc.k[i] = (float)(wc.k[i] * s);
}

// ---------------------------------------------------------------------------------------------------
// Plan: everything that depends only on (w0, h0, params)
// ---------------------------------------------------------------------------------------------------
struct LevelPlan {
    int w = 0, h = 0, ld = 0;
    long long ps = 0;  // plane stride (elements)
    double sigma = 0, scale = 1;
    int ksize = 3;
    int chunk = 1;  // image pairs per launch at this level (sized to fill the chip)
    // pyramid tables (full-res -> level)
    int *d_xofs = nullptr, *d_yofs = nullptr;
    float *d_alpha = nullptr, *d_beta = nullptr, *d_kern = nullptr;
    int mode = 1, xmax = 0, nrows_max = 0;
    float h_kern[3] = {0, 0, 0};  // host copy of the taps when ksize == 3
    std::vector<float> h_taps;    // host copy of all taps
    int pitch_b = 0;              // staged-row pitch (bytes) of tw_pyr_level_lds, 0: does not fit LDS
    int gen_th = PYR_TH, gen_nrows_max = 0;  // tile height of the UNSTAGED kernel (tw_pyr_level) and its row-buffer height
    // flow upsample tables (level k+1 -> k)
    int *d_uxofs = nullptr, *d_uyofs = nullptr;
    float *d_ualpha = nullptr, *d_ubeta = nullptr;
    int uxmax = 0;
};

struct Plan {
    int w0 = 0, h0 = 0;
    // levels 2 and 3 come from ONE launch of tw_pyr_23 (exact halving pyramid, 9- / 19-tap smoothing): set by get_plan
    // when the resize tables say so; k19 / k9 = [0, taps..., 0...] as the kernel's shared byte windows want them
    bool fused23 = false;
    float k19[24] = {0}, k9[16] = {0};
    int levels = 0;  // index of the coarsest level
    std::vector<LevelPlan> lv;
    std::vector<void*> owned;  // device allocations
    size_t owned_bytes = 0;    // ... and their total size (tw_debug_memory)
    unsigned long long last_use = 0;  // engine's plan clock at the last get_plan (LRU eviction)
};

constexpr int HOST_RECS = 1024;  // flagged vectors per pair copied back eagerly
constexpr int NCTX = 3;          // batches that may be in flight (host-side state only)

struct Job {
    const uint8_t *h_a = nullptr, *h_b = nullptr;  // host inputs already staged (null for device inputs)
    const uint8_t *d_a = nullptr, *d_b = nullptr;
    long long stride = 0;
    bool waited = false;
    int png_ch[2] = {0, 0};  // tw_submit_png8: channels of the filtered rows staged in Ctx::d_filt (0: plain gray)
};

// One batch of pairs: host-side state that must outlive the asynchronous execution.
struct Ctx {
    hipEvent_t ev_start = nullptr, ev_stop = nullptr, ev_done = nullptr;
    std::vector<Job> jobs;
    int64_t first_ticket = 0;
    bool launched = false;
    int pending = 0;  // jobs not yet waited
    int w = 0, h = 0, span = 0;
    double threshold = 0;
    bool any_host = false;
    // device copy of the batch's host images (one region per context: the next batch uploads while this one
    // computes) and the event that marks its uploads complete on the copy stream
    uint8_t* d_img = nullptr;
    size_t d_img_cap = 0;
    hipEvent_t ev_h2d = nullptr;
    // cold-start ramp (round 6): the copy stream is marked when a quarter and a half of a full batch's uploads have been
    // queued (submit_common); a batch that is launched into an IDLE compute stream goes out in up to three pieces, each
    // behind its own mark, instead of waiting for the last upload of the whole batch (flush_ctx)
    hipEvent_t ev_seg[2] = {nullptr, nullptr};
    int seg_at[2] = {0, 0};  // pairs queued when the mark was recorded
    int nseg = 0;
    // tw_submit_png8: filtered PNG rows of the batch (2 slots of filt_slot bytes per job) and the job table of
    // tw_png_unfilter, which reconstructs them into d_img on the copy stream before ev_h2d
    uint8_t *d_filt = nullptr, *d_filt_raw = nullptr;  // d_filt = d_filt_raw + 256 (slack on both sides)
    size_t d_filt_cap = 0, filt_slot = 0;  // filt_slot: bytes per image of the open batch (0: no filtered image yet)
    PngJob* d_png = nullptr;
    PngJob* h_png = nullptr;  // pinned, [2*cap]
    bool any_png = false;
    long long copy_ops_at_flush = 0;  // tw_engine::copy_ops when this batch was launched
    // ordered hit records of the batch [cap][G]: tw_wait reads them late when a pair has more than HOST_RECS hits,
    // so they belong to the context, not to the engine (a later batch must not overwrite them)
    ScanRec* d_rec = nullptr;
    size_t d_rec_cap = 0;
    // pinned host memory
    uint8_t* h_img = nullptr;
    size_t h_img_cap = 0;
    const uint8_t** h_ptrs = nullptr;  // [2*cap]
    int* h_count = nullptr;            // [cap]
    ScanRec* h_rec = nullptr;          // [cap][HOST_RECS]
};

struct ProfPair {
    hipEvent_t a, b;
};

// what a captured single-pair schedule depends on besides the engine's (fixed) parameters and workspace addresses
struct GraphKey {
    int w, h, span;
    long long stride;
    int aligned4, scan_fused, poly_f32;
    bool operator<(const GraphKey& o) const
    {
        return std::tie(w, h, span, stride, aligned4, scan_fused, poly_f32) <
               std::tie(o.w, o.h, o.span, o.stride, o.aligned4, o.scan_fused, o.poly_f32);
    }
};

// Optional roctx ranges (TW_ROCTX=1): one range per batch and per pyramid level on the submitting thread, so that a
// rocprofv3 --marker-trace timeline shows which launches belong to which batch / level (SURVEY §5 "tracing"; the
// reference only times the call, which `seconds` preserves).  The roctx library is loaded on demand: no link dependency.
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        const char* ev = getenv("TW_ROCTX");
        if (!ev || !atoi(ev)) return;
        // rocprofv3 (rocprofiler-sdk) listens to its own roctx library; roctracer's libroctx64 is the fallback
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        pop = (int (*)())dlsym(h, "roctxRangePop");
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
Roctx& roctx()
{
    static Roctx r;
    return r;
}
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char* fmt, int a = 0, int b = 0, int c = 0) : on(roctx().push != nullptr)
    {
        if (on) {
            char buf[96];
            snprintf(buf, sizeof(buf), fmt, a, b, c);
            roctx().push(buf);
        }
    }
    ~RoctxRange()
    {
        if (on) roctx().pop();
    }
};

}  // namespace

struct tw_engine {
    int device = 0;
    tw_params p;
    int cap = 1;  // pairs per batch ("slots")
    hipStream_t stream = nullptr;
    Ctx ctx[NCTX];
    int cur = 0;  // context accepting submissions
    int64_t next_ticket = 1;
    std::map<std::pair<int, int>, Plan*> plans;
    unsigned long long plan_clock = 0;
    PolyCoef pc;
    WinCoef wc;
    int win_m = 15;
    int box = 0;           // flags without 256: box window (FarnebackUpdateFlow_Blur), scan kernels
    double* Vd = nullptr;  // running column sums of the box window (5 double planes per pair of a chunk)
    int copy_sync_skipped = 0;    // batches since the host last synchronised the copy stream (tw_wait)
    long long copy_ops = 0;       // host images queued on the copy stream so far
    uint8_t* h_bounce = nullptr;  // pinned bounce buffer of tw_submit_png8's pageable inputs
    size_t h_bounce_cap = 0;
    size_t Vd_cap = 0;
    int img_aligned4 = 0;  // every image pointer and the row stride of the batch being enqueued are 4-byte aligned
    int lanes = 1;         // TW_LANES=2: the two halves of a batch run on two streams (memory-bound kernels of one
                           // half overlap the VALU-bound blur of the other); per-kernel hipEvent durations then
                           // include the co-running kernel
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int upd_ny = 2;        // TW_UPD_NY: pixels per lane of tw_update_matrices (1 or 2)
    int scan_fused = 0;    // TW_OPT_SCAN_FUSED_FINAL
    int poly_f32 = 0;      // TW_OPT_POLYEXP_F32 (measurement variant: float accumulators, not bit-exact)
    long long mfree_min_px = 0;  // TW_MFREE_MIN_PX: smallest level (pixels) that takes tw_flow_iter
    int mfree_min_w = 320;       // TW_MFREE_MIN_W: narrowest level (columns; strips are 160 outputs wide) that takes it
    int mfree = 1;         // TW_MFREE=0: every level runs update + blur launches again (A/B; tw_flow_iter, no M in HBM, is the
                           // default); 2: tw_flow_iter also for launches of a few workgroups (tests, tools/fuzz_parity.py)
    int cu_count = 256;
    int pyr23_threads = 0;    // TW_PYR23_THREADS=256 / 512 / 1024: waves per tw_pyr_23 workgroup (A/B); 0 = 512, and 1024 for a
                              // single pair (432 workgroups: sixteen waves cover the staging latency, -0.5 us of 10.6)
    int blur_cm = 0;       // TW_BLUR_CM=1: tw_blur_solve4y (51-tap window) walks an XCD's tiles column-major (A/B)
    int pyr_fused = 1;     // TW_PYR_FUSED=0: levels 2 and 3 as two tw_pyr_taps launches again (A/B; tw_pyr_23 is the default)
    int pyr_generic = 0;   // TW_PYR_GENERIC=1: always the generic pyramid kernel, 2: tw_pyr_level_lds for every level (A/B)
    int blur_variant = 4;  // 4: tw_blur_solve4 (default); 8: tw_blur_solve8 (packed f32) — TW_BLUR_VARIANT
    int pp_waves = 512;    // TW_PP_WAVES: below this many waves of 96x8 tiles a level takes the plane-parallel kernel
    unsigned long long* dbg_stamps = nullptr;  // TW_DEBUG_STAMPS=1: phase stamps of tw_pyr_taps (diagnostic runs only)
    int blur_nomask = 0;   // TW_BLUR_NOMASK
    int blur_pipe = 0;     // TW_BLUR_PIPE: tiles a workgroup of tw_blur_solve4p walks (0: tw_blur_solve4 for every launch)
    int blur_small = -1;   // TW_BLUR_SMALL: force the small-grid tile choice of the 31-tap blur (-1: by grid size)
    int blur_small_lv[8] = {-1, -1, -1, -1, -1, -1, -1, -1};  // TW_BLUR_SMALL_LEVELS="a,b,c,d": the same per pyramid level
    int poly_variant = 1;  // 1: tw_polyexp_pk<N,8> (packed f32, default); 2: tw_polyexp_pk<N,16>; 0: tw_polyexp (scalar f32) — TW_POLY_VARIANT
    std::string err;
    // device workspace, shared by all batches (execution is ordered on one stream)
    size_t ws_elems = 0;                 // capacity of I (floats); R = 5x, M = 5x each
    float *I = nullptr, *R = nullptr, *M[2] = {nullptr, nullptr};
    std::vector<float*> flow;            // per level, cap (levels>=1) or chunk0 (level 0) pairs x 2 planes
    std::vector<size_t> flow_cap;
    // single-pair latency schedule (BASELINE config 2): a batch of one pair runs the pyramid + polynomial expansion of
    // every level on stream2 (they depend only on the images) while the main stream walks the flow chain coarse to
    // fine; needs every level's I and R at once (TW_LATENCY_STREAMS=0: everything on one stream)
    int lat_streams = 1;
    long long lat_min_px = 100000;  // TW_LATENCY_MIN_PX
    float *lat_I = nullptr, *lat_R = nullptr;
    size_t lat_cap = 0;  // floats in lat_I (lat_R holds 5x)
    std::vector<hipEvent_t> lat_ev;
    int lat_s2_max = 1000;  // TW_LAT_S2_LEVELS: finest..this level's image-only work goes to the second stream
    int ramp = 1;  // TW_RAMP=0: no cold-start ramp (A/B)
    int fi_maxseg = 4, fi_minsteps = 16;  // TW_FI_MAXSEG / TW_FI_MINSTEPS: most row segments per strip / fewest steps per segment (A/B)
    int fi_skip = 0;    // TW_FI_SKIP (variants library): timing experiments on tw_flow_iter (results are wrong)
    int fi_nt = 1024;   // TW_FI_NT=512 (variants library): tw_flow_iter as two 512-thread workgroups per CU on 64-output strips (A/B)
    int lat_fused = 1;  // TW_LAT_FUSED=0: the two-stream single-pair schedule of rounds 2-4 instead of the twin launches (A/B)
    int lat_graph = 0;  // TW_LAT_GRAPH=1: replay the single-pair schedule from a captured hipGraph (measured SLOWER on
                        // ROCm 7.2: 0.55 ms vs 0.37 ms — profiles/r03_latency.md — so it is opt-in, kept for re-measuring)
    std::map<GraphKey, hipGraphExec_t> lat_graphs;  // captured single-pair schedules (dropped whenever a buffer moves)
    hipStream_t copy_stream = nullptr;  // host -> device image uploads, overlapped with the compute stream
    const uint8_t** d_ptrs = nullptr;  // [2*cap]
    int* d_count = nullptr;            // [cap]
    float2* d_grid = nullptr;          // [cap][G] dense grid samples (dx,dy)
    size_t d_grid_cap = 0;
    // profiling: per kernel class, -2 = off, -1 = every level, k = level k only
    int prof_level[TW_K_COUNT] = {-2, -2, -2, -2, -2};
    std::vector<ProfPair> prof_pending[TW_K_COUNT];
    std::vector<hipEvent_t> prof_free;
    // diagnostics (include/twflow_debug.h: tw_debug_launch_counts): launches per kernel family since the engine was created /
    // the counts were last reset, and the pairs (grid z) of each family's latest launch — what lets a test prove WHICH kernel ran
    unsigned long long launches[TW_DF_COUNT] = {};
    unsigned long long last_grid_z[TW_DF_COUNT] = {};
};

namespace {

// every kernel launch of the library goes through here (tw_copy_f4, the copy-rate yardstick, excepted)
#define TW_LAUNCH(e, fam, kern, grid, ...)                         \
    do {                                                           \
        const dim3 _g = (grid);                                    \
        if (e) {                                                   \
            (e)->launches[fam]++;                                  \
            (e)->last_grid_z[fam] = _g.z;                          \
        }                                                          \
        hipLaunchKernelGGL(kern, _g, __VA_ARGS__);                 \
    } while (0)

#define TW_HIP(e, call)                                                                   \
    do {                                                                                  \
        hipError_t _err = (call);                                                         \
        if (_err != hipSuccess) {                                                         \
            (void)hipGetLastError(); /* the failure is reported here: do not leave it for a later launch check */ \
            (e)->err = std::string(#call) + ": " + hipGetErrorString(_err);               \
            return _err == hipErrorOutOfMemory ? TW_E_NOMEM : TW_E_DEVICE;                \
        }                                                                                 \
    } while (0)

size_t staged_image_bytes(int w, int h);

// ---- the runtime is never handed a pageable host pointer ------------------------------------------------
// Round 4 recorded a GPU memory-access fault on a malloc-heap address inside a synchronous copy, right after a
// hipHostRegister / hipHostUnregister cycle on a neighbouring, non-page-aligned heap block: for a
// pageable operand the runtime pins the enclosing pages on the fly, and that pinning raced with the pages a partial-page
// registration had just released.  Two rules close it at the source: tw_host_register only takes whole pages the
// caller owns (below), and EVERY synchronous copy of this library whose host side is not a page-locked range the
// library knows goes through the engine's own page-locked bounce buffer, in bounded chunks.
bool host_range_is_page_locked(const void* p, size_t n);  // PinRegistry::covers (defined below)
constexpr size_t BOUNCE_CHUNK = 8u << 20;
tw_status bounce_reserve(tw_engine* e, size_t bytes)
{
    if (bytes <= e->h_bounce_cap) return TW_OK;
    if (e->h_bounce) (void)hipHostFree(e->h_bounce);
    e->h_bounce = nullptr;
    e->h_bounce_cap = 0;
    TW_HIP(e, hipHostMalloc((void**)&e->h_bounce, bytes, hipHostMallocDefault));
    e->h_bounce_cap = bytes;
    return TW_OK;
}
// synchronous host -> device / device -> host copies of any host pointer (null stream, like the hipMemcpy they replace)
tw_status h2d_sync(tw_engine* e, void* d, const void* h, size_t n)
{
    if (!n) return TW_OK;
    if (host_range_is_page_locked(h, n)) {
        TW_HIP(e, hipMemcpy(d, h, n, hipMemcpyHostToDevice));
        return TW_OK;
    }
    tw_status r = bounce_reserve(e, std::min(n, BOUNCE_CHUNK));
    if (r) return r;
    for (size_t o = 0; o < n; o += BOUNCE_CHUNK) {
        const size_t k = std::min(BOUNCE_CHUNK, n - o);
        memcpy(e->h_bounce, (const uint8_t*)h + o, k);
        TW_HIP(e, hipMemcpy((uint8_t*)d + o, e->h_bounce, k, hipMemcpyHostToDevice));
    }
    return TW_OK;
}
tw_status d2h_sync(tw_engine* e, void* h, const void* d, size_t n)
{
    if (!n) return TW_OK;
    if (host_range_is_page_locked(h, n)) {
        TW_HIP(e, hipMemcpy(h, d, n, hipMemcpyDeviceToHost));
        return TW_OK;
    }
    tw_status r = bounce_reserve(e, std::min(n, BOUNCE_CHUNK));
    if (r) return r;
    for (size_t o = 0; o < n; o += BOUNCE_CHUNK) {
        const size_t k = std::min(BOUNCE_CHUNK, n - o);
        TW_HIP(e, hipMemcpy(e->h_bounce, (const uint8_t*)d + o, k, hipMemcpyDeviceToHost));
        memcpy((uint8_t*)h + o, e->h_bounce, k);
    }
    return TW_OK;
}
#define TW_TRY(call)                      \
    do {                                  \
        const tw_status _st = (call);     \
        if (_st != TW_OK) return _st;     \
    } while (0)

// scoped device allocations (per-stage test entry points, slow paths)
struct Tmp {
    std::vector<void*> v;
    ~Tmp()
    {
        for (void* p : v) (void)hipFree(p);
    }
    template <typename T>
    T* alloc(size_t n)
    {
        void* p = nullptr;
        if (hipMalloc(&p, n * sizeof(T) + 256) != hipSuccess) return nullptr;
        v.push_back(p);
        return (T*)p;
    }
};
template <typename T>
tw_status upload_vec(tw_engine* e, Plan* pl, const std::vector<T>& v, T** out)
{
    void* d = nullptr;
    TW_HIP(e, hipMalloc(&d, v.size() * sizeof(T) + 16));
    pl->owned.push_back(d);
    pl->owned_bytes += v.size() * sizeof(T) + 16;
    TW_TRY(h2d_sync(e, d, v.data(), v.size() * sizeof(T)));
    *out = (T*)d;
    return TW_OK;
}

int plan_levels(int w0, int h0, double pyr_scale, int levels)
{
    const int min_size = 32;
    int k;
    double scale;
    for (k = 0, scale = 1; k < levels; k++) {
        scale *= pyr_scale;
        if (w0 * scale < min_size || h0 * scale < min_size) break;
    }
    return k;
}

void level_geometry(int w0, int h0, double pyr_scale, int k, int* w, int* h, double* sigma, int* ksize,
                    double* scale_out)
{
    double scale = 1;
    for (int i = 0; i < k; i++) scale *= pyr_scale;
    const double sg = (1. / scale - 1) * 0.5;
    int smooth_sz = cv_round(sg * 5) | 1;
    if (smooth_sz < 3) smooth_sz = 3;
    *w = cv_round(w0 * scale);
    *h = cv_round(h0 * scale);
    *sigma = sg;
    *ksize = smooth_sz;
    *scale_out = scale;
}

int pyr_nrows_max(const ResizeTab& t, int h, int h0, int r, int th = PYR_TH)
{
    int mx = 0;
    for (int y0 = 0; y0 < h; y0 += th) {
        const int yB = std::min(y0 + th - 1, h - 1);
        const int ylo = std::min(std::max(t.yofs[y0], 0), h0 - 1) - r;
        const int yhi = std::min(std::max(t.yofs[yB] + 1, 0), h0 - 1) + r;
        mx = std::max(mx, yhi - ylo + 1);
    }
    return mx;
}

// captured schedules hold workspace and table addresses: whenever one of those may move, they go
void drop_graphs(tw_engine* e)
{
    for (auto& kv : e->lat_graphs) (void)hipGraphExecDestroy(kv.second);
    e->lat_graphs.clear();
}

void free_plan(Plan* pl)
{
    for (void* d : pl->owned) (void)hipFree(d);
    delete pl;
}

// pairs one launch covers at a level of w x h pixels (level index k): enough 8-row x 240-column tiles of two images to cover
// the 256 CUs many times over.  One rule for the plan and for the byte model (tw_algorithmic_bytes_launch).
int pairs_per_launch(const tw_engine* e, int w, int h, int k)
{
    const long long tiles = (long long)((w + PE_TW - 1) / PE_TW) * ((h + PE_TH - 1) / PE_TH) * 2;
    // measured at 1080p: 4096 -> 16384 tiles per launch is +11 % pairs/s (fewer tails and gaps), 16384 -> 34000
    // (16 pairs per level-0 launch) another +2.5 %, 34000 -> a whole 64-pair batch per launch +1.5 %, a whole
    // 128-pair batch +4 % (round 2): every kernel boundary costs a drain, a fill and an L2 write-back of the dirty
    // M planes, and 288 GB of HBM make the workspace of a whole batch (23 GB at 128 x 1080p) a non-issue.  So a
    // launch covers the whole batch unless that exceeds 300 000 tiles.
    long long target = 300000;
    if (const char* ev = getenv("TW_CHUNK_TILES")) target = std::max(1, atoi(ev));
    long long c = (target + tiles - 1) / tiles;
    // TW_CHUNK_PAIRS_LEVELS="p0,p1,...": pairs per launch at level 0, 1, ... (0 / missing: the rule above) — the
    // round-4 Infinity-Cache experiment (does a level's chain run faster when a chunk's R / M planes fit the 256 MB
    // memory-side cache between its kernels?  profiles/r04_mall_chunks.md)
    if (const char* ev = getenv("TW_CHUNK_PAIRS_LEVELS")) {
        const char* q = ev;
        for (int lv = 0; lv < k && q; lv++) {
            q = strchr(q, ',');
            if (q) q++;
        }
        if (q && atoi(q) > 0) c = atoi(q);
    }
    return (int)std::min<long long>(std::max<long long>(c, 1), e->cap);
}

tw_status get_plan(tw_engine* e, int w0, int h0, Plan** out)
{
    auto key = std::make_pair(w0, h0);
    auto it = e->plans.find(key);
    if (it != e->plans.end()) {
        it->second->last_use = ++e->plan_clock;
        *out = it->second;
        return TW_OK;
    }
    if (e->plans.size() >= 64) {
        // a long-running service that sees arbitrary image sizes must not collect plans (and their device tables) for
        // ever: the 65th size evicts the least recently used one (emptying the whole cache made a service
        // that cycles through 65 sizes rebuild all of them on every wrap).  Queued launches may still read the evicted
        // plan's tables, so the streams are drained first — once per new size, and only beyond 64 sizes.
        auto victim = e->plans.begin();
        for (auto jt = e->plans.begin(); jt != e->plans.end(); ++jt)
            if (jt->second->last_use < victim->second->last_use) victim = jt;
        TW_HIP(e, hipStreamSynchronize(e->stream));
        TW_HIP(e, hipStreamSynchronize(e->stream2));
        drop_graphs(e);
        free_plan(victim->second);
        e->plans.erase(victim);
    }
    Plan* pl = new Plan();
    pl->w0 = w0;
    pl->h0 = h0;
    pl->last_use = ++e->plan_clock;
    const int req = std::min(std::max(e->p.pyrLevels, 0), 60);
    pl->levels = plan_levels(w0, h0, e->p.pyrScale, req);
    pl->lv.resize(pl->levels + 1);
    for (int k = 0; k <= pl->levels; k++) {
        LevelPlan& L = pl->lv[k];
        level_geometry(w0, h0, e->p.pyrScale, k, &L.w, &L.h, &L.sigma, &L.ksize, &L.scale);
        if (L.w < 1 || L.h < 1 || L.ksize > PYR_MAXK - 1) {
            e->err = "pyramid level degenerate or smoothing kernel too large";
            free_plan(pl);
            return TW_E_UNSUPPORTED;
        }
        L.ld = round_up(L.w, 32);
        L.ps = (long long)L.ld * L.h;
        L.chunk = pairs_per_launch(e, L.w, L.h, k);
    }
    for (int k = 0; k <= pl->levels; k++) {
        LevelPlan& L = pl->lv[k];
        ResizeTab t;
        make_resize_tab(w0, h0, L.w, L.h, t);
        std::vector<float> kern;
        gaussian_kernel(L.ksize, L.sigma, kern);
        L.mode = t.mode;
        L.xmax = t.xmax;
        if (L.ksize == 3) memcpy(L.h_kern, kern.data(), sizeof(L.h_kern));
        L.h_taps = kern;
        L.nrows_max = pyr_nrows_max(t, L.h, h0, L.ksize / 2);
        // Round 6: the unstaged kernel takes 4-row tiles when an 8-row tile's row buffer is more than 32 KB (a deep level of a
        // large image: BASELINE configs[4]'s 39- / 79-tap levels of a 4K pyramid).  Twice the workgroups at 0.6 of the work
        // each, five / three of them per CU instead of three / two, and the last round of a launch no longer a quarter full
        // (TW_PYR_GEN_TH=8: the old tiles, A/B).
        {
            static const int gen_th_env = getenv("TW_PYR_GEN_TH") ? atoi(getenv("TW_PYR_GEN_TH")) : 0;
            const size_t rb8 = (size_t)L.nrows_max * (t.mode == 0 ? PYR_TW : 2 * PYR_TW) * 4;
            L.gen_th = gen_th_env >= 1 && gen_th_env <= PYR_TH ? gen_th_env : (rb8 > 32 * 1024 ? 4 : PYR_TH);
            L.gen_nrows_max = L.gen_th == PYR_TH ? L.nrows_max : pyr_nrows_max(t, L.h, h0, L.ksize / 2, L.gen_th);
        }
        const size_t lds = (size_t)(PYR_MAXK + (size_t)L.nrows_max * (t.mode == 0 ? PYR_TW : 2 * PYR_TW)) * 4;
        if (lds > 160 * 1024) {
            e->err = "pyramid tile does not fit LDS (scale/kernel too large)";
            free_plan(pl);
            return TW_E_UNSUPPORTED;
        }
        {
            // widest staged source span over the tiles of this level (same formulas as the kernel)
            const int r = L.ksize / 2;
            int span = 0;
            for (int x0 = 0; x0 < L.w; x0 += PYR_TW) {
                const int xl = std::min(x0, L.w - 1), xr = std::min(x0 + PYR_TW - 1, L.w - 1);
                const int Xfirst = t.mode == 0 ? xl : t.xofs[xl];
                const int Xlast = t.mode == 0 ? xr : t.xofs[xr] + 1;
                const int xlo_a = (Xfirst - r) & ~3;
                span = std::max(span, ((Xlast + r - xlo_a) / 4 + 1) * 4);
            }
            const size_t lds2 = lds + (size_t)L.nrows_max * span;
            L.pitch_b = (lds2 + (size_t)L.nrows_max * 8 + 64 <= 64 * 1024 && L.ksize <= 63) ? span : 0;
        }

        tw_status s;
        if ((s = upload_vec(e, pl, t.xofs, &L.d_xofs)) || (s = upload_vec(e, pl, t.yofs, &L.d_yofs)) ||
            (s = upload_vec(e, pl, t.alpha, &L.d_alpha)) || (s = upload_vec(e, pl, t.beta, &L.d_beta)) ||
            (s = upload_vec(e, pl, kern, &L.d_kern))) {
            free_plan(pl);
            return s;
        }
        if (k < pl->levels) {
            const LevelPlan& P = pl->lv[k + 1];
            ResizeTab u;
            make_resize_tab(P.w, P.h, L.w, L.h, u);
            if (u.mode == 2) {
                // cannot happen for pyr_scale < 1 (the previous level is never twice as large)
                e->err = "unexpected area-fast flow resize";
                free_plan(pl);
                return TW_E_UNSUPPORTED;
            }
            L.uxmax = u.xmax;
            if ((s = upload_vec(e, pl, u.xofs, &L.d_uxofs)) || (s = upload_vec(e, pl, u.yofs, &L.d_uyofs)) ||
                (s = upload_vec(e, pl, u.alpha, &L.d_ualpha)) || (s = upload_vec(e, pl, u.beta, &L.d_ubeta))) {
                free_plan(pl);
                return s;
            }
        }
    }
    // tw_pyr_23 eligibility: both levels are exact INTER_LINEAR reductions by 4 and 8 (sample columns s*x + s/2 - 1 and
    // + 1 with weights 0.5 / 0.5, no single-tap tail, rows alike) with the 9- / 19-tap kernels of pyr_scale = 0.5
    if (pl->levels >= 3 && pl->lv[2].ksize == 9 && pl->lv[3].ksize == 19 && pl->lv[2].w == 2 * pl->lv[3].w &&
        pl->lv[2].h == 2 * pl->lv[3].h && w0 >= 64 && h0 >= 64) {
        bool exact = true;
        for (int k = 2; k <= 3 && exact; k++) {
            const LevelPlan& L = pl->lv[k];
            ResizeTab t;
            make_resize_tab(w0, h0, L.w, L.h, t);
            const int sc = 1 << k;
            exact = t.mode == 1 && t.xmax == L.w && L.w * sc == w0 && L.h * sc == h0;
            for (int x = 0; x < L.w && exact; x++)
                exact = t.xofs[x] == sc * x + sc / 2 - 1 && t.alpha[2 * x] == 0.5f && t.alpha[2 * x + 1] == 0.5f;
            for (int y = 0; y < L.h && exact; y++)
                exact = t.yofs[y] == sc * y + sc / 2 - 1 && t.beta[2 * y] == 0.5f && t.beta[2 * y + 1] == 0.5f;
        }
        if (exact) {
            pl->fused23 = true;
            memcpy(pl->k19 + 1, pl->lv[3].h_taps.data(), sizeof(float) * 19);
            memcpy(pl->k9 + 1, pl->lv[2].h_taps.data(), sizeof(float) * 9);
        }
    }
    e->plans[key] = pl;
    *out = pl;
    return TW_OK;
}

// Device workspace for a plan (grown, never shrunk).  Safe to reallocate only when nothing is queued.
tw_status reserve_workspace(tw_engine* e, const Plan* pl, int span, bool need_img)
{
    size_t need = 0;
    for (const LevelPlan& L : pl->lv) need = std::max(need, (size_t)L.ps * 2 * L.chunk);
    need *= e->lanes;  // one workspace per lane, carved from the same allocations
    bool grow = need > e->ws_elems || e->flow.size() < pl->lv.size() || (e->box && need / 2 * 5 > e->Vd_cap);
    if (!grow)
        for (size_t k = 0; k < pl->lv.size(); k++) {
            const size_t fc = (size_t)pl->lv[k].ps * 2 * (k == 0 ? pl->lv[0].chunk * e->lanes : e->cap);
            if (fc > e->flow_cap[k]) grow = true;
        }
    (void)need_img;  // image regions belong to the batch contexts (submit_common)
    const size_t G = span > 0 ? (size_t)tw_grid_capacity(pl->w0, pl->h0, span) : 0;
    if (G * e->cap > e->d_grid_cap) grow = true;
    size_t need_lat = 0;
    for (const LevelPlan& L : pl->lv) need_lat += (size_t)L.ps * 2;
    if (e->lat_streams && (need_lat > e->lat_cap || e->lat_ev.size() < pl->lv.size())) grow = true;
    if (!grow) return TW_OK;
    TW_HIP(e, hipStreamSynchronize(e->stream));
    TW_HIP(e, hipStreamSynchronize(e->stream2));
    drop_graphs(e);
    if (need > e->ws_elems) {
        if (e->I) (void)hipFree(e->I);
        if (e->R) (void)hipFree(e->R);
        if (e->M[0]) (void)hipFree(e->M[0]);
        if (e->M[1]) (void)hipFree(e->M[1]);
        e->I = e->R = e->M[0] = e->M[1] = nullptr;
        e->ws_elems = 0;
        TW_HIP(e, hipMalloc((void**)&e->I, need * 4 + 256));
        TW_HIP(e, hipMalloc((void**)&e->R, need * 5 * 4 + 256));
        TW_HIP(e, hipMalloc((void**)&e->M[0], need / 2 * 5 * 4 + 256));
        TW_HIP(e, hipMalloc((void**)&e->M[1], need / 2 * 5 * 4 + 256));
        e->ws_elems = need;
    }
    if (e->lat_streams && need_lat > e->lat_cap) {
        if (e->lat_I) (void)hipFree(e->lat_I);
        if (e->lat_R) (void)hipFree(e->lat_R);
        e->lat_I = e->lat_R = nullptr;
        e->lat_cap = 0;
        TW_HIP(e, hipMalloc((void**)&e->lat_I, need_lat * 4 + 256));
        TW_HIP(e, hipMalloc((void**)&e->lat_R, need_lat * 5 * 4 + 256));
        e->lat_cap = need_lat;
    }
    while (e->lat_streams && e->lat_ev.size() < pl->lv.size()) {
        hipEvent_t ev = nullptr;
        TW_HIP(e, hipEventCreateWithFlags(&ev, getenv("TW_LAT_EVTIMING") ? hipEventDefault : hipEventDisableTiming));
        e->lat_ev.push_back(ev);
    }
    if (e->box && need / 2 * 5 > e->Vd_cap) {
        if (e->Vd) (void)hipFree(e->Vd);
        e->Vd = nullptr;
        e->Vd_cap = 0;
        TW_HIP(e, hipMalloc((void**)&e->Vd, need / 2 * 5 * sizeof(double) + 256));
        e->Vd_cap = need / 2 * 5;
    }
    if (e->flow.size() < pl->lv.size()) {
        e->flow.resize(pl->lv.size(), nullptr);
        e->flow_cap.resize(pl->lv.size(), 0);
    }
    for (size_t k = 0; k < pl->lv.size(); k++) {
        const size_t fc = (size_t)pl->lv[k].ps * 2 * (k == 0 ? pl->lv[0].chunk * e->lanes : e->cap);
        if (fc > e->flow_cap[k]) {
            if (e->flow[k]) (void)hipFree(e->flow[k]);
            e->flow[k] = nullptr;
            e->flow_cap[k] = 0;
            TW_HIP(e, hipMalloc((void**)&e->flow[k], fc * 4 + 256));
            e->flow_cap[k] = fc;
        }
    }
    if (G * e->cap > e->d_grid_cap) {
        if (e->d_grid) (void)hipFree(e->d_grid);
        e->d_grid = nullptr;
        e->d_grid_cap = 0;
        TW_HIP(e, hipMalloc((void**)&e->d_grid, G * e->cap * sizeof(float2) + 256));
        e->d_grid_cap = G * e->cap;
    }
    return TW_OK;
}

// ---- profiling hooks -------------------------------------------------------------------------------
hipEvent_t prof_event(tw_engine* e)
{
    if (!e->prof_free.empty()) {
        hipEvent_t ev = e->prof_free.back();
        e->prof_free.pop_back();
        return ev;
    }
    hipEvent_t ev = nullptr;
    (void)hipEventCreate(&ev);
    return ev;
}

struct ProfScope {
    tw_engine* e;
    hipStream_t st;
    bool on;
    int kc;
    ProfPair pp;
    ProfScope(tw_engine* e_, hipStream_t st_, int kclass, int level) : e(e_), st(st_), kc(kclass)
    {
        on = (e->prof_level[kclass] == -1 || (e->prof_level[kclass] >= 0 && e->prof_level[kclass] == level));
        if (on) {
            pp.a = prof_event(e);
            pp.b = prof_event(e);
            (void)hipEventRecord(pp.a, st);
        }
    }
    ~ProfScope()
    {
        if (on) {
            (void)hipEventRecord(pp.b, st);
            e->prof_pending[kc].push_back(pp);
        }
    }
};

// A side job of the single-pair schedule: image-only work (a band of a level's polynomial expansion, or the level 0 / 1
// images) that rides in the same launch as a kernel of the dependent flow chain (the twin kernels of twflow_kernels.hip.h).
// A launcher that has a twin for its kernel sets *used; the caller launches the job by itself otherwise.
struct SideJob {
    int kind = 0;        // 1: polynomial-expansion band (tw_polyexp_pk<7, 8, 0>), 2: tw_pyr_k3f
    int need_level = 0;  // the finest level whose flow chain reads what the job writes
    int carry_level = -1;       // >= 0: only a chain launch of this level carries the job
    bool windows_only = false;  // ... and only its window launches, not its FarnebackUpdateMatrices launch
    PolyArgs pa;
    PyrK3fArgs ka;
    VGrid g;             // the job's workgroups
    unsigned n() const { return g.gx * g.gy * g.gz; }
};
TwinGrid make_twin(const dim3& ga, const SideJob& s)
{
    TwinGrid t;
    t.a = VGrid{ga.x, ga.y, ga.z, 0u, 0u};
    t.b = s.g;
    t.nA = ga.x * ga.y * ga.z;
    t.nA8 = (t.nA + 7u) & ~7u;
    return t;
}
void launch_side_alone(tw_engine* e, hipStream_t st, const SideJob& s)
{
    (void)e;
    if (s.kind == 1) TW_LAUNCH(e, TW_DF_POLYEXP, tw_polyexp_band, dim3(s.n()), dim3(256), 0, st, s.pa, s.g);
    else if (s.kind == 2) TW_LAUNCH(e, TW_DF_PYR_K3F, tw_pyr_k3f, dim3(s.g.gx, s.g.gy, s.g.gz), dim3(256), 0, st, s.ka);
}

// tw_update_matrices<UPSAMPLE, NY>: NY pixels of a column per lane (TW_UPD_NY, default 2)
template <bool UP>
void launch_upd_kernel(tw_engine* e, hipStream_t st, int w, int h, int npairs, const UpdArgs& a, const SideJob* side = nullptr,
                       bool* side_used = nullptr)
{
#ifdef TW_VARIANTS
    if (e->upd_ny == 1) {
        TW_LAUNCH(e, TW_DF_UPDATE_MATRICES, (tw_update_matrices<UP, 1>), dim3((w + 63) / 64, (h + 3) / 4, npairs), dim3(256), 0, st, a);
        return;
    }
#endif
    const dim3 grid((w + 63) / 64, (h + 7) / 8, npairs);
    if (side && side_used && side->kind == 1) {
        const TwinGrid t = make_twin(grid, *side);
        TW_LAUNCH(e, TW_DF_TWIN, tw_twin_upd_poly<UP>, dim3(t.nA8 + side->n()), dim3(256), 0, st, a, side->pa, t);
        *side_used = true;
        return;
    }
    TW_LAUNCH(e, TW_DF_UPDATE_MATRICES, (tw_update_matrices<UP, 2>), grid, dim3(256), 0, st, a);
}

// ---- kernel launch helpers (nz = images or pairs in this launch) -------------------------------------
void launch_pyr(tw_engine* e, hipStream_t st, const Plan* pl, int k, const uint8_t* const* d_srcs, long long stride,
                float* I, int nimg)
{
    const LevelPlan& L = pl->lv[k];
    PyrArgs a;
    a.srcs = d_srcs;
    a.dst = I;
    a.dst_zs = L.ps;
    a.stride = stride;
    a.w0 = pl->w0;
    a.h0 = pl->h0;
    a.w = L.w;
    a.h = L.h;
    a.ld = L.ld;
    a.xofs = L.d_xofs;
    a.alpha = L.d_alpha;
    a.yofs = L.d_yofs;
    a.beta = L.d_beta;
    a.kern = L.d_kern;
    a.ksize = L.ksize;
    a.mode = L.mode;
    a.xmax = L.xmax;
    a.nrows_max = L.nrows_max;
    a.th = PYR_TH;
    const int P = (L.mode == 0) ? PYR_TW : 2 * PYR_TW;
    const size_t lds = (size_t)(PYR_MAXK + (size_t)L.nrows_max * P) * 4;
    dim3 grid((L.w + PYR_TW - 1) / PYR_TW, (L.h + PYR_TH - 1) / PYR_TH, nimg);
    ProfScope ps(e, st, TW_K_PYR, k);
    if (L.ksize == 3 && (L.mode == 0 || L.mode == 2) && pl->w0 >= 16 && !e->pyr_generic) {
        // register-only fast path (3 taps, same-size or exact 2x2 area resize)
        PyrK3Args b;
        b.srcs = d_srcs;
        b.dst = I;
        b.dst_zs = L.ps;
        b.stride = stride;
        b.w0 = pl->w0;
        b.h0 = pl->h0;
        b.w = L.w;
        b.h = L.h;
        b.ld = L.ld;
        b.k0 = L.h_kern[1];
        b.k1 = L.h_kern[0];
        b.aligned4 = (stride % 4 == 0) ? e->img_aligned4 : 0;
        if (L.mode == 0) TW_LAUNCH(e, TW_DF_PYR_K3, tw_pyr_k3<0>, dim3((L.w + 255) / 256, (L.h + 7) / 8, nimg), dim3(256), 0, st, b);
        else TW_LAUNCH(e, TW_DF_PYR_K3, tw_pyr_k3<2>, dim3((L.w + 255) / 256, (L.h + 3) / 4, nimg), dim3(256), 0, st, b);
        return;
    }
    if (L.pitch_b > 0 && L.mode != 0 && L.ksize >= 7 && e->pyr_generic == 0) {
        // coarse levels: shared-window row filter (tw_pyr_taps)
        PyrTapsArgs b;
        b.p = a;
        b.pd = (L.pitch_b / 4 + 1) | 1;
        b.aligned4 = (stride % 4 == 0) ? e->img_aligned4 : 0;
        b.dbg = e->dbg_stamps;
        memset(b.kext, 0, sizeof(b.kext));
        memcpy(b.kext + 1, L.h_taps.data(), sizeof(float) * L.ksize);
        const size_t lds3 = ((size_t)L.nrows_max * (2 * PYR_TW) + (size_t)L.nrows_max * b.pd + 4) * 4;
        switch (L.ksize) {
            case 9: TW_LAUNCH(e, TW_DF_PYR_TAPS, tw_pyr_taps<9>, grid, dim3(256), lds3, st, b); break;
            case 19: TW_LAUNCH(e, TW_DF_PYR_TAPS, tw_pyr_taps<19>, grid, dim3(256), lds3, st, b); break;
            case 39: TW_LAUNCH(e, TW_DF_PYR_TAPS, tw_pyr_taps<39>, grid, dim3(256), lds3, st, b); break;
            default: TW_LAUNCH(e, TW_DF_PYR_TAPS, tw_pyr_taps<0>, grid, dim3(256), lds3, st, b); break;
        }
        return;
    }
    if (L.pitch_b > 0 && e->pyr_generic != 1) {
        PyrLdsArgs b;
        b.p = a;
        b.pitch_b = L.pitch_b;
        b.aligned4 = (stride % 4 == 0) ? e->img_aligned4 : 0;
        TW_LAUNCH(e, TW_DF_PYR_LEVEL, tw_pyr_level_lds, grid, dim3(256), lds + (size_t)L.nrows_max * L.pitch_b, st, b);
        return;
    }
    {
        a.th = L.gen_th;
        a.nrows_max = L.gen_nrows_max;
        const size_t lds_g = (size_t)(PYR_MAXK + (size_t)L.gen_nrows_max * P) * 4;
        const dim3 grid_g((L.w + PYR_TW - 1) / PYR_TW, (L.h + L.gen_th - 1) / L.gen_th, nimg);
        TW_LAUNCH(e, TW_DF_PYR_LEVEL, tw_pyr_level, grid_g, dim3(256), lds_g, st, a);
    }
}

// levels 3 and 2 of `nimg` images from one read of each image (tw_pyr_23; pl->fused23): I3 / I2 = the level images,
// z-th at + z * ps of the level
void launch_pyr23(tw_engine* e, hipStream_t st, const Plan* pl, const uint8_t* const* d_srcs, long long stride, float* I3,
                  float* I2, int nimg)
{
    const LevelPlan &L3 = pl->lv[3], &L2 = pl->lv[2];
    Pyr23Args a;
    a.srcs = d_srcs;
    a.dst3 = I3;
    a.dst2 = I2;
    a.zs3 = L3.ps;
    a.zs2 = L2.ps;
    a.stride = stride;
    a.w0 = pl->w0;
    a.h0 = pl->h0;
    a.w3 = L3.w;
    a.h3 = L3.h;
    a.ld3 = L3.ld;
    a.w2 = L2.w;
    a.h2 = L2.h;
    a.ld2 = L2.ld;
    a.aligned4 = (stride % 4 == 0) ? e->img_aligned4 : 0;
    memcpy(a.k19, pl->k19, sizeof(a.k19));
    memcpy(a.k9, pl->k9, sizeof(a.k9));
    ProfScope ps(e, st, TW_K_PYR, 3);
    const dim3 grid((L3.w + P23_T3W - 1) / P23_T3W, (L3.h + P23_T3H - 1) / P23_T3H, nimg);
    const int nt = e->pyr23_threads ? e->pyr23_threads : nimg <= 2 ? 1024 : 512;
    if (nt == 256) TW_LAUNCH(e, TW_DF_PYR_23, tw_pyr_23<256>, grid, dim3(256), 0, st, a);
    else if (nt == 1024) TW_LAUNCH(e, TW_DF_PYR_23, tw_pyr_23<1024>, grid, dim3(1024), 0, st, a);
    else TW_LAUNCH(e, TW_DF_PYR_23, tw_pyr_23<512>, grid, dim3(512), 0, st, a);
}

tw_status launch_polyexp(tw_engine* e, hipStream_t st, int w, int h, int ld, long long ps, const float* I, float* R,
                         int nimg, int level, const SideJob* side = nullptr, bool* side_used = nullptr)
{
    PolyArgs a;
    a.src = I;
    a.dst = R;
    a.w = w;
    a.h = h;
    a.ld = ld;
    a.ps = ps;
    a.c = e->pc;
    dim3 grid((w + PE_TW - 1) / PE_TW, (h + PE_TH - 1) / PE_TH, nimg);
    ProfScope pscope(e, st, TW_K_POLYEXP, level);
#ifdef TW_VARIANTS
    if (e->poly_variant == 0) {
        // scalar-f32 kernel (A/B: TW_POLY_VARIANT=0)
        switch (e->p.polyN) {
            case 1: TW_LAUNCH(e, TW_DF_POLYEXP, tw_polyexp<1>, grid, dim3(256), 0, st, a); break;
            case 2: TW_LAUNCH(e, TW_DF_POLYEXP, tw_polyexp<2>, grid, dim3(256), 0, st, a); break;
            case 3: TW_LAUNCH(e, TW_DF_POLYEXP, tw_polyexp<3>, grid, dim3(256), 0, st, a); break;
            case 4: TW_LAUNCH(e, TW_DF_POLYEXP, tw_polyexp<4>, grid, dim3(256), 0, st, a); break;
            case 5: TW_LAUNCH(e, TW_DF_POLYEXP, tw_polyexp<5>, grid, dim3(256), 0, st, a); break;
            case 6: TW_LAUNCH(e, TW_DF_POLYEXP, tw_polyexp<6>, grid, dim3(256), 0, st, a); break;
            case 7: TW_LAUNCH(e, TW_DF_POLYEXP, tw_polyexp<7>, grid, dim3(256), 0, st, a); break;
            default: e->err = "polyN must be 1..7"; return TW_E_UNSUPPORTED;
        }
        return TW_OK;
    }
#endif
    if (e->poly_f32) {
        // measurement variant (TW_OPT_POLYEXP_F32): float horizontal accumulators — NOT bit-exact, never the default
        switch (e->p.polyN) {
            case 5: TW_LAUNCH(e, TW_DF_POLYEXP, (tw_polyexp_pk<5, 8, 1>), grid, dim3(256), 0, st, a); break;
            case 7:
                if (e->poly_f32 == 2) TW_LAUNCH(e, TW_DF_POLYEXP, (tw_polyexp_pk<7, 8, 2>), grid, dim3(256), 0, st, a);
                else TW_LAUNCH(e, TW_DF_POLYEXP, (tw_polyexp_pk<7, 8, 1>), grid, dim3(256), 0, st, a);
                break;
            default: e->err = "TW_OPT_POLYEXP_F32 needs polyN 5 or 7"; return TW_E_UNSUPPORTED;
        }
        return TW_OK;
    }
    // packed-f32 kernel, 240 x 8 tiles; TW_POLY_VARIANT=2 selects 240 x 16 tiles (960 two-by-two items = 15 full
    // waves of the horizontal pass, 30-row window per 16 output rows, 48 KB LDS: measured equal within the noise)
#ifdef TW_VARIANTS
    const bool t16 = e->poly_variant == 2;
    if (t16) grid.y = (h + 15) / 16;
#define TW_PK_CASE(n)                                                                        \
    case n:                                                                                  \
        if (t16) TW_LAUNCH(e, TW_DF_POLYEXP, (tw_polyexp_pk<n, 16, 0>), grid, dim3(256), 0, st, a);      \
        else TW_LAUNCH(e, TW_DF_POLYEXP, (tw_polyexp_pk<n, 8, 0>), grid, dim3(256), 0, st, a);           \
        break;
#else
#define TW_PK_CASE(n)                                                                        \
    case n: TW_LAUNCH(e, TW_DF_POLYEXP, (tw_polyexp_pk<n, 8, 0>), grid, dim3(256), 0, st, a); break;
#endif
    if (side && side_used && side->kind == 2 && e->p.polyN == 7 && e->poly_variant == 1) {
        const TwinGrid t = make_twin(grid, *side);
        TW_LAUNCH(e, TW_DF_TWIN, tw_twin_poly_k3f, dim3(t.nA8 + side->n()), dim3(256), 0, st, a, side->ka, t);
        *side_used = true;
        return TW_OK;
    }
    switch (e->p.polyN) {
        TW_PK_CASE(1) TW_PK_CASE(2) TW_PK_CASE(3) TW_PK_CASE(4) TW_PK_CASE(5) TW_PK_CASE(6) TW_PK_CASE(7)
        default: e->err = "polyN must be 1..7"; return TW_E_UNSUPPORTED;
    }
#undef TW_PK_CASE
    return TW_OK;
}

// levels 1 and 0 of `nimg` images from one read of each image (tw_pyr_k3f): both 3-tap levels of an exact halving
bool pyr01_fusable(const Plan* pl)
{
    return pl->levels >= 1 && pl->lv[0].ksize == 3 && pl->lv[1].ksize == 3 && pl->lv[0].mode == 0 && pl->lv[1].mode == 2 &&
           pl->w0 == 2 * pl->lv[1].w && pl->h0 == 2 * pl->lv[1].h && pl->w0 >= 16;
}
void launch_pyr01(tw_engine* e, hipStream_t st, const Plan* pl, const uint8_t* const* d_srcs, long long stride, float* I1,
                  float* I0, int nimg)
{
    const LevelPlan &L0 = pl->lv[0], &L1 = pl->lv[1];
    PyrK3fArgs a;
    a.srcs = d_srcs;
    a.dst0 = I0;
    a.dst1 = I1;
    a.zs0 = L0.ps;
    a.zs1 = L1.ps;
    a.stride = stride;
    a.w0 = pl->w0;
    a.h0 = pl->h0;
    a.ld0 = L0.ld;
    a.w1 = L1.w;
    a.h1 = L1.h;
    a.ld1 = L1.ld;
    a.a0 = L0.h_kern[1];
    a.a1 = L0.h_kern[0];
    a.b0 = L1.h_kern[1];
    a.b1 = L1.h_kern[0];
    a.aligned4 = (stride % 4 == 0) ? e->img_aligned4 : 0;
    ProfScope ps(e, st, TW_K_PYR, 1);
    TW_LAUNCH(e, TW_DF_PYR_K3F, tw_pyr_k3f, dim3((L1.w + 255) / 256, (L1.h + 3) / 4, nimg), dim3(256), 0, st, a);
}

// One whole iteration of the flow update without M in HBM (tw_flow_iter; round 5): flow_out = solve(window average of
// FarnebackUpdateMatrices(R0, R1, flow_in)).  flow_in: a flow buffer, or (prev != nullptr) the bilinear upsample x
// 1/pyr_scale of the coarser level's flow computed in place of the load, or zero (both null: the coarsest level).
bool flow_iter_eligible(const tw_engine* e, int w, int h)
{
    return e->win_m == 15 && !e->box && w >= e->mfree_min_w && h >= 4 * FI_TH;
}
// ONE predicate for the schedule (flush_ctx) and the byte model (tw_algorithmic_bytes_launch): does a launch of nc
// pairs at a level of w x h pixels run tw_flow_iter?  lat: the batch takes the single-pair schedule; grid_only: the level's
// last iteration is evaluated at the span-grid points only (TW_OPT_SCAN_FUSED_FINAL, level 0).  A launch of fewer workgroups
// than ~3/4 of the CUs — one or two 1080p pairs — leaves the chip to the 224 x 8 tiles of tw_blur_solve4, of which a single
// pair already makes 1 215 (TW_MFREE=2 lifts that gate: tests, tools/fuzz_parity.py).
bool level_runs_flow_iter(const tw_engine* e, int w, int h, int nc, bool lat, bool grid_only)
{
    const int it = e->p.pyrIterations;
    const int fi_out = (e->fi_nt == 512 ? FI_SC_512 : FI_SC) - 30;
    const long long fi_wgs = (long long)((w + fi_out - 1) / fi_out) * std::min(4, std::max(1, h / (16 * FI_TH))) * nc;
    return e->mfree && !lat && it > (grid_only ? 1 : 0) && flow_iter_eligible(e, w, h) &&
           (fi_wgs * 4 >= (long long)e->cu_count * 3 || e->mfree == 2) &&
           (e->mfree_min_px <= 0 || (long long)w * h >= e->mfree_min_px);
}
// does a batch of n pairs of w0 x h0 pixels take the single-pair schedule (its image-only work beside the flow chain)?
bool single_pair_schedule(const tw_engine* e, int n, int w0, int h0, int levels)
{
    return n == 1 && e->lat_streams && levels >= 1 && (long long)w0 * h0 >= e->lat_min_px;
}
// the last level-0 iteration at the span-grid points only?
bool scan_fused_level0(const tw_engine* e, int span)
{
    return e->p.pyrIterations > 0 && e->scan_fused && span == 10 && e->win_m == 15 && !e->box;
}
// pairs per launch when n pairs are split into balanced launches of at most `chunk` (64 pairs with a 63-pair chunk are 32 + 32)
int balanced_launch_pairs(int n, int chunk)
{
    const int nlaunch = (n + chunk - 1) / chunk;
    return nlaunch > 0 ? (n + nlaunch - 1) / nlaunch : 1;
}
struct FlowUps {  // the coarser level's flow and the resize tables to this level (UpdArgs' upsample fields)
    const float* prev;
    int pw, ph, pld;
    long long pfps;
    const int *xofs, *yofs;
    const float *alpha, *beta;
    int xmax;
    float scale;
};
void launch_flow_iter(tw_engine* e, hipStream_t st, int w, int h, int ld, long long ps, const float* R, const float* flow_in,
                      long long fps_in, float* flow_out, long long fps_out, const FlowUps* ups, int npairs, int level)
{
    FlowIterArgs a;
    memset(&a, 0, sizeof(a));
    a.R = R;
    a.flow_in = flow_in;
    a.flow_out = flow_out;
    a.w = w;
    a.h = h;
    a.ld = ld;
    a.ps = ps;
    a.fps_in = fps_in;
    a.fps_out = fps_out;
    a.zero_flow = (!flow_in && !ups) ? 1 : 0;
    a.c = e->wc;
#ifdef TW_VARIANTS
    a.dbg_skip = e->fi_skip;                     // TW_FI_SKIP (timing experiments; read once in tw_engine_create)
    a.dbg = (unsigned long long*)e->dbg_stamps;  // TW_DEBUG_STAMPS=1
#endif
    const int OUT = (e->fi_nt == 512 ? FI_SC_512 : FI_SC) - 30;
    const int slots = e->cu_count * (e->fi_nt == 512 ? 2 : 1);  // workgroups resident at once
    const int nstrips = (w + OUT - 1) / OUT, nsteps = (h + FI_TH - 1) / FI_TH;
    // Row segments per strip: one 1024-thread workgroup per CU is resident, so the launch runs in rounds of (CU count)
    // workgroups; a segment pays NCH = 7 chunks of M for its window's warm-up.  Take the split (1 .. 4) with the least
    // (rounds x (steps + warm-up)) — 128 pairs x 12 strips of 1080p: 2 segments = 12.0 rounds of 108 + 7 steps.
    int best_seg = 1;
    double best_cost = 1e300;
    for (int sg = 1; sg <= e->fi_maxseg; sg++) {
        const int nt = (nsteps + sg - 1) / sg;
        if (sg > 1 && nt < e->fi_minsteps) break;
        const long long wgs = (long long)nstrips * sg * npairs;
        const long long rounds = (wgs + slots - 1) / slots;
        const double cost = (double)rounds * (nt + 7 * 0.25);  // a warm-up chunk is a quarter of a step (phase C only)
        if (cost < best_cost) {
            best_cost = cost;
            best_seg = sg;
        }
    }
    a.nt = (nsteps + best_seg - 1) / best_seg;
    const int nseg = (nsteps + a.nt - 1) / a.nt;
    const dim3 grid(nstrips, nseg, npairs);
    ProfScope pscope(e, st, TW_K_BLUR_SOLVE, level);
    if (ups) {
        a.prev = ups->prev;
        a.pw = ups->pw;
        a.ph = ups->ph;
        a.pld = ups->pld;
        a.pfps = ups->pfps;
        a.xofs = ups->xofs;
        a.alpha = ups->alpha;
        a.yofs = ups->yofs;
        a.beta = ups->beta;
        a.xmax = ups->xmax;
        a.scale = ups->scale;
    }
#ifdef TW_VARIANTS
    if (e->fi_nt == 512) {  // measured 23 % slower (profiles/r05_m_free.md): variants library only
        if (ups) TW_LAUNCH(e, TW_DF_FLOW_ITER_UPS, (tw_flow_iter<15, 1, 512>), grid, dim3(512), 0, st, a);
        else if (a.zero_flow) TW_LAUNCH(e, TW_DF_FLOW_ITER_ZERO, (tw_flow_iter<15, 2, 512>), grid, dim3(512), 0, st, a);
        else TW_LAUNCH(e, TW_DF_FLOW_ITER, (tw_flow_iter<15, 0, 512>), grid, dim3(512), 0, st, a);
        return;
    }
#endif
    if (ups) {
        TW_LAUNCH(e, TW_DF_FLOW_ITER_UPS, (tw_flow_iter<15, 1>), grid, dim3(1024), 0, st, a);
    } else if (a.zero_flow) {
        TW_LAUNCH(e, TW_DF_FLOW_ITER_ZERO, (tw_flow_iter<15, 2>), grid, dim3(1024), 0, st, a);
    } else {
        TW_LAUNCH(e, TW_DF_FLOW_ITER, (tw_flow_iter<15, 0>), grid, dim3(1024), 0, st, a);
    }
}

// span-grid samples a tw_blur_solve4 launch stores next to the flow (the single-pair schedule's last level-0 launch)
struct GridOut {
    float2* g;
    int span, gw, gh;
    bool* used;  // set if the launch took a kernel that stores them (the caller launches tw_span_gather otherwise)
};
void launch_blur(tw_engine* e, hipStream_t st, int w, int h, int ld, long long ps, const float* Min, float* Mout,
                 float* flow, const float* R, int update, int level, int npairs, double* Vbox = nullptr,
                 const SideJob* side = nullptr, bool* side_used = nullptr, const GridOut* go = nullptr, bool quads = false)
{
    BlurArgs a;
    a.Min = Min;
    a.Mout = Mout;
    a.flow = flow;
    a.R = R;
    a.w = w;
    a.h = h;
    a.ld = ld;
    a.ps = ps;
    a.fps = ps;
    a.update = update;
    a.xsh = 0;
    a.rot = 0;
    a.store_flow = level < 0 ? 1 : 0;  // level -1: the per-stage test entry point, which returns the flow as well
    a.m = e->win_m;
    a.nomask = e->blur_nomask;
    a.cm = e->blur_cm;
    a.grid = nullptr;
    a.gspan = a.gw = a.gh = 0;
    a.gmagic = 0;
    a.c = e->wc;
    // tw_blur_solve4<15, ...> stores the span-grid samples next to the flow when asked to (GridOut)
    auto set_grid_out = [&]() {
        if (go && go->span > 1 && w < 65536 && h < 65536) {
            a.grid = go->g;
            a.gspan = go->span;
            a.gw = go->gw;
            a.gh = go->gh;
            a.gmagic = (unsigned)((0x100000000ull + (unsigned)go->span - 1) / (unsigned)go->span);
            *go->used = true;
        }
    };
    const int gy = (h + BS_TH - 1) / BS_TH;
    const bool wide = w > 480;  // 224-column tiles; narrow levels use 96-column tiles (less edge waste)
    if (e->box) {
        // box window: sequential double running sums (two scan kernels) + the standard refresh kernel
        double* V = Vbox ? Vbox : e->Vd;  // TW_LANES=2: each lane (stream) has its own column sums
        Tmp tmp;
        if (level < 0) {  // per-stage test entry: no engine workspace
            V = tmp.alloc<double>((size_t)ps * 5 * npairs);
            if (!V) return;
        }
        BoxArgs b;
        b.Min = Min;
        b.V = V;
        b.flow = flow;
        b.w = w;
        b.h = h;
        b.ld = ld;
        b.ps = ps;
        b.fps = ps;
        b.m = e->win_m;
        b.scale = 1. / ((double)e->p.winSize * e->p.winSize);
        {
            ProfScope pscope(e, st, TW_K_BLUR_SOLVE, level);
            TW_LAUNCH(e, TW_DF_BOX, tw_box_vscan, dim3((w + 63) / 64, 5, npairs), dim3(64), 0, st, b);
            TW_LAUNCH(e, TW_DF_BOX, tw_box_hscan_solve, dim3((h + 11) / 12, 1, npairs), dim3(64), 0, st, b);
        }
        if (update) {
            UpdArgs u;
            memset(&u, 0, sizeof(u));
            u.R = R;
            u.flow = flow;
            u.M = Mout;
            u.w = w;
            u.h = h;
            u.ld = ld;
            u.ps = ps;
            u.fps = ps;
            ProfScope pscope(e, st, TW_K_UPDATE_MATRICES, level);
            launch_upd_kernel<false>(e, st, w, h, npairs, u);
        }
        if (level < 0) (void)hipStreamSynchronize(st);  // tmp is freed on return
        return;
    }
    ProfScope pscope(e, st, TW_K_BLUR_SOLVE, level);
#ifdef TW_VARIANTS
    if (e->win_m == 15 && e->blur_variant == 8) {
        // packed-f32 structure (same speed as v4 at 1080p, lower VALU load); TW_BLUR_VARIANT=8 for A/B
        if (wide) TW_LAUNCH(e, TW_DF_BLUR_SOLVE8, (tw_blur_solve8<15, 256, 16, 8, true, true>), dim3((w + 223) / 224, gy, npairs), dim3(256), 0, st, a);
        else TW_LAUNCH(e, TW_DF_BLUR_SOLVE8, (tw_blur_solve8<15, 128, 16, 8, true, true>), dim3((w + 95) / 96, gy, npairs), dim3(128), 0, st, a);
        return;
    }
#endif
    if (e->win_m == 15) {
        // Small grids (a single pair, the coarse levels): a level that would launch fewer than two workgroups per
        // CU takes smaller tiles, so that more CUs share it and each workgroup's serial V -> H -> S chain is shorter
        // (BASELINE config 2, single-pair latency).  Batched launches always have enough tiles and keep the big ones.
        auto nwg = [&](int tw, int th) { return (long long)((w + tw - 1) / tw) * ((h + th - 1) / th) * npairs; };
        int small = 0;  // 0: 224x8 / 96x8 tiles as below, 1: 96x8 (128 threads), 2: 96x4 (128 threads), 3: 32x4 (64 threads)
                        // 4: plane-parallel 32x8 (320 threads), 5: plane-parallel 96x8 (640 threads)
        if (level >= 0 && level < 8 && e->blur_small_lv[level] >= 0) small = e->blur_small_lv[level];
        else if (e->blur_small >= 0) small = e->blur_small;
        else if (wide && nwg(224, 8) >= 1024) small = 0;   // four 256-thread workgroups per CU: throughput regime
        else if (nwg(96, 8) * 2 >= e->pp_waves) small = wide ? 1 : 0;  // enough 2-wave workgroups to keep every SIMD busy
        else small = 4;  // otherwise many small plane-parallel workgroups (5 waves per 32x8 pixels)
        if (small) {
            const int tw = (small == 3 || small == 4) ? 32 : 96, th = (small == 2 || small == 3) ? 4 : 8;
            a.xsh = ((w + 16 + tw - 1) / tw == (w + tw - 1) / tw) ? 16 : 0;
            a.rot = a.xsh;
            const dim3 grid((w + a.xsh + tw - 1) / tw, (h + th - 1) / th, npairs);
            if (small == 1) {
                set_grid_out();
                if (side && side_used && side->kind == 1) {
                    // (round 6) two-wave workgroups that leave most of the chip idle: a band of a finer level's expansion rides along
                    const TwinGrid t = make_twin(grid, *side);
                    TW_LAUNCH(e, TW_DF_TWIN, tw_twin_s4_poly, dim3(t.nA8 + side->n()), dim3(256), 0, st, a, side->pa, t);
                    *side_used = true;
                } else {
                    TW_LAUNCH(e, TW_DF_BLUR_SOLVE4, (tw_blur_solve4<15, 128, 16, 8, true>), grid, dim3(128), 0, st, a);
                }
            }
#ifdef TW_VARIANTS
            else if (small == 2) TW_LAUNCH(e, TW_DF_BLUR_SOLVE4, (tw_blur_solve4<15, 128, 16, 4, true>), grid, dim3(128), 0, st, a);
            else if (small == 3) TW_LAUNCH(e, TW_DF_BLUR_SOLVE4, (tw_blur_solve4<15, 64, 16, 4, true>), grid, dim3(64), 0, st, a);
            else if (small == 5) TW_LAUNCH(e, TW_DF_BLUR_PP, (tw_blur_solve_pp<15, 128, 16, 8>), grid, dim3(640), 0, st, a);
#endif
            else if (side && side_used && side->kind == 1) {
                const TwinGrid t = make_twin(grid, *side);
                TW_LAUNCH(e, TW_DF_TWIN, tw_twin_pp_poly, dim3(t.nA8 + side->n()), dim3(320), 0, st, a, side->pa, t);
                *side_used = true;
            }
            else TW_LAUNCH(e, TW_DF_BLUR_PP, (tw_blur_solve_pp<15, 64, 16, 8>), grid, dim3(320), 0, st, a);
            return;
        }
#ifdef TW_VARIANTS
        if (wide && (e->blur_variant == 60 || e->blur_variant == 61)) {
            // plane pipeline through a two-plane LDS ring (tw_blur_solve6), A/B
            a.xsh = ((w + 16 + 223) / 224 == (w + 223) / 224) ? 16 : 0;
            a.rot = a.xsh;
            const dim3 grid((w + a.xsh + 223) / 224, gy, npairs);
            if (e->blur_variant == 60) TW_LAUNCH(e, TW_DF_BLUR_VARIANT, (tw_blur_solve6<15, 256, 16, 8, 5>), grid, dim3(256), 0, st, a);
            else TW_LAUNCH(e, TW_DF_BLUR_VARIANT, (tw_blur_solve6<15, 256, 16, 8, 4>), grid, dim3(256), 0, st, a);
            return;
        }
        if (wide && e->blur_variant == 5 && w >= 960) {
            // 480-column tiles (512 threads, 80 KB LDS, two workgroups per CU): 93.75 % of the columns of the vertical
            // pass and of the lanes of the horizontal pass are useful (87.5 % with 224-column tiles), and 1920 / 960 /
            // 3840-pixel rows are covered exactly (nine 224-column tiles cover 2016)
            a.xsh = ((w + 16 + 479) / 480 == (w + 479) / 480) ? 16 : 0;
            a.rot = a.xsh;
            TW_LAUNCH(e, TW_DF_BLUR_SOLVE4, (tw_blur_solve4<15, 512, 16, 8, true>), dim3((w + a.xsh + 479) / 480, gy, npairs), dim3(512), 0, st, a);
            return;
        }
#endif
        // shift the tile grid 16 px left when that costs no extra tile column (see the kernel)
        const int tw = wide ? 224 : 96;
        a.xsh = ((w + 16 + tw - 1) / tw == (w + tw - 1) / tw) ? 16 : 0;
        a.rot = a.xsh;
#ifdef TW_VARIANTS
        if (e->blur_variant == 9) {  // round 4: solve + refresh by the horizontal item's owner, 16-byte R0 / M accesses
            if (wide) TW_LAUNCH(e, TW_DF_BLUR_VARIANT, (tw_blur_solve4q<15, 256, 16, 8, true>), dim3((w + a.xsh + 223) / 224, gy, npairs), dim3(256), 0, st, a);
            else TW_LAUNCH(e, TW_DF_BLUR_VARIANT, (tw_blur_solve4q<15, 128, 16, 8, true>), dim3((w + a.xsh + 95) / 96, gy, npairs), dim3(128), 0, st, a);
            return;
        }
        if (wide && e->blur_variant == 2) { TW_LAUNCH(e, TW_DF_BLUR_SOLVE4Y, (tw_blur_solve4y<15, 256, 16, 8, 2>), dim3((w + a.xsh + 223) / 224, (h + 15) / 16, npairs), dim3(256), 0, st, a); return; }
        if (wide && e->blur_variant == 7) { TW_LAUNCH(e, TW_DF_BLUR_SOLVE4, (tw_blur_solve4<15, 256, 16, 8, true, 2, 2, 2, true, false>), dim3((w + a.xsh + 223) / 224, gy, npairs), dim3(256), 0, st, a); return; }
        if (wide && e->blur_variant == 6) { TW_LAUNCH(e, TW_DF_BLUR_SOLVE4, (tw_blur_solve4<15, 256, 16, 8, true, 2, 2, 2, false>), dim3((w + a.xsh + 223) / 224, gy, npairs), dim3(256), 0, st, a); return; }
#endif
#ifdef TW_VARIANTS
        if (wide && update && e->blur_pipe > 0) {
            // refreshing launch as a cross-tile pipeline: a workgroup walks blur_pipe vertically adjacent tiles
            const int nt = e->blur_pipe;
            const dim3 grid((w + a.xsh + 223) / 224, (gy + nt - 1) / nt, npairs);
            switch (nt) {
                case 3: TW_LAUNCH(e, TW_DF_BLUR_VARIANT, (tw_blur_solve4p<15, 256, 16, 8, 3>), grid, dim3(256), 0, st, a); return;
                case 5: TW_LAUNCH(e, TW_DF_BLUR_VARIANT, (tw_blur_solve4p<15, 256, 16, 8, 5>), grid, dim3(256), 0, st, a); return;
                case 9: TW_LAUNCH(e, TW_DF_BLUR_VARIANT, (tw_blur_solve4p<15, 256, 16, 8, 9>), grid, dim3(256), 0, st, a); return;
                case 15: TW_LAUNCH(e, TW_DF_BLUR_VARIANT, (tw_blur_solve4p<15, 256, 16, 8, 15>), grid, dim3(256), 0, st, a); return;
                case 109: TW_LAUNCH(e, TW_DF_BLUR_VARIANT, (tw_blur_solve4p<15, 256, 16, 8, 9, 3>), dim3(grid.x, (gy + 8) / 9, npairs), dim3(256), 0, st, a); return;
                default: break;
            }
        }
#endif
        set_grid_out();
        // quads (the single-pair schedule): the launch is bound by its workgroups' serial chain, not by throughput — the kernel
        // that solves and refreshes in the horizontal item's owner (two barriers and an LDS round trip fewer) is 6 % shorter there
        if (wide && quads) TW_LAUNCH(e, TW_DF_BLUR_SOLVE4Q, (tw_blur_solve4q<15, 256, 16, 8, true>), dim3((w + a.xsh + 223) / 224, gy, npairs), dim3(256), 0, st, a);
        else if (wide) TW_LAUNCH(e, TW_DF_BLUR_SOLVE4, (tw_blur_solve4<15, 256, 16, 8, true>), dim3((w + a.xsh + 223) / 224, gy, npairs), dim3(256), 0, st, a);
        else TW_LAUNCH(e, TW_DF_BLUR_SOLVE4, (tw_blur_solve4<15, 128, 16, 8, true>), dim3((w + a.xsh + 95) / 96, gy, npairs), dim3(128), 0, st, a);
    } else if (e->win_m == 25) {
        // winSize 50/51 (BASELINE config 5): packed-f32 structure, single 58-row register window
#ifdef TW_VARIANTS
        if (wide && e->blur_variant == 8) { TW_LAUNCH(e, TW_DF_BLUR_SOLVE8, (tw_blur_solve8<25, 256, 32, 8, true, false>), dim3((w + 191) / 192, gy, npairs), dim3(256), 0, st, a); return; }
        // round 3 trials for config 5: shorter tiles make room for the R0 prefetch (QPRE)
        if (wide && e->blur_variant == 256) { TW_LAUNCH(e, TW_DF_BLUR_SOLVE4, (tw_blur_solve4<25, 256, 32, 6, true, 2, 2, 2, true, false>), dim3((w + 191) / 192, (h + 5) / 6, npairs), dim3(256), 0, st, a); return; }
        if (wide && e->blur_variant == 255) { TW_LAUNCH(e, TW_DF_BLUR_SOLVE4, (tw_blur_solve4<25, 256, 32, 5, true, 2, 2, 2, true, false>), dim3((w + 191) / 192, (h + 4) / 5, npairs), dim3(256), 0, st, a); return; }
        if (wide && e->blur_variant == 254) { TW_LAUNCH(e, TW_DF_BLUR_SOLVE4, (tw_blur_solve4<25, 256, 32, 4, true, 2, 2, 2, true, false>), dim3((w + 191) / 192, (h + 3) / 4, npairs), dim3(256), 0, st, a); return; }
        if (wide && e->blur_variant == 258) { TW_LAUNCH(e, TW_DF_BLUR_SOLVE4, (tw_blur_solve4<25, 256, 32, 8, true, 2, 2, 2, true, false>), dim3((w + 191) / 192, gy, npairs), dim3(256), 0, st, a); return; }
        if (wide && e->blur_variant == 41) { TW_LAUNCH(e, TW_DF_BLUR_SOLVE4, (tw_blur_solve4<25, 256, 32, 8, true, 2, 2, 2, false, false>), dim3((w + 191) / 192, gy, npairs), dim3(256), 0, st, a); return; }
        if (wide && e->blur_variant == 259) { TW_LAUNCH(e, TW_DF_BLUR_SOLVE4, (tw_blur_solve4<25, 256, 32, 5, true, 2, 2, 2, false, false>), dim3((w + 191) / 192, (h + 4) / 5, npairs), dim3(256), 0, st, a); return; }
#endif
        // wide levels: two vertically adjacent 8-row sub-tiles share one 66-row register window (round 3: -7 % against the
        // one-sub-tile kernel with its 58-row window per 8 rows, which TW_BLUR_VARIANT=41 of the variants library selects)
#ifndef TW_4Y_ILP  // (A/B builds: vertical / horizontal interleave and solve unroll of the 51-tap kernel)
#define TW_4Y_ILP 2, 2, 2
#endif
        if (wide) TW_LAUNCH(e, TW_DF_BLUR_SOLVE4Y, (tw_blur_solve4y<25, 256, 32, 8, 2, TW_4Y_ILP>), dim3((w + 191) / 192, (h + 15) / 16, npairs), dim3(256), 0, st, a);
        else TW_LAUNCH(e, TW_DF_BLUR_SOLVE8, (tw_blur_solve8<25, 128, 32, 8, true, false>), dim3((w + 63) / 64, gy, npairs), dim3(128), 0, st, a);
    } else {
        // any other window size: generic kernel (same arithmetic, runtime loops)
        const size_t lds = (size_t)5 * BS_TH * (64 + 2 * e->win_m) * 4;
        TW_LAUNCH(e, TW_DF_BLUR_GENERIC, tw_blur_solve_generic, dim3((w + 63) / 64, gy, npairs), dim3(256), lds, st, a);
    }
}

void launch_update(tw_engine* e, hipStream_t st, const Plan* pl, int k, const float* R, float* flow,
                   const float* prev, float* M, int npairs, const SideJob* side = nullptr, bool* side_used = nullptr)
{
    const LevelPlan& L = pl->lv[k];
    UpdArgs a;
    memset(&a, 0, sizeof(a));
    a.R = R;
    a.flow = flow;
    a.M = M;
    a.w = L.w;
    a.h = L.h;
    a.ld = L.ld;
    a.ps = L.ps;
    a.fps = L.ps;
    a.store_flow = e->p.pyrIterations == 0 ? 1 : 0;
    ProfScope pscope(e, st, TW_K_UPDATE_MATRICES, k);
    if (k < pl->levels) {
        const LevelPlan& P = pl->lv[k + 1];
        a.prev = prev;
        a.pw = P.w;
        a.ph = P.h;
        a.pld = P.ld;
        a.pfps = P.ps;
        a.xofs = L.d_uxofs;
        a.alpha = L.d_ualpha;
        a.yofs = L.d_uyofs;
        a.beta = L.d_ubeta;
        a.xmax = L.uxmax;
        a.scale = (float)(1. / e->p.pyrScale);
        launch_upd_kernel<true>(e, st, L.w, L.h, npairs, a, side, side_used);
    } else {
        a.zero_flow = 1;
        launch_upd_kernel<false>(e, st, L.w, L.h, npairs, a, side, side_used);
    }
}

// Enqueue a whole batch, level-major: every level is processed for all pairs (in chunks sized to fill the
// chip) before the next finer level starts; level 0 chunks are scanned as soon as their flow exists.
tw_status flush_ctx(tw_engine* e, Ctx& c)
{
    if (c.launched || c.jobs.empty()) return TW_OK;
    hipStream_t st = e->stream;
    Plan* pl = nullptr;
    tw_status r = get_plan(e, c.w, c.h, &pl);
    if (r) return r;
    if ((r = reserve_workspace(e, pl, c.span, c.any_host))) return r;
    if (c.span > 0) {
        const size_t need_rec = (size_t)tw_grid_capacity(c.w, c.h, c.span) * e->cap;
        if (need_rec > c.d_rec_cap) {
            // this context has no results outstanding (it is being launched), nothing on the stream uses its records
            if (c.d_rec) (void)hipFree(c.d_rec);
            c.d_rec = nullptr;
            c.d_rec_cap = 0;
            TW_HIP(e, hipMalloc((void**)&c.d_rec, need_rec * sizeof(ScanRec) + 256));
            c.d_rec_cap = need_rec;
        }
    }
    const int n = (int)c.jobs.size();
    int ramp_b[4] = {0, 0, 0, 0}, ramp_n = 1;  // cold-start ramp: piece i = pairs [ramp_b[i], ramp_b[i + 1]); ramp_n - 1 pieces
    RoctxRange batch_range("tw_batch %dx%d pairs=%d", c.w, c.h, n);
    const size_t npx = staged_image_bytes(c.w, c.h);  // staged images are 256-byte aligned
    long long stride = c.jobs[0].stride;
    if (c.any_png) {
        // tw_submit_png8: reconstruct the filtered rows into the gray image slots, on the copy stream behind the
        // uploads — it runs beside the previous batch's kernels like the uploads themselves
        for (int j = 0; j < n; j++)
            for (int q = 0; q < 2; q++) {
                PngJob& pj = c.h_png[2 * j + q];
                pj.ch = c.jobs[j].png_ch[q];
                pj.src = c.d_filt ? c.d_filt + c.filt_slot * (size_t)(2 * j + q) : nullptr;
                pj.dst = c.d_img + npx * (size_t)(2 * j + q);
                pj.pad = 0;
            }
        TW_HIP(e, hipMemcpyAsync(c.d_png, c.h_png, sizeof(PngJob) * 2 * (size_t)n, hipMemcpyHostToDevice, e->copy_stream));
        PngArgs pa;
        pa.jobs = c.d_png;
        pa.w = c.w;
        pa.h = c.h;
        // waves per image: as many as the 128 KB of LDS row buffers allow for this width
        if (c.w <= PNG_LDS_PIXELS / 16) TW_LAUNCH(e, TW_DF_PNG_UNFILTER, tw_png_unfilter<16>, dim3(2 * n), dim3(1024), 0, e->copy_stream, pa);
        else if (c.w <= PNG_LDS_PIXELS / 4) TW_LAUNCH(e, TW_DF_PNG_UNFILTER, tw_png_unfilter<4>, dim3(2 * n), dim3(256), 0, e->copy_stream, pa);
        else TW_LAUNCH(e, TW_DF_PNG_UNFILTER, tw_png_unfilter<1>, dim3(2 * n), dim3(64), 0, e->copy_stream, pa);
    }
    c.copy_ops_at_flush = e->copy_ops;
    if (c.any_host) {
        // the uploads were queued on the copy stream as the jobs came in (submit_common)
        TW_HIP(e, hipEventRecord(c.ev_h2d, e->copy_stream));
        // cold-start ramp: nothing of an earlier batch is still running on the compute stream (the first batch after a
        // pause, or a pipeline the host cannot keep filled) — the pairs whose uploads are already marked start now, the
        // rest behind their own marks, instead of all of them behind the batch's last upload (531 MB for 128 pairs of
        // 1080p: ~14 ms of an idle GPU).  A busy stream takes the whole batch in one launch per kernel and level, as before.
        if (e->ramp && e->lanes == 1 && !c.any_png && c.nseg > 0 && c.seg_at[0] < n) {
            bool idle = true;
            for (Ctx& o : e->ctx)
                if (&o != &c && o.launched && hipEventQuery(o.ev_done) == hipErrorNotReady) idle = false;
            (void)hipGetLastError();  // (a "not ready" answer is not an error to report later)
            if (idle) {
                for (int i = 0; i < c.nseg; i++)
                    if (c.seg_at[i] < n) ramp_b[ramp_n++] = c.seg_at[i];
            }
        }
        ramp_b[ramp_n++] = n;
        if (ramp_n == 2) TW_HIP(e, hipStreamWaitEvent(st, c.ev_h2d, 0));  // (no ramp: the whole batch behind its last upload)
        stride = c.w;
    }
    bool al4 = (stride % 4) == 0;
    for (int j = 0; j < n; j++) {
        const Job& jb = c.jobs[j];
        if (jb.h_a) {
            c.h_ptrs[2 * j] = c.d_img + npx * (2 * j);
            c.h_ptrs[2 * j + 1] = c.d_img + npx * (2 * j + 1);
        } else {
            c.h_ptrs[2 * j] = jb.d_a;
            c.h_ptrs[2 * j + 1] = jb.d_b;
        }
        al4 = al4 && ((uintptr_t)c.h_ptrs[2 * j] % 4 == 0) && ((uintptr_t)c.h_ptrs[2 * j + 1] % 4 == 0);
    }
    e->img_aligned4 = al4 ? 1 : 0;
    TW_HIP(e, hipMemcpyAsync((void*)e->d_ptrs, c.h_ptrs, sizeof(void*) * 2 * n, hipMemcpyHostToDevice, st));
    TW_HIP(e, hipEventRecord(c.ev_start, st));
    const int it = e->p.pyrIterations;
    const int nlanes = (e->lanes > 1 && n >= 2) ? e->lanes : 1;
    if (nlanes > 1) {
        TW_HIP(e, hipEventRecord(e->ev_fork, st));  // uploads + whatever ran before on the main stream
        TW_HIP(e, hipStreamWaitEvent(e->stream2, e->ev_fork, 0));
    }
    size_t ws_lane = 0;
    for (const LevelPlan& L : pl->lv) ws_lane = std::max(ws_lane, (size_t)L.ps * 2 * L.chunk);
    // one pair: image-only work (pyramid, polynomial expansion) of all levels on the second stream
    // (only worth its cross-stream hand-offs when the image-only work is tens of microseconds: >= 0.1 Mpixel)
    const bool lat = nlanes == 1 && e->lat_I && single_pair_schedule(e, n, c.w, c.h, pl->levels);
    bool prof_on = false;
    for (int i = 0; i < TW_K_COUNT; i++) prof_on = prof_on || e->prof_level[i] != -2;
    // ... or, for the default parameters at sizes whose pyramid halves exactly (1080p: BASELINE config 2), ONE stream of twin
    // launches: every coarse-level chain kernel carries a piece of the finer levels' image-only work in its own launch
    // (twflow_kernels.hip.h: tw_twin_*) — no cross-queue hand-off anywhere in the pair
    const bool lat2 = lat && e->lat_fused && !e->lat_graph && !prof_on && pl->levels == 3 && pl->fused23 && e->pyr_fused &&
                      pyr01_fusable(pl) && !e->pyr_generic && e->p.polyN == 7 && !e->poly_f32 && e->poly_variant == 1 &&
                      e->upd_ny == 2 && e->win_m == 15 && !e->box && it >= 1;
    // a single pair's 224 x 8-tile window launches take tw_blur_solve4q (launch_blur; TW_LAT_QUADS=0: tw_blur_solve4 as in a batch)
    const bool lat_quads = lat && (!getenv("TW_LAT_QUADS") || atoi(getenv("TW_LAT_QUADS")) != 0);
    std::vector<SideJob> sideq;
    size_t side_next = 0;
    std::vector<size_t> lat_off(pl->lv.size(), 0);
    if (lat) {
        size_t off = 0;
        for (size_t k = 0; k < pl->lv.size(); k++) {
            lat_off[k] = off;
            off += (size_t)pl->lv[k].ps * 2;
        }
    }
    // Everything between the batch's start event and the ordered scan, as one function: the single-pair schedule
    // replays it from a captured hipGraph (below), every other batch enqueues it directly.
    // does level k of a launch of nc pairs run tw_flow_iter?  (level_runs_flow_iter: the predicate the byte model shares)
    auto level_mfree = [&](int k, int nc) -> bool {
        const LevelPlan& L = pl->lv[k];
        return level_runs_flow_iter(e, L.w, L.h, nc, lat, k == 0 && scan_fused_level0(e, c.span));
    };
    auto enqueue_levels = [&]() -> tw_status {
    if (lat && !lat2) {
        TW_HIP(e, hipEventRecord(e->ev_fork, st));  // pointer table + uploads + the previous batch's use of lat_I / lat_R
        TW_HIP(e, hipStreamWaitEvent(e->stream2, e->ev_fork, 0));
    }
    // image-only work of level k in the single-pair schedule: the coarsest level's is needed at once and stays on the
    // main stream (no hand-off to wait for); the finer levels' go to the second stream.  It is ENQUEUED one level
    // ahead of the flow chain that consumes it (round 3: with all of it enqueued up front the host needed ~40 us for
    // those launches before the first flow-chain kernel was even submitted — the GPU sat idle behind the host).
    auto lat_images = [&](int k) -> tw_status {
        const LevelPlan& L = pl->lv[k];
        const bool second = k < pl->levels && k <= e->lat_s2_max;
        hipStream_t ws = second ? e->stream2 : st;
        launch_pyr(e, ws, pl, k, e->d_ptrs, stride, e->lat_I + lat_off[k], 2);
        tw_status rr = launch_polyexp(e, ws, L.w, L.h, L.ld, L.ps, e->lat_I + lat_off[k], e->lat_R + 5 * lat_off[k], 2, k);
        if (rr) return rr;
        if (second) TW_HIP(e, hipEventRecord(e->lat_ev[k], e->stream2));
        return TW_OK;
    };
    // the side jobs of the twin schedule, in the order the chain needs their results
    auto side_poly = [&](int k, int z0, int nz, int y0, int ny, int carry_level = -1, bool windows_only = false) {
        const LevelPlan& L = pl->lv[k];
        SideJob j;
        j.kind = 1;
        j.need_level = k;
        j.carry_level = carry_level;
        j.windows_only = windows_only;
        j.pa.src = e->lat_I + lat_off[k];
        j.pa.dst = e->lat_R + 5 * lat_off[k];
        j.pa.w = L.w;
        j.pa.h = L.h;
        j.pa.ld = L.ld;
        j.pa.ps = L.ps;
        j.pa.c = e->pc;
        j.g = VGrid{(unsigned)((L.w + PE_TW - 1) / PE_TW), (unsigned)ny, (unsigned)nz, (unsigned)y0, (unsigned)z0};
        sideq.push_back(j);
    };
    // the next side job, for the launcher of a chain kernel; side_done launches it by itself if the launcher had no twin
    // measurement: TW_LAT_ALONE=1 every side job as its own launch; =2 also the chain kernels through their twins (empty side)
    static const int side_alone = getenv("TW_LAT_ALONE") ? atoi(getenv("TW_LAT_ALONE")) : 0;
    SideJob empty_side = SideJob();  // (a local: engines of several host threads run this function at once)
    empty_side.kind = 1;
    empty_side.g = VGrid{0u, 0u, 0u, 0u, 0u};
    auto side_peek = [&]() -> const SideJob* {
        if (side_next >= sideq.size()) return nullptr;
        return side_alone == 2 ? &empty_side : side_alone ? nullptr : &sideq[side_next];
    };
    // may a launch of level k's chain (its update launch / one of its window launches) carry the next job?
    auto side_ok = [&](int k, bool window) -> bool {
        if (side_next >= sideq.size()) return false;
        const SideJob& j = sideq[side_next];
        return (j.carry_level < 0 || j.carry_level == k) && (window || !j.windows_only);
    };
    auto side_done = [&](bool used) {
        if (side_next >= sideq.size()) return;
        if (!used || side_alone) launch_side_alone(e, st, sideq[side_next]);
        side_next++;
    };
    if (lat2) {
        sideq.clear();
        side_next = 0;
        launch_pyr23(e, st, pl, e->d_ptrs, stride, e->lat_I + lat_off[3], e->lat_I + lat_off[2], 2);
        {   // the coarsest level's expansion carries the level 0 / 1 images
            const LevelPlan &L0 = pl->lv[0], &L1 = pl->lv[1], &L3 = pl->lv[3];
            SideJob j;
            j.kind = 2;
            j.need_level = 1;
            PyrK3fArgs& a = j.ka;
            a.srcs = e->d_ptrs;
            a.dst0 = e->lat_I + lat_off[0];
            a.dst1 = e->lat_I + lat_off[1];
            a.zs0 = L0.ps;
            a.zs1 = L1.ps;
            a.stride = stride;
            a.w0 = pl->w0;
            a.h0 = pl->h0;
            a.ld0 = L0.ld;
            a.w1 = L1.w;
            a.h1 = L1.h;
            a.ld1 = L1.ld;
            a.a0 = L0.h_kern[1];
            a.a1 = L0.h_kern[0];
            a.b0 = L1.h_kern[1];
            a.b1 = L1.h_kern[0];
            a.aligned4 = (stride % 4 == 0) ? e->img_aligned4 : 0;
            j.g = VGrid{(unsigned)((L1.w + 255) / 256), (unsigned)((L1.h + 3) / 4), 2u, 0u, 0u};
            bool used = false;
            if ((r = launch_polyexp(e, st, L3.w, L3.h, L3.ld, L3.ps, e->lat_I + lat_off[3], e->lat_R + 5 * lat_off[3], 2, 3, &j, &used)))
                return r;
            if (!used) launch_side_alone(e, st, j);
        }
        // carriers: the update + `it` window launches of levels 3 and 2, and level 1's update
        const int carriers = 2 * (it + 1) + 1;
        const int gy2 = (pl->lv[2].h + PE_TH - 1) / PE_TH, gy1 = (pl->lv[1].h + PE_TH - 1) / PE_TH, gy0 = (pl->lv[0].h + PE_TH - 1) / PE_TH;
        // Few, large jobs on the coarsest level's launches: a chain launch stretches to its side job's own duration plus
        // ~2 us when it has CUs to spare (level 3: 136 workgroups), but by ~0.7 of the job's when it fills the chip itself
        // (level 2's window launches: 510 five-wave workgroups) — profiles/r05_single_pair.md.  TW_LAT_BANDS=n: n bands per
        // level-0 image and level 1 per image (n = 3: the first version's nine jobs on nine carriers)
        static const int bands_env = getenv("TW_LAT_BANDS") ? atoi(getenv("TW_LAT_BANDS")) : 0;
        (void)carriers;
        side_poly(2, 0, 2, 0, gy2);
        // TW_LAT_PLAN=0: round 5's plan — everything on level 3's launches (level 1's expansion, then level 0's image by image).
        // 1 (default, round 6): level 1's expansion in `it` bands on level 3's window launches, level 0's in `it` bands (of both
        // images) on LEVEL 1's window launches (tw_twin_s4_poly: 680 two-wave workgroups leave the chip three quarters idle);
        // levels 2's launches and level 1's update carry nothing (they fill the chip: a carried job costs 0.7 of itself there).
        const int plan_env = getenv("TW_LAT_PLAN") ? atoi(getenv("TW_LAT_PLAN")) : 1;  // (read per batch: the tests switch it)
        const bool l1_twin = pl->lv[1].w > 480 && (long long)((pl->lv[1].w + 95) / 96) * ((pl->lv[1].h + 7) / 8) * 2 >= e->pp_waves &&
                             (long long)((pl->lv[1].w + 223) / 224) * ((pl->lv[1].h + 7) / 8) < 1024 && e->blur_small < 0 &&
                             e->blur_small_lv[1] < 0;  // level 1 takes the 96 x 8 tiles (launch_blur)
        if (plan_env >= 1 && l1_twin && bands_env == 0) {
            auto bands = [&](int k, int gy, int nb, int carry) {  // level k's expansion (both images) as nb bands of tile rows
                const int rows = (gy + nb - 1) / nb;
                for (int y0 = 0; y0 < gy; y0 += rows) side_poly(k, 0, 2, y0, std::min(rows, gy - y0), carry, true);
            };
            bands(1, gy1, it, 3);
            bands(0, gy0, it, 1);  // (unequal bands, two of five bands on level 3's launches: all within the noise of this,
                                   //  profiles/r06_single_pair.md)
        } else {
        if (bands_env >= 2) {
            side_poly(1, 0, 1, 0, gy1);
            side_poly(1, 1, 1, 0, gy1);
        } else {
            side_poly(1, 0, 2, 0, gy1);
        }
        const int nb = std::min(8, std::max(1, bands_env));  // bands per level-0 image
        const int rows = (gy0 + nb - 1) / nb;
        for (int z = 0; z < 2; z++)
            for (int y0 = 0; y0 < gy0; y0 += rows) side_poly(0, z, 1, y0, std::min(rows, gy0 - y0));
        }
    } else if (lat) {
        if ((r = lat_images(pl->levels))) return r;
        if (pl->levels >= 1 && (r = lat_images(pl->levels - 1))) return r;
    }
    const bool ramp = ramp_n > 2;  // (only with nlanes == 1 and never for a single pair: see where ramp_b is filled)
    const int nparts = ramp ? ramp_n - 1 : nlanes;
    for (int part = 0; part < nparts; part++) {
        const int lane = ramp ? 0 : part;
        hipStream_t ls = lane == 0 ? st : e->stream2;
        const int lo = ramp ? ramp_b[part] : (int)((long long)n * lane / nlanes);
        const int hi = ramp ? ramp_b[part + 1] : (int)((long long)n * (lane + 1) / nlanes);
        if (ramp) TW_HIP(e, hipStreamWaitEvent(st, part + 1 < nparts ? c.ev_seg[part] : c.ev_h2d, 0));
        float* I = e->I + ws_lane * lane;
        float* R = e->R + ws_lane * 5 * lane;
        float* M0 = e->M[0] + ws_lane / 2 * 5 * lane;
        float* M1 = e->M[1] + ws_lane / 2 * 5 * lane;
        for (int k = pl->levels; k >= 0; k--) {
            const LevelPlan& L = pl->lv[k];
            RoctxRange level_range("tw_level %d (%dx%d)", k, L.w, L.h);
            // balanced launches: 64 pairs with a 63-pair chunk are 32 + 32, not 63 + 1
            const int per = balanced_launch_pairs(hi - lo, L.chunk);
            for (int j0 = lo; j0 < hi; j0 += per) {
                const int nc = std::min(per, hi - j0);
                // flow buffers: levels >= 1 keep every pair of the batch, level 0 only the lane's current chunk
                float* flow_cur = e->flow[k] + (k == 0 ? (size_t)lane * L.chunk * 2 * L.ps : (size_t)j0 * 2 * L.ps);
                const float* flow_prev =
                    k < pl->levels ? e->flow[k + 1] + (size_t)j0 * 2 * pl->lv[k + 1].ps : nullptr;
                if (lat2) {
                    R = e->lat_R + 5 * lat_off[k];
                    // whatever this level reads and no chain kernel was left to carry goes out by itself
                    while (side_next < sideq.size() && sideq[side_next].need_level >= k) side_done(false);
                } else if (lat) {
                    R = e->lat_R + 5 * lat_off[k];
                    if (k < pl->levels && k <= e->lat_s2_max) TW_HIP(e, hipStreamWaitEvent(ls, e->lat_ev[k], 0));
                } else {
                    // levels 3 and 2 from one read of the images (tw_pyr_23): level 3's launch also writes level 2's
                    // images, behind its own in the workspace (sized for level 0: there is room), and level 2 skips its
                    // pyramid launch.  Only when both levels cover the same pairs per launch.
                    // (one launch per level for this lane: the loop is level-major, a second chunk of level 3 would
                    // overwrite the first chunk's level-2 images before level 2 reads them)
                    const bool f23 = pl->fused23 && e->pyr_fused && pl->lv[3].chunk >= hi - lo && pl->lv[2].chunk >= hi - lo &&
                                     (size_t)(pl->lv[3].ps + pl->lv[2].ps) * 2 * nc <= ws_lane;
                    float* I2side = f23 ? I + (size_t)pl->lv[3].ps * 2 * nc : nullptr;  // (a plan may have fewer than 4 levels: ADVICE r5)
                    // levels 1 and 0 likewise (tw_pyr_k3f): level 0's images wait in the M1 region, which nothing else
                    // touches while levels 1 and 0 both run tw_flow_iter (their iterations ping-pong through M0 only) and
                    // which is large enough (5 planes per pair against 2 images of one)
                    const bool f01 = e->pyr_fused && pyr01_fusable(pl) && !e->pyr_generic && pl->lv[1].chunk >= hi - lo &&
                                     pl->lv[0].chunk >= hi - lo && level_mfree(0, nc) && level_mfree(1, nc);
                    float* I0side = M1;
                    if (f23 && k == 3) launch_pyr23(e, ls, pl, e->d_ptrs + 2 * j0, stride, I, I2side, 2 * nc);
                    else if (f01 && k == 1) launch_pyr01(e, ls, pl, e->d_ptrs + 2 * j0, stride, I, I0side, 2 * nc);
                    else if (!(f23 && k == 2) && !(f01 && k == 0)) launch_pyr(e, ls, pl, k, e->d_ptrs + 2 * j0, stride, I, 2 * nc);
                    const float* Isrc = (f23 && k == 2) ? I2side : (f01 && k == 0) ? I0side : I;
                    if ((r = launch_polyexp(e, ls, L.w, L.h, L.ld, L.ps, Isrc, R, 2 * nc, k))) return r;
                }
                // M-free iterations (tw_flow_iter; TW_MFREE): the level's iterations read the previous flow and recompute M
                // inside the kernel — no FarnebackUpdateMatrices launch, no M planes.  Flows ping-pong between the level's
                // flow buffer and the (now unused) M0 workspace so that the last iteration lands in the flow buffer.
                // scan-fused final iteration (option TW_OPT_SCAN_FUSED_FINAL): nothing but the span grid of the last
                // level-0 flow is read afterwards, so the last window average + solve runs at the grid points only
                const bool grid_only = k == 0 && scan_fused_level0(e, c.span);
                bool iterated = false;
                bool grid_stored = false;  // the last window launch wrote the span-grid samples itself (single pair)
                // (with the scan-fused last iteration: it - 1 iterations here, then the M of the last flow from
                // tw_update_matrices<false> into M1 — tw_blur_grid evaluates the window average + solve at the span-grid
                // points from it; a single iteration has no flow of this level to start from and takes the old launches)
                if (level_mfree(k, nc)) {
                    const int nfi = grid_only ? it - 1 : it;
                    float* buf[2] = {flow_cur, M0};  // iteration i writes buf[(nfi - 1 - i) & 1]: the last one the flow buffer
                    FlowUps ups;
                    if (k < pl->levels) {
                        const LevelPlan& Pv = pl->lv[k + 1];
                        ups = FlowUps{flow_prev, Pv.w, Pv.h, Pv.ld, Pv.ps, L.d_uxofs, L.d_uyofs, L.d_ualpha, L.d_ubeta, L.uxmax,
                                      (float)(1. / e->p.pyrScale)};
                    }
                    for (int i = 0; i < nfi; i++) {
                        float* out = buf[(nfi - 1 - i) & 1];
                        const float* in = i == 0 ? nullptr : buf[(nfi - i) & 1];
                        launch_flow_iter(e, ls, L.w, L.h, L.ld, L.ps, R, in, L.ps, out, L.ps,
                                         (i == 0 && k < pl->levels) ? &ups : nullptr, nc, k);
                    }
                    if (grid_only) {
                        UpdArgs u;
                        memset(&u, 0, sizeof(u));
                        u.R = R;
                        u.flow = flow_cur;
                        u.M = M1;
                        u.w = L.w;
                        u.h = L.h;
                        u.ld = L.ld;
                        u.ps = L.ps;
                        u.fps = L.ps;
                        {
                            ProfScope pscope(e, ls, TW_K_UPDATE_MATRICES, k);
                            launch_upd_kernel<false>(e, ls, L.w, L.h, nc, u);
                        }
                        BlurGridArgs g;
                        g.Min = M1;
                        g.w = L.w;
                        g.h = L.h;
                        g.ld = L.ld;
                        g.ps = L.ps;
                        g.gw = (L.w + c.span - 1) / c.span;
                        g.gh = (L.h + c.span - 1) / c.span;
                        g.gzs = (long long)g.gw * g.gh;
                        g.g = e->d_grid + (size_t)j0 * g.gw * g.gh;
                        g.c = e->wc;
                        ProfScope pscope(e, ls, TW_K_BLUR_SOLVE, 0);
                        TW_LAUNCH(e, TW_DF_BLUR_GRID, (tw_blur_grid<15, 10, 2>), dim3((g.gw + 21) / 22, (g.gh + 1) / 2, nc), dim3(256), 0, ls, g);
                    }
                    iterated = true;
                }
                if (!iterated) {
                    if (lat2 && k >= 1 && side_ok(k, false)) {
                        bool used = false;
                        launch_update(e, ls, pl, k, R, flow_cur, flow_prev, M0, nc, side_peek(), &used);
                        side_done(used);
                    } else {
                        launch_update(e, ls, pl, k, R, flow_cur, flow_prev, M0, nc);
                    }
                }
                for (int i = 0; i < it && !iterated; i++) {
                    if (grid_only && i == it - 1) {
                        BlurGridArgs g;
                        g.Min = (i & 1) ? M1 : M0;
                        g.w = L.w;
                        g.h = L.h;
                        g.ld = L.ld;
                        g.ps = L.ps;
                        g.gw = (L.w + c.span - 1) / c.span;
                        g.gh = (L.h + c.span - 1) / c.span;
                        g.gzs = (long long)g.gw * g.gh;
                        g.g = e->d_grid + (size_t)j0 * g.gw * g.gh;
                        g.c = e->wc;
                        ProfScope pscope(e, ls, TW_K_BLUR_SOLVE, 0);
                        TW_LAUNCH(e, TW_DF_BLUR_GRID, (tw_blur_grid<15, 10, 2>), dim3((g.gw + 21) / 22, (g.gh + 1) / 2, nc), dim3(256), 0, ls, g);
                        break;
                    }
                    if (lat2 && k >= 1 && side_ok(k, true)) {
                        bool used = false;
                        launch_blur(e, ls, L.w, L.h, L.ld, L.ps, (i & 1) ? M1 : M0, (i & 1) ? M0 : M1, flow_cur, R, i < it - 1, k, nc,
                                    nullptr, side_peek(), &used);
                        side_done(used);
                        continue;
                    }
                    if (lat && k == 0 && i == it - 1 && c.span > 0 && !grid_only) {
                        // the launch that stores the final flow stores its span-grid samples too: no tw_span_gather launch
                        GridOut go{e->d_grid + (size_t)j0 * ((L.w + c.span - 1) / c.span) * ((L.h + c.span - 1) / c.span), c.span,
                                   (L.w + c.span - 1) / c.span, (L.h + c.span - 1) / c.span, &grid_stored};
                        launch_blur(e, ls, L.w, L.h, L.ld, L.ps, (i & 1) ? M1 : M0, (i & 1) ? M0 : M1, flow_cur, R, 0, k, nc,
                                    e->Vd ? e->Vd + ws_lane / 2 * 5 * lane : nullptr, nullptr, nullptr, &go, lat_quads);
                        continue;
                    }
                    launch_blur(e, ls, L.w, L.h, L.ld, L.ps, (i & 1) ? M1 : M0, (i & 1) ? M0 : M1, flow_cur, R,
                                i < it - 1, k, nc, e->Vd ? e->Vd + ws_lane / 2 * 5 * lane : nullptr, nullptr, nullptr, nullptr, lat_quads);
                }
                if (k == 0 && c.span > 0 && !grid_only && !grid_stored) {
                    // grid samples of this chunk -> dense per-pair buffer (the ordered scan runs once per batch).
                    // (A scan that samples the flow planes itself — tw_span_scan<true>, TW_SCAN_DIRECT=1 of the variants library — saves this launch and
                    // costs more than it: one workgroup's 41 472 scattered dwords are 19.5 us on one CU's memory pipe against
                    // 4.8 + 7.3 us for the gather on 81 workgroups + the scan; gpurun_out/r7lat)
                    GatherArgs g;
                    g.flow = flow_cur;
                    g.fzs = 2 * L.ps;
                    g.fps = L.ps;
                    g.ld = L.ld;
                    g.span = c.span;
                    g.gw = (L.w + c.span - 1) / c.span;
                    g.gh = (L.h + c.span - 1) / c.span;
                    g.g = e->d_grid + (size_t)j0 * g.gw * g.gh;
                    ProfScope pscope(e, ls, TW_K_SCAN, 0);
                    TW_LAUNCH(e, TW_DF_SPAN_GATHER, tw_span_gather, dim3((g.gw + 63) / 64, (g.gh + 3) / 4, nc), dim3(256), 0, ls, g);
                }
                // the image-only work two levels below goes out once this level's chain is enqueued
                if (lat && !lat2 && k >= 2 && (r = lat_images(k - 2))) return r;
            }
        }
    }
    if (nlanes > 1) {
        TW_HIP(e, hipEventRecord(e->ev_join, e->stream2));
        TW_HIP(e, hipStreamWaitEvent(st, e->ev_join, 0));
    }
    return TW_OK;
    };  // enqueue_levels

    // Single pair (BASELINE config 2): ~26 launches and ~10 event operations, half of them kernels of 5-10 us.  With
    // TW_LAT_GRAPH=1 the whole DAG — both streams, their hand-offs — is captured once per (size, span, stride,
    // alignment, options) and replayed with one hipGraphLaunch; the ordered scan and the result copies stay outside
    // (their targets belong to the batch context).  Round 3 measured it: the host is NOT what holds the schedule back
    // (the direct enqueue takes 110 us of host time for 375 us of GPU time, TW_DEBUG_HOSTTIME), and the replayed graph
    // is slower (0.55 ms: kernels of the forked branch stretch to ~40 us quanta behind cross-queue signals), so the
    // direct path is the default and the graph an opt-in A/B switch (profiles/r03_latency.md).
    bool launched_graph = false;
    if (lat && e->lat_graph && !prof_on) {
        const GraphKey key{c.w, c.h, c.span, stride, e->img_aligned4, e->scan_fused, e->poly_f32};
        auto git = e->lat_graphs.find(key);
        if (git == e->lat_graphs.end()) {
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            bool ok = hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) == hipSuccess;
            tw_status br = TW_OK;
            if (ok) {
                br = enqueue_levels();
                ok = hipStreamEndCapture(st, &graph) == hipSuccess && graph != nullptr && br == TW_OK;
            }
            if (br != TW_OK) {
                if (graph) (void)hipGraphDestroy(graph);
                (void)hipGetLastError();
                return br;
            }
            ok = ok && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
            if (graph) (void)hipGraphDestroy(graph);
            if (ok) {
                git = e->lat_graphs.emplace(key, exec).first;
            } else {
                (void)hipGetLastError();
                e->lat_graph = 0;  // this runtime cannot capture the schedule: direct launches from now on
            }
        }
        if (git != e->lat_graphs.end()) {
            TW_HIP(e, hipGraphLaunch(git->second, st));
            launched_graph = true;
        }
    }
    if (!launched_graph) {
        static const bool host_time = getenv("TW_DEBUG_HOSTTIME") != nullptr;  // diagnostic: host cost of the enqueue
        const auto t_h0 = std::chrono::steady_clock::now();
        if ((r = enqueue_levels())) return r;
        if (host_time)
            fprintf(stderr, "twflow: enqueue of %d pair(s) %dx%d took %.1f us on the host\n", n, c.w, c.h,
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_h0).count());
    }
    if (c.span > 0) {
        const LevelPlan& L = pl->lv[0];
        ScanArgs a;
        a.g = e->d_grid;
        a.span = c.span;
        a.gw = (L.w + c.span - 1) / c.span;
        a.gh = (L.h + c.span - 1) / c.span;
        a.thr2 = c.threshold * c.threshold;
        a.count = e->d_count;
        a.rec_zs = (long long)a.gw * a.gh;
        a.rec = c.d_rec;
        a.flow = e->flow[0];
        a.fps = L.ps;
        a.ld = L.ld;
        const bool grid_only = e->p.pyrIterations > 0 && e->scan_fused && c.span == 10 && e->win_m == 15 && !e->box;
        (void)grid_only;
        ProfScope pscope(e, st, TW_K_SCAN, 0);
#ifdef TW_VARIANTS
        static const bool scan_direct = getenv("TW_SCAN_DIRECT") != nullptr;
        if (lat && !grid_only && scan_direct) {
            TW_LAUNCH(e, TW_DF_SPAN_SCAN, tw_span_scan<true>, dim3(n), dim3(1024), 0, st, a);
        } else
#endif
        {
            // a single pair: 16 segments, each workgroup counting the hits before its own (tw_span_scan_seg: -5 us of one CU's VALU time)
            const long long G = (long long)a.gw * a.gh;
            const int S = (int)(((G + 15) / 16 + 1023) / 1024 * 1024);
            const bool scan_seg = !getenv("TW_SCAN_SEG") || atoi(getenv("TW_SCAN_SEG")) != 0;  // (read per batch: a test compares the two kernels)
            if (n == 1 && lat && scan_seg && G > 2048 && S <= SCAN_IT * 1024)
                TW_LAUNCH(e, TW_DF_SPAN_SCAN_SEG, tw_span_scan_seg, dim3((unsigned)((G + S - 1) / S)), dim3(1024), 0, st, a, S);
            else
                TW_LAUNCH(e, TW_DF_SPAN_SCAN, tw_span_scan<false>, dim3(n), dim3(1024), 0, st, a);
        }
    }
    TW_HIP(e, hipEventRecord(c.ev_stop, st));
    if (c.span > 0) {
        const size_t G = (size_t)tw_grid_capacity(c.w, c.h, c.span);
        const size_t nrec = std::min((size_t)HOST_RECS, G);
        TW_HIP(e, hipMemcpyAsync(c.h_count, e->d_count, sizeof(int) * n, hipMemcpyDeviceToHost, st));
        TW_HIP(e, hipMemcpy2DAsync(c.h_rec, HOST_RECS * sizeof(ScanRec), c.d_rec, G * sizeof(ScanRec),
                                   nrec * sizeof(ScanRec), n, hipMemcpyDeviceToHost, st));
    }
    TW_HIP(e, hipGetLastError());
    TW_HIP(e, hipEventRecord(c.ev_done, st));
    c.launched = true;
    return TW_OK;
}

size_t staged_image_bytes(int w, int h) { return ((size_t)w * h + 255) / 256 * 256; }

tw_status check_dims(tw_engine* e, int width, int height)
{
    if (width < 1 || height < 1 || width > 32768 || height > 32768) {
        e->err = "bad image size";
        return TW_E_BAD_PARAMETER;
    }
    if ((long long)width * height > (1ll << 28)) {
        // the stencil kernels address a plane with 32-bit byte offsets (raw buffer loads): planes stay below 4 GiB
        // with room for the row pitch; 268 Mpixel (16384 x 16384) is far beyond any screenshot
        e->err = "image larger than 2^28 pixels";
        return TW_E_UNSUPPORTED;
    }
    return TW_OK;
}

// Page-locked host ranges this library handed out (tw_host_alloc) or was told about (tw_host_register): process-wide,
// because such memory is portable across devices and one engine may free what another one uploads from (a
// per-engine cache of hipPointerGetAttributes answers went stale exactly then, and a stale "pinned" answer turns the
// safe staged upload into a DMA from pageable memory).  Any pointer outside these ranges is treated as pageable and
// staged — always correct, at the cost of one memcpy.  No runtime query per submit (each one also cost ~1 KB of host
// memory that never came back: DESIGN.md §7 "Soak").
struct PinRegistry {
    enum Kind { ALLOC = 1, REGISTERED = 2 };  // how the range was page-locked: decides which HIP call releases it
    struct Range {
        size_t bytes;
        int kind;
    };
    std::mutex m;
    std::map<uintptr_t, Range> ranges;  // start -> bytes, kind
    void add(const void* p, size_t n, int kind)
    {
        std::lock_guard<std::mutex> lk(m);
        ranges[(uintptr_t)p] = Range{n, kind};
    }
    // removes the range that STARTS at p if it was made the same way; 0 = removed, 1 = unknown, 2 = the other kind
    // (tw_host_free on a registered block must not reach hipHostFree on the caller's malloc memory)
    int remove(const void* p, int kind)
    {
        std::lock_guard<std::mutex> lk(m);
        auto it = ranges.find((uintptr_t)p);
        if (it == ranges.end()) return 1;
        if (it->second.kind != kind) return 2;
        ranges.erase(it);
        return 0;
    }
    bool covers(const void* p, size_t n)
    {
        std::lock_guard<std::mutex> lk(m);
        auto it = ranges.upper_bound((uintptr_t)p);
        if (it == ranges.begin()) return false;
        --it;
        return (uintptr_t)p >= it->first && (uintptr_t)p + n <= it->first + it->second.bytes;
    }
};
PinRegistry& pin_registry()
{
    static PinRegistry r;
    return r;
}
bool host_range_is_page_locked(const void* p, size_t n) { return pin_registry().covers(p, n); }

size_t png_rows_bytes(int width, int height, int ch) { return (size_t)height * ((size_t)width * (size_t)ch + 1); }

// ch_a / ch_b > 0 (tw_submit_png8): that host image is `height` filtered PNG rows of 1 + width * ch bytes
tw_status submit_common(tw_engine* e, const uint8_t* h_a, const uint8_t* h_b, const void* d_a, const void* d_b,
                        int width, int height, ptrdiff_t stride, int span, double threshold, tw_ticket* ticket,
                        int ch_a = 0, int ch_b = 0)
{
    if (!e) return TW_E_BAD_PARAMETER;
    e->err.clear();
    tw_status r = check_dims(e, width, height);
    if (r) return r;
    if (span < 0 || stride < width || (!h_a && !d_a) || (!h_b && !d_b)) {
        e->err = "bad argument";
        return TW_E_BAD_PARAMETER;
    }
    const bool png = ch_a > 0 || ch_b > 0;
    if (png) {
        if (!h_a || !h_b || ch_a < 0 || ch_a > 4 || ch_b < 0 || ch_b > 4 || stride != width) {
            e->err = "bad argument";
            return TW_E_BAD_PARAMETER;
        }
        // the filter type byte of every row (ISO/IEC 15948 §9.2: 0-4): checked here, on the host, so that the kernel
        // never has to answer for a damaged stream (libpng: "bad adaptive filter value" -> imread fails)
        const uint8_t* img[2] = {h_a, h_b};
        const int chs[2] = {ch_a, ch_b};
        for (int q = 0; q < 2; q++) {
            if (!chs[q]) continue;
            const size_t rs = (size_t)width * (size_t)chs[q] + 1;
            for (int y = 0; y < height; y++)
                if (img[q][(size_t)y * rs] > 4) {
                    e->err = "bad PNG filter type";
                    return TW_E_BAD_IMAGE_FORMAT;
                }
        }
    }
    const size_t filt_need = png ? (png_rows_bytes(width, height, std::max(ch_a, ch_b)) + 255) / 256 * 256 : 0;
    TW_HIP(e, hipSetDevice(e->device));
    // host images are staged densely; device images are read in place with their own row stride
    const long long eff_stride = h_a ? (long long)width : (long long)stride;
    Ctx* c = &e->ctx[e->cur];
    // a batch is homogeneous: same size, span, threshold and row stride
    const bool open = !c->launched && !c->jobs.empty();
    // (a filtered image larger than the slots this batch's earlier filtered images were given starts a new batch)
    const bool fits = open && c->w == width && c->h == height && c->span == span && c->threshold == threshold &&
                      c->jobs[0].stride == eff_stride && (int)c->jobs.size() < e->cap &&
                      (!png || c->filt_slot == 0 || filt_need <= c->filt_slot);
    if (!fits) {
        if (open && (r = flush_ctx(e, *c))) {
            // the open batch cannot run (unsupported plan, out of memory): drop it as tw_wait does, so that the
            // engine stays usable; its tickets answer "unknown ticket"
            c->pending = 0;
            c->jobs.clear();
            return r;
        }
        if (c->pending > 0) {  // still owed to the caller: move on to the next context
            const int nxt = (e->cur + 1) % NCTX;
            if (e->ctx[nxt].pending > 0) return TW_E_BUSY;
            e->cur = nxt;
            c = &e->ctx[nxt];
        }
        c->jobs.clear();
        c->launched = false;
        c->pending = 0;
        c->w = width;
        c->h = height;
        c->span = span;
        c->threshold = threshold;
        c->any_host = false;
        c->any_png = false;
        c->filt_slot = 0;
        c->nseg = 0;
        c->first_ticket = e->next_ticket;
    }
    Job jb;
    jb.stride = eff_stride;
    if (h_a) {
        const size_t npx = staged_image_bytes(width, height);
        const size_t need = npx * 2 * e->cap;
        const size_t j = c->jobs.size();
        if (need > c->d_img_cap) {
            // only ever happens on the first job of a batch (all jobs of a batch have one size)
            TW_HIP(e, hipStreamSynchronize(e->copy_stream));
            if (c->d_img) (void)hipFree(c->d_img);
            c->d_img = nullptr;
            c->d_img_cap = 0;
            TW_HIP(e, hipMalloc((void**)&c->d_img, need + 256));
            c->d_img_cap = need;
        }
        uint8_t* dst_a = c->d_img + npx * (2 * j);
        uint8_t* dst_b = c->d_img + npx * (2 * j + 1);
        if (png) {
            // tw_submit_png8: filtered rows go to the batch's d_filt slots (tw_png_unfilter writes d_img when the batch
            // is launched), plain gray images of the pair straight to d_img
            if (c->filt_slot == 0) {
                // first filtered image of this batch: slots as large as the region allows, at least what it needs
                const size_t have = c->d_filt_cap / (2 * (size_t)e->cap) / 256 * 256;
                const size_t slot = std::max(filt_need, have);
                if (slot * 2 * (size_t)e->cap > c->d_filt_cap) {
                    TW_HIP(e, hipStreamSynchronize(e->copy_stream));
                    if (c->d_filt_raw) (void)hipFree(c->d_filt_raw);
                    c->d_filt = c->d_filt_raw = nullptr;
                    c->d_filt_cap = 0;
                    // 256 bytes of slack on either side: tw_png_unfilter fetches whole 16-byte groups and may read a few
                    // bytes before the first and after the last row of an image
                    TW_HIP(e, hipMalloc((void**)&c->d_filt_raw, slot * 2 * (size_t)e->cap + 512));
                    c->d_filt = c->d_filt_raw + 256;
                    c->d_filt_cap = slot * 2 * (size_t)e->cap;
                }
                // (only now — a failed synchronise / allocation above leaves filt_slot at 0, so that the next
                // filtered image of this batch tries again instead of copying to a null region)
                c->filt_slot = slot;
            }
            if (!c->d_filt) {
                e->err = "tw_submit_png8: no device region for the filtered rows";
                return TW_E_NOMEM;
            }
            const uint8_t* src[2] = {h_a, h_b};
            const int chs[2] = {ch_a, ch_b};
            uint8_t* gray_dst[2] = {dst_a, dst_b};
            for (int q = 0; q < 2; q++) {
                const size_t nb = chs[q] ? png_rows_bytes(width, height, chs[q]) : (size_t)width * height;
                uint8_t* dst = chs[q] ? c->d_filt + c->filt_slot * (2 * j + q) : gray_dst[q];
                if (pin_registry().covers(src[q], nb)) {
                    TW_HIP(e, hipMemcpyAsync(dst, src[q], nb, hipMemcpyHostToDevice, e->copy_stream));
                    continue;
                }
                // pageable memory: the caller may reuse it as soon as this call returns, and the runtime is never handed
                // a pointer it would have to page-lock on the fly — the image goes through the engine's own pinned bounce
                // buffer and the copy is over before the next one starts (the host layer hands over page-locked arenas;
                // this is the slow, always-correct path)
                if (nb > e->h_bounce_cap) {
                    if (e->h_bounce) (void)hipHostFree(e->h_bounce);
                    e->h_bounce = nullptr;
                    e->h_bounce_cap = 0;
                    TW_HIP(e, hipHostMalloc((void**)&e->h_bounce, nb, hipHostMallocDefault));
                    e->h_bounce_cap = nb;
                }
                memcpy(e->h_bounce, src[q], nb);
                TW_HIP(e, hipMemcpyAsync(dst, e->h_bounce, nb, hipMemcpyHostToDevice, e->copy_stream));
                TW_HIP(e, hipStreamSynchronize(e->copy_stream));
            }
            jb.png_ch[0] = ch_a;
            jb.png_ch[1] = ch_b;
            c->any_png = true;
        } else {
        const size_t span_bytes = (size_t)stride * (size_t)(height - 1) + (size_t)width;
        if (pin_registry().covers(h_a, span_bytes) && pin_registry().covers(h_b, span_bytes)) {
            // page-locked caller memory (tw_host_alloc / tw_host_register): DMA straight from it.  The caller keeps
            // the buffers unchanged until tw_wait() of this ticket returns.
            if (stride == width) {
                // dense rows: plain 1-D transfers (the DMA engines; a pitched 2-D copy may run as a blit kernel on
                // the CUs the flow kernels are using)
                TW_HIP(e, hipMemcpyAsync(dst_a, h_a, (size_t)width * height, hipMemcpyHostToDevice, e->copy_stream));
                TW_HIP(e, hipMemcpyAsync(dst_b, h_b, (size_t)width * height, hipMemcpyHostToDevice, e->copy_stream));
            } else {
                TW_HIP(e, hipMemcpy2DAsync(dst_a, (size_t)width, h_a, (size_t)stride, (size_t)width, (size_t)height,
                                           hipMemcpyHostToDevice, e->copy_stream));
                TW_HIP(e, hipMemcpy2DAsync(dst_b, (size_t)width, h_b, (size_t)stride, (size_t)width, (size_t)height,
                                           hipMemcpyHostToDevice, e->copy_stream));
            }
        } else {
            if (need > c->h_img_cap) {
                if (c->h_img) (void)hipHostFree(c->h_img);
                c->h_img = nullptr;
                c->h_img_cap = 0;
                TW_HIP(e, hipHostMalloc((void**)&c->h_img, need, hipHostMallocDefault));
                c->h_img_cap = need;
            }
            uint8_t* da = c->h_img + npx * (2 * j);
            uint8_t* db = c->h_img + npx * (2 * j + 1);
            for (int y = 0; y < height; y++) {
                memcpy(da + (size_t)y * width, h_a + (size_t)y * stride, width);
                memcpy(db + (size_t)y * width, h_b + (size_t)y * stride, width);
            }
            // both images of the pair in one transfer, queued now: it overlaps the caller's next submit and
            // whatever the compute stream is still doing for the previous batch
            TW_HIP(e, hipMemcpyAsync(dst_a, da, npx * 2, hipMemcpyHostToDevice, e->copy_stream));
        }
        }  // !png
        jb.h_a = h_a;
        jb.h_b = h_b;
        c->any_host = true;
        e->copy_ops++;
    } else {
        jb.d_a = (const uint8_t*)d_a;
        jb.d_b = (const uint8_t*)d_b;
    }
    if (h_a && !png && e->ramp && e->cap >= 64 && c->nseg < 2) {
        // cold-start ramp: mark the copy stream behind the first quarter / half of a full batch's uploads
        // (before the job is booked: a failure here leaves the batch as it was)
        const int sz = (int)c->jobs.size() + 1;
        if (sz == e->cap / 4 || sz == e->cap / 2) {
            TW_HIP(e, hipEventRecord(c->ev_seg[c->nseg], e->copy_stream));
            c->seg_at[c->nseg++] = sz;
        }
    }
    c->jobs.push_back(jb);
    c->pending++;
    if (ticket) *ticket = e->next_ticket;
    e->next_ticket++;
    if ((int)c->jobs.size() == e->cap && (r = flush_ctx(e, *c))) {
        c->pending = 0;  // as above: a batch that cannot be launched is dropped, not retried by every later submit
        c->jobs.clear();
        return r;
    }
    return TW_OK;
}

Ctx* find_ctx(tw_engine* e, tw_ticket t, int* idx)
{
    for (Ctx& c : e->ctx) {
        const int64_t j = t - c.first_ticket;
        if (c.pending > 0 && j >= 0 && j < (int64_t)c.jobs.size() && !c.jobs[j].waited) {
            *idx = (int)j;
            return &c;
        }
    }
    return nullptr;
}

}  // namespace

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" {

void tw_default_params(tw_params* p)
{
    if (!p) return;
    p->pyrScale = 0.5;
    p->pyrLevels = 3;
    p->winSize = 30;
    p->pyrIterations = 3;
    p->polyN = 7;
    p->polySigma = 1.5;
    p->flags = 256;
}

int tw_abi_version(void) { return TWFLOW_ABI_VERSION; }

int tw_has_variants(void)
{
#ifdef TW_VARIANTS
    return 1;
#else
    return 0;
#endif
}

int tw_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

tw_status tw_device_pci_bus_id(int device, char* buf, int cap)
{
    if (!buf || cap < 16) return TW_E_BAD_PARAMETER;
    buf[0] = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return TW_E_DEVICE;
    if (hipDeviceGetPCIBusId(buf, cap, device) != hipSuccess) {
        (void)hipGetLastError();
        buf[0] = 0;
        return TW_E_DEVICE;
    }
    for (char* c = buf; *c; c++)
        if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');  // sysfs names are lower case
    return TW_OK;
}

const char* tw_strerror(tw_status s)
{
    switch (s) {
        case TW_OK: return "OK";
        case TW_E_BAD_PARAMETER: return "BadParameter";
        case TW_E_BAD_IMAGE_FORMAT: return "BadImageFormat";
        case TW_E_DONT_MATCH_SIZE: return "Don't match image size";
        case TW_E_DEVICE: return "HIP device error";
        case TW_E_NOMEM: return "out of device memory";
        case TW_E_UNSUPPORTED: return "unsupported parameter";
        case TW_E_BUSY: return "all slots busy";
    }
    return "unknown";
}

const char* tw_last_error(const tw_engine* e) { return e ? e->err.c_str() : ""; }

int tw_grid_capacity(int width, int height, int span)
{
    if (span <= 0 || width <= 0 || height <= 0) return 0;
    return ((height + span - 1) / span) * ((width + span - 1) / span);
}

tw_status tw_engine_create(int device, const tw_params* params, int slots, tw_engine** out)
{
    if (!out) return TW_E_BAD_PARAMETER;
    *out = nullptr;
    tw_params p;
    if (params) p = *params;
    else tw_default_params(&p);
    if (!(p.pyrScale < 1) || !(p.pyrScale > 0)) return TW_E_UNSUPPORTED;  // CV_Assert(pyr_scale < 1)
    if (p.polyN < 1 || p.polyN > 7) return TW_E_UNSUPPORTED;
    if (p.winSize < 2 || p.winSize / 2 > 32) return TW_E_UNSUPPORTED;
    if (p.pyrIterations < 0 || p.pyrLevels < 0) return TW_E_BAD_PARAMETER;
    if (slots < 1) slots = 1;
    if (slots > 256) slots = 256;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return TW_E_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return TW_E_DEVICE;
    tw_engine* e = new tw_engine();
    e->device = device;
    e->p = p;
    e->cap = slots;
    polyexp_setup(p.polyN, p.polySigma, e->pc);
    window_kernel(p.winSize, e->wc);
    e->win_m = p.winSize / 2;
    e->box = (p.flags & 256) ? 0 : 1;
    if (const char* ev = getenv("TW_BLUR_VARIANT")) e->blur_variant = atoi(ev);
    if (const char* ev = getenv("TW_PYR_GENERIC")) e->pyr_generic = atoi(ev);
    if (const char* ev = getenv("TW_PYR_FUSED")) e->pyr_fused = atoi(ev) != 0;
    if (const char* ev = getenv("TW_BLUR_CM")) e->blur_cm = atoi(ev);
    if (const char* ev = getenv("TW_PYR23_THREADS")) e->pyr23_threads = atoi(ev) == 256 ? 256 : atoi(ev) == 1024 ? 1024 : atoi(ev) == 512 ? 512 : 0;
    if (const char* ev = getenv("TW_MFREE")) e->mfree = atoi(ev);
    if (const char* ev = getenv("TW_MFREE_MIN_PX")) e->mfree_min_px = atoll(ev);
    if (const char* ev = getenv("TW_MFREE_MIN_W")) e->mfree_min_w = std::max(31, atoi(ev));
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) e->cu_count = cus;
    }
    if (e->pyr_generic) e->pyr_fused = 0;
    if (const char* ev = getenv("TW_POLY_VARIANT")) e->poly_variant = atoi(ev);
    if (const char* ev = getenv("TW_BLUR_SMALL")) e->blur_small = atoi(ev);
    if (const char* ev = getenv("TW_BLUR_SMALL_LEVELS")) {
        int k = 0;
        for (const char* q = ev; *q && k < 8; k++) {
            e->blur_small_lv[k] = atoi(q);
            while (*q && *q != ',') q++;
            if (*q == ',') q++;
        }
    }
    if (const char* ev = getenv("TW_PP_WAVES")) e->pp_waves = atoi(ev);
    if (const char* ev = getenv("TW_BLUR_NOMASK")) e->blur_nomask = atoi(ev);
    if (const char* ev = getenv("TW_BLUR_PIPE")) e->blur_pipe = atoi(ev);
    if (const char* ev = getenv("TW_POLYEXP_F32")) e->poly_f32 = atoi(ev) == 2 ? 2 : (atoi(ev) ? 1 : 0);  // = TW_OPT_POLYEXP_F32 (measurement runs)
#ifndef TW_VARIANTS
    // The measured-slower A/B kernels are compiled only into `make VARIANTS=1` builds (libtwflow_variants.so): a
    // default build refuses their switches instead of silently running something else.
    {
        bool bad_lv = false;
        for (int v : e->blur_small_lv) bad_lv = bad_lv || v == 2 || v == 3 || v == 5;
        const bool bad = bad_lv || (e->blur_variant != 4) || (e->poly_variant != 1) || (e->blur_pipe != 0) || (e->blur_small == 2 || e->blur_small == 3 ||
                         e->blur_small == 5) || (getenv("TW_UPD_NY") && atoi(getenv("TW_UPD_NY")) == 1);
        if (bad) {
            fprintf(stderr, "twflow: TW_BLUR_VARIANT / TW_POLY_VARIANT / TW_BLUR_PIPE / TW_BLUR_SMALL=2,3,5 / TW_UPD_NY=1 select kernels "
                            "that are only in a VARIANTS=1 build (tidal-wave_amd/libtwflow_variants.so)\n");
            delete e;
            return TW_E_UNSUPPORTED;
        }
    }
#endif
    if (const char* ev = getenv("TW_DEBUG_STAMPS"))
        if (atoi(ev) && hipMalloc((void**)&e->dbg_stamps, 4096 * sizeof(unsigned long long)) == hipSuccess)
            (void)hipMemset(e->dbg_stamps, 0, 4096 * sizeof(unsigned long long));
    if (const char* ev = getenv("TW_UPD_NY")) e->upd_ny = atoi(ev) == 1 ? 1 : 2;
    if (const char* ev = getenv("TW_LANES")) e->lanes = std::min(2, std::max(1, atoi(ev)));
    if (const char* ev = getenv("TW_LATENCY_STREAMS")) e->lat_streams = atoi(ev) ? 1 : 0;
    if (const char* ev = getenv("TW_LATENCY_MIN_PX")) e->lat_min_px = atoll(ev);
    if (const char* ev = getenv("TW_LAT_GRAPH")) e->lat_graph = atoi(ev) ? 1 : 0;
    if (const char* ev = getenv("TW_LAT_FUSED")) e->lat_fused = atoi(ev) ? 1 : 0;
#ifdef TW_VARIANTS
    if (const char* ev = getenv("TW_FI_NT")) e->fi_nt = atoi(ev) == 512 ? 512 : 1024;
    if (const char* ev = getenv("TW_FI_SKIP")) e->fi_skip = atoi(ev);
#endif
    if (const char* ev = getenv("TW_LAT_S2_LEVELS")) e->lat_s2_max = atoi(ev);
    if (const char* ev = getenv("TW_RAMP")) e->ramp = atoi(ev) ? 1 : 0;
    if (const char* ev = getenv("TW_FI_MAXSEG")) e->fi_maxseg = std::min(64, std::max(1, atoi(ev)));
    if (const char* ev = getenv("TW_FI_MINSTEPS")) e->fi_minsteps = std::max(2, atoi(ev));
    // the main stream carries the dependent flow chain: highest priority, so that its small launches are not queued
    // behind the second stream's image-only work
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    bool ok = hipStreamCreateWithPriority(&e->stream, hipStreamNonBlocking, prio_hi) == hipSuccess &&
              hipStreamCreateWithPriority(&e->stream2, hipStreamNonBlocking, e->lanes > 1 ? prio_hi : prio_lo) == hipSuccess &&
              hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming) == hipSuccess &&
              hipMalloc((void**)&e->d_ptrs, sizeof(void*) * 2 * slots + 256) == hipSuccess &&
              hipMalloc((void**)&e->d_count, sizeof(int) * slots + 256) == hipSuccess;
    for (Ctx& c : e->ctx) {
        ok = ok && hipEventCreate(&c.ev_start) == hipSuccess && hipEventCreate(&c.ev_stop) == hipSuccess &&
             hipEventCreateWithFlags(&c.ev_done, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c.ev_h2d, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c.ev_seg[0], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c.ev_seg[1], hipEventDisableTiming) == hipSuccess &&
             hipHostMalloc((void**)&c.h_ptrs, sizeof(void*) * 2 * slots, hipHostMallocDefault) == hipSuccess &&
             hipHostMalloc((void**)&c.h_count, sizeof(int) * slots, hipHostMallocDefault) == hipSuccess &&
             hipHostMalloc((void**)&c.h_png, sizeof(PngJob) * 2 * (size_t)slots, hipHostMallocDefault) == hipSuccess &&
             hipMalloc((void**)&c.d_png, sizeof(PngJob) * 2 * (size_t)slots + 256) == hipSuccess &&
             hipHostMalloc((void**)&c.h_rec, sizeof(ScanRec) * HOST_RECS * (size_t)slots, hipHostMallocDefault) ==
                 hipSuccess;
    }
    if (!ok) {
        tw_engine_destroy(e);
        return TW_E_DEVICE;
    }
    *out = e;
    return TW_OK;
}

void tw_engine_destroy(tw_engine* e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    drop_graphs(e);
    if (e->I) (void)hipFree(e->I);
    if (e->R) (void)hipFree(e->R);
    if (e->M[0]) (void)hipFree(e->M[0]);
    if (e->M[1]) (void)hipFree(e->M[1]);
    for (float* f : e->flow)
        if (f) (void)hipFree(f);
    if (e->d_ptrs) (void)hipFree((void*)e->d_ptrs);
    if (e->d_count) (void)hipFree(e->d_count);
    if (e->d_grid) (void)hipFree(e->d_grid);
    if (e->Vd) (void)hipFree(e->Vd);
    if (e->h_bounce) (void)hipHostFree(e->h_bounce);
    if (e->dbg_stamps) (void)hipFree(e->dbg_stamps);
    if (e->lat_I) (void)hipFree(e->lat_I);
    if (e->lat_R) (void)hipFree(e->lat_R);
    for (hipEvent_t ev : e->lat_ev) (void)hipEventDestroy(ev);
    for (Ctx& c : e->ctx) {
        if (c.h_img) (void)hipHostFree(c.h_img);
        if (c.d_img) (void)hipFree(c.d_img);
        if (c.d_rec) (void)hipFree(c.d_rec);
        if (c.d_filt_raw) (void)hipFree(c.d_filt_raw);
        if (c.d_png) (void)hipFree(c.d_png);
        if (c.h_png) (void)hipHostFree(c.h_png);
        if (c.ev_h2d) (void)hipEventDestroy(c.ev_h2d);
        for (hipEvent_t ev : c.ev_seg)
            if (ev) (void)hipEventDestroy(ev);
        if (c.h_ptrs) (void)hipHostFree((void*)c.h_ptrs);
        if (c.h_count) (void)hipHostFree(c.h_count);
        if (c.h_rec) (void)hipHostFree(c.h_rec);
        if (c.ev_start) (void)hipEventDestroy(c.ev_start);
        if (c.ev_stop) (void)hipEventDestroy(c.ev_stop);
        if (c.ev_done) (void)hipEventDestroy(c.ev_done);
    }
    if (e->stream) (void)hipStreamDestroy(e->stream);
    if (e->stream2) (void)hipStreamDestroy(e->stream2);
    if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    for (auto& kv : e->plans) free_plan(kv.second);
    for (auto& pend : e->prof_pending)
        for (ProfPair& pp : pend) {
            (void)hipEventDestroy(pp.a);
            (void)hipEventDestroy(pp.b);
        }
    for (hipEvent_t ev : e->prof_free) (void)hipEventDestroy(ev);
    delete e;
}

tw_status tw_submit_u8(tw_engine* e, const uint8_t* expect, const uint8_t* target, int width, int height,
                       ptrdiff_t stride, int span, double threshold, tw_ticket* ticket)
{
    if (!expect || !target) return TW_E_BAD_PARAMETER;
    return submit_common(e, expect, target, nullptr, nullptr, width, height, stride, span, threshold, ticket);
}

tw_status tw_submit_png8(tw_engine* e, const uint8_t* expect, int expect_channels, const uint8_t* target,
                         int target_channels, int width, int height, int span, double threshold, tw_ticket* ticket)
{
    if (expect_channels == 0 && target_channels == 0)  // two plain gray images: the ordinary host path
        return submit_common(e, expect, target, nullptr, nullptr, width, height, width, span, threshold, ticket);
    return submit_common(e, expect, target, nullptr, nullptr, width, height, width, span, threshold, ticket,
                         expect_channels, target_channels);
}

tw_status tw_submit_dev(tw_engine* e, const void* d_expect, const void* d_target, int width, int height,
                        ptrdiff_t stride, int span, double threshold, tw_ticket* ticket)
{
    if (!d_expect || !d_target) return TW_E_BAD_PARAMETER;
    return submit_common(e, nullptr, nullptr, d_expect, d_target, width, height, stride, span, threshold, ticket);
}

tw_status tw_flush(tw_engine* e)
{
    if (!e) return TW_E_BAD_PARAMETER;
    TW_HIP(e, hipSetDevice(e->device));
    return flush_ctx(e, e->ctx[e->cur]);
}

tw_status tw_wait(tw_engine* e, tw_ticket ticket, tw_vector* out, int cap, int* n, float* seconds)
{
    if (!e) return TW_E_BAD_PARAMETER;
    int j = 0;
    Ctx* c = find_ctx(e, ticket, &j);
    if (!c) {
        e->err = "unknown ticket";
        return TW_E_BAD_PARAMETER;
    }
    TW_HIP(e, hipSetDevice(e->device));
    tw_status r;
    if (!c->launched && (r = flush_ctx(e, *c))) {
        // the batch cannot run: drop it so that the engine stays usable
        c->pending = 0;
        c->jobs.clear();
        return r;
    }
    hipError_t herr = hipEventSynchronize(c->ev_done);
    if (c->any_host && c->pending == (int)c->jobs.size()) {
        // Once per batch the host synchronises the copy stream.  Nothing else ever waits on it from the host (the
        // compute stream does, through ev_h2d), and this runtime releases a stream's per-command bookkeeping only on
        // a host-side hipStreamSynchronize: without it the process grew by ~1 KB per uploaded image (found by a
        // 300 000-pair soak; an event wait or a stream query does not release it).  The uploads it waits for are this
        // batch's (long done) and, at most, the next batch's, which that batch needs before it can start anyway.
        // Round 4: a caller that keeps two batches in flight has the NEXT batches' uploads (and their scanline
        // reconstruction) queued there — 4-6 ms of waiting per batch for nothing.  So: synchronise when the stream is
        // idle anyway (free), and otherwise only every 16th batch (the bookkeeping stays bounded at ~1 MB).
        // (the engine counts what it queues there itself: a hipStreamQuery per batch leaves bookkeeping of its own behind)
        if (e->copy_ops == c->copy_ops_at_flush || ++e->copy_sync_skipped >= 16) {
            (void)hipStreamSynchronize(e->copy_stream);
            e->copy_sync_skipped = 0;
        }
    }
    c->jobs[j].waited = true;
    c->pending--;
    if (herr != hipSuccess) {
        e->err = std::string("hipEventSynchronize: ") + hipGetErrorString(herr);
        return TW_E_DEVICE;
    }
    if (seconds) {
        // device compute time of the batch, shared equally by its pairs (exact for a batch of one)
        float ms = 0.f;
        TW_HIP(e, hipEventElapsedTime(&ms, c->ev_start, c->ev_stop));
        *seconds = ms * 1e-3f / (float)c->jobs.size();
    }
    if (c->span > 0) {
        const int cnt = c->h_count[j];
        if (n) *n = cnt;
        const int want = std::min(cnt, cap);
        const ScanRec* hr = c->h_rec + (size_t)j * HOST_RECS;
        std::vector<ScanRec> extra;
        if (want > HOST_RECS && out) {
            // rare: more hits than the eager copy holds; the batch context's own record region still has them
            const size_t G = (size_t)tw_grid_capacity(c->w, c->h, c->span);
            extra.resize(want);
            TW_TRY(d2h_sync(e, extra.data(), c->d_rec + (size_t)j * G, (size_t)want * sizeof(ScanRec)));
            hr = extra.data();
        }
        if (out)
            for (int i = 0; i < want; i++) {
                out[i].x = hr[i].x;
                out[i].y = hr[i].y;
                out[i].dx = hr[i].dx;  // float widened to double, src/consumer.cpp:72-73
                out[i].dy = hr[i].dy;
            }
    } else if (n) {
        *n = 0;
    }
    return TW_OK;
}

tw_status tw_diff_u8(tw_engine* e, const uint8_t* expect, const uint8_t* target, int width, int height,
                     ptrdiff_t stride, int span, double threshold, tw_vector* out, int cap, int* n, float* seconds)
{
    if (span < 1) return TW_E_BAD_PARAMETER;
    tw_ticket t;
    tw_status r = tw_submit_u8(e, expect, target, width, height, stride, span, threshold, &t);
    if (r) return r;
    return tw_wait(e, t, out, cap, n, seconds);
}

tw_status tw_flow_u8(tw_engine* e, const uint8_t* expect, const uint8_t* target, int width, int height,
                     ptrdiff_t stride, float* flowx, float* flowy, float* seconds)
{
    if (!e) return TW_E_BAD_PARAMETER;
    // the dense flow of a pair lives in the level-0 chunk buffer: run this pair as a batch of its own
    tw_status r = tw_flush(e);
    if (r) return r;
    tw_ticket t;
    r = tw_submit_u8(e, expect, target, width, height, stride, 0, 0.0, &t);
    if (r) return r;
    r = tw_wait(e, t, nullptr, 0, nullptr, seconds);
    if (r) return r;
    Plan* pl = nullptr;
    if ((r = get_plan(e, width, height, &pl))) return r;
    const LevelPlan& L0 = pl->lv[0];
    const float* f = e->flow[0];
    // on the engine's own stream, then one host-side synchronise of it (copies on the null stream are never
    // followed by one, and this runtime keeps a little bookkeeping per command until a stream is synchronised)
    // The planes come down into the engine's own page-locked bounce buffer and are copied out by the CPU, unless the
    // caller's planes are page-locked blocks the library knows (tw_host_alloc / tw_host_register): the runtime is never
    // handed a pageable pointer it would have to page-lock on the fly (round 4: a fault on a malloc-heap address inside
    // this call, after an earlier hipHostRegister / hipHostUnregister of a neighbouring heap block).
    const size_t plane = (size_t)width * height * 4;
    float* dst[2] = {flowx, flowy};
    float* via[2] = {flowx, flowy};
    size_t need = 0;
    for (int q = 0; q < 2; q++)
        if (dst[q] && !pin_registry().covers(dst[q], plane)) need += plane;
    if (need > e->h_bounce_cap) {
        if (e->h_bounce) (void)hipHostFree(e->h_bounce);
        e->h_bounce = nullptr;
        e->h_bounce_cap = 0;
        TW_HIP(e, hipHostMalloc((void**)&e->h_bounce, need, hipHostMallocDefault));
        e->h_bounce_cap = need;
    }
    size_t off = 0;
    for (int q = 0; q < 2; q++) {
        if (!dst[q]) continue;
        if (!pin_registry().covers(dst[q], plane)) {
            via[q] = (float*)(e->h_bounce + off);
            off += plane;
        }
        TW_HIP(e, hipMemcpy2DAsync(via[q], (size_t)width * 4, f + (q ? L0.ps : 0), (size_t)L0.ld * 4, (size_t)width * 4,
                                   height, hipMemcpyDeviceToHost, e->stream));
    }
    TW_HIP(e, hipStreamSynchronize(e->stream));
    for (int q = 0; q < 2; q++)
        if (dst[q] && via[q] != dst[q]) memcpy(dst[q], via[q], plane);
    return TW_OK;
}

tw_status tw_dev_alloc(tw_engine* e, size_t bytes, void** dptr)
{
    if (!e || !dptr) return TW_E_BAD_PARAMETER;
    TW_HIP(e, hipSetDevice(e->device));
    TW_HIP(e, hipMalloc(dptr, bytes));
    return TW_OK;
}
tw_status tw_dev_free(tw_engine* e, void* dptr)
{
    if (!e) return TW_E_BAD_PARAMETER;
    TW_HIP(e, hipSetDevice(e->device));
    TW_HIP(e, hipFree(dptr));
    return TW_OK;
}
tw_status tw_dev_upload(tw_engine* e, void* dptr, const void* host, size_t bytes)
{
    if (!e || !dptr || !host) return TW_E_BAD_PARAMETER;
    TW_HIP(e, hipSetDevice(e->device));
    TW_TRY(h2d_sync(e, dptr, host, bytes));
    return TW_OK;
}

tw_status tw_set_option(tw_engine* e, int option, int value)
{
    if (!e) return TW_E_BAD_PARAMETER;
    switch (option) {
        case TW_OPT_SCAN_FUSED_FINAL: e->scan_fused = value ? 1 : 0; return TW_OK;
        case TW_OPT_POLYEXP_F32: e->poly_f32 = value == 2 ? 2 : (value ? 1 : 0); return TW_OK;
        default: e->err = "unknown option"; return TW_E_BAD_PARAMETER;
    }
}

tw_status tw_host_alloc(tw_engine* e, size_t bytes, void** hptr)
{
    if (!e || !hptr) return TW_E_BAD_PARAMETER;
    TW_HIP(e, hipSetDevice(e->device));
    TW_HIP(e, hipHostMalloc(hptr, bytes, hipHostMallocPortable));  // usable by the engines of every device (one queue, N consumers)
    pin_registry().add(*hptr, bytes, PinRegistry::ALLOC);
    return TW_OK;
}
tw_status tw_host_free(tw_engine* e, void* hptr)
{
    if (!e) return TW_E_BAD_PARAMETER;
    TW_HIP(e, hipSetDevice(e->device));
    TW_HIP(e, hipStreamSynchronize(e->copy_stream));  // an upload of THIS engine may still be reading it (see twflow.h)
    if (const int why = pin_registry().remove(hptr, PinRegistry::ALLOC)) {
        e->err = why == 2 ? "tw_host_free: the block was registered (tw_host_register): release it with tw_host_unregister"
                          : "tw_host_free: not a tw_host_alloc block";
        return TW_E_BAD_PARAMETER;
    }
    TW_HIP(e, hipHostFree(hptr));
    return TW_OK;
}
tw_status tw_host_register(tw_engine* e, void* hptr, size_t bytes)
{
    if (!e || !hptr || !bytes) return TW_E_BAD_PARAMETER;
    // Whole pages only, and only pages the caller owns outright: a range that starts or ends inside a page shares
    // that page with whatever else the allocator placed there, and page-locking / releasing it behind the runtime's
    // back is what round 4's GPU fault traced to (a pageable copy of a NEIGHBOURING heap block right after such a
    // block's hipHostUnregister).  malloc / new / cv::Mat heap blocks are refused: take the memory from
    // tw_host_alloc, or from an allocation of your own that is page-aligned and a page multiple (mmap,
    // aligned_alloc(page, n * page), posix_memalign).
    const size_t page = (size_t)sysconf(_SC_PAGESIZE);
    if (((uintptr_t)hptr % page) != 0 || (bytes % page) != 0) {
        e->err = "tw_host_register: the range must start on a page boundary and be a whole number of pages (" +
                 std::to_string(page) + " bytes): register memory you own page-wise (mmap / aligned_alloc) or use tw_host_alloc";
        return TW_E_BAD_PARAMETER;
    }
    TW_HIP(e, hipSetDevice(e->device));
    TW_HIP(e, hipHostRegister(hptr, bytes, hipHostRegisterPortable));
    pin_registry().add(hptr, bytes, PinRegistry::REGISTERED);
    return TW_OK;
}
tw_status tw_host_unregister(tw_engine* e, void* hptr)
{
    if (!e || !hptr) return TW_E_BAD_PARAMETER;
    TW_HIP(e, hipSetDevice(e->device));
    TW_HIP(e, hipStreamSynchronize(e->copy_stream));
    if (const int why = pin_registry().remove(hptr, PinRegistry::REGISTERED)) {
        e->err = why == 2 ? "tw_host_unregister: the block came from tw_host_alloc: release it with tw_host_free"
                          : "tw_host_unregister: not a registered block";
        return TW_E_BAD_PARAMETER;
    }
    TW_HIP(e, hipHostUnregister(hptr));
    return TW_OK;
}

// ---- instrumentation ---------------------------------------------------------------------------------
tw_status tw_prof_select(tw_engine* e, int kclass, int level)
{
    if (!e || kclass >= TW_K_COUNT) return TW_E_BAD_PARAMETER;
    if (kclass < 0) {
        for (int i = 0; i < TW_K_COUNT; i++) e->prof_level[i] = -2;
        return TW_OK;
    }
    e->prof_level[kclass] = level < -1 ? -2 : level;
    return TW_OK;
}

tw_status tw_prof_read(tw_engine* e, int kclass, double* ms_total, int* launches)
{
    if (!e || kclass < 0 || kclass >= TW_K_COUNT) return TW_E_BAD_PARAMETER;
    TW_HIP(e, hipSetDevice(e->device));
    TW_HIP(e, hipDeviceSynchronize());
    double tot = 0;
    int cnt = 0;
    for (ProfPair& pp : e->prof_pending[kclass]) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, pp.a, pp.b) == hipSuccess) {
            tot += ms;
            cnt++;
        }
        e->prof_free.push_back(pp.a);
        e->prof_free.push_back(pp.b);
    }
    e->prof_pending[kclass].clear();
    if (ms_total) *ms_total = tot;
    if (launches) *launches = cnt;
    return TW_OK;
}

int tw_num_levels(const tw_engine* e, int width, int height)
{
    if (!e) return -1;
    return plan_levels(width, height, e->p.pyrScale, std::min(std::max(e->p.pyrLevels, 0), 60));
}

int tw_level_chunk(tw_engine* e, int width, int height, int level)
{
    if (!e) return -1;
    if (hipSetDevice(e->device) != hipSuccess) return -1;
    Plan* pl = nullptr;
    if (get_plan(e, width, height, &pl) != TW_OK || level < 0 || level > pl->levels) return -1;
    return pl->lv[level].chunk;
}

int tw_level_runs_flow_iter(const tw_engine* e, int width, int height, int level, int npairs)
{
    if (!e || npairs < 1) return -1;
    const int L = tw_num_levels(e, width, height);
    if (level < 0 || level > L) return -1;
    int w, h, ks;
    double sg, sc;
    level_geometry(width, height, e->p.pyrScale, level, &w, &h, &sg, &ks, &sc);
    const int nc = balanced_launch_pairs(npairs, pairs_per_launch(e, w, h, level));
    return level_runs_flow_iter(e, w, h, nc, single_pair_schedule(e, npairs, width, height, L),
                                level == 0 && scan_fused_level0(e, 10)) ? 1 : 0;
}

double tw_algorithmic_bytes(const tw_engine* e, int kclass, int level, int width, int height)
{
    return e ? tw_algorithmic_bytes_launch(e, kclass, level, width, height, e->cap) : 0;
}

double tw_algorithmic_bytes_launch(const tw_engine* e, int kclass, int level, int width, int height, int npairs)
{
    if (!e || npairs < 1) return 0;
    const int L = tw_num_levels(e, width, height);
    if (level < 0 || level > L) return 0;
    int w, h, ks;
    double sg, sc;
    level_geometry(width, height, e->p.pyrScale, level, &w, &h, &sg, &ks, &sc);
    const double N = (double)w * h, N0 = (double)width * height;
    const int it = e->p.pyrIterations;
    // as built (round 5), with the SAME predicates as the schedule (flush_ctx): levels that run tw_flow_iter have no
    // FarnebackUpdateMatrices launch and no M planes (flow 8N + R0 20N + R1 20N in, flow 8N out = 56N per iteration; the first
    // one reads the coarser level's flow, 8 N_{k+1}, instead of this level's); levels 3 and 2 of an exact-halving pyramid come
    // from one read of the images.  A single pair, a small batch or a narrow level keeps tw_update_matrices + tw_blur_solve*.
    const bool mfree = tw_level_runs_flow_iter(e, width, height, level, npairs) == 1;
    const bool lat = single_pair_schedule(e, npairs, width, height, L);
    auto whole_batch = [&](int k) {  // one launch covers the batch at level k
        int wk, hk;
        level_geometry(width, height, e->p.pyrScale, k, &wk, &hk, &sg, &ks, &sc);
        return pairs_per_launch(e, wk, hk, k) >= npairs;
    };
    const bool fused23 = e->pyr_fused && L >= 3 && e->p.pyrScale == 0.5 && width % 8 == 0 && height % 8 == 0 && width >= 64 &&
                         height >= 64 && (lat || (whole_batch(3) && whole_batch(2)));
    switch (kclass) {
        case TW_K_PYR: {
            if (fused23 && level == 2) return 2 * (4 * N);  // written by level 3's launch
            // levels 1 and 0 from one read (tw_pyr_k3f) when both run tw_flow_iter, level 1 is an exact halving and a launch
            // covers the batch at both levels (the single-pair schedule's twin launches carry tw_pyr_k3f as well)
            int w1 = 0, h1 = 0;
            if (L >= 1) level_geometry(width, height, e->p.pyrScale, 1, &w1, &h1, &sg, &ks, &sc);
            const bool exact01 = e->pyr_fused && !e->pyr_generic && L >= 1 && width == 2 * w1 && height == 2 * h1 && width >= 16;
            const bool fused01 = exact01 && (lat ? (e->lat_fused && L == 3 && fused23)
                                                 : (whole_batch(0) && whole_batch(1) &&
                                                    tw_level_runs_flow_iter(e, width, height, 0, npairs) == 1 &&
                                                    tw_level_runs_flow_iter(e, width, height, 1, npairs) == 1));
            if (fused01 && level == 0) return 2 * (4 * N);  // written by level 1's launch
            return 2 * (N0 + 4 * N);
        }
        case TW_K_POLYEXP: return 48 * N;  // 2 images x (4 + 20) B/px
        case TW_K_UPDATE_MATRICES: {
            if (mfree) return 0;
            // R0 20N + R1 20N read, M 20N written; the level's initial flow is read from the coarser level (8 N_{k+1})
            // and is only stored when no iteration follows (the first blur launch recomputes nothing from it: the
            // refresh of iteration 0 uses M, and the flow itself is rewritten by the first solve)
            double up = it == 0 ? 8 * N : 0;
            if (level < L) {
                int pw, ph;
                level_geometry(width, height, e->p.pyrScale, level + 1, &pw, &ph, &sg, &ks, &sc);
                up += 8.0 * pw * ph;
            }
            return 60 * N + up;
        }
        case TW_K_BLUR_SOLVE:
            // average over the `it` launches of a level.  The last one reads M (20N) and writes the flow (8N).  The
            // others are fused with the FarnebackUpdateMatrices refresh: M 20N + R0 20N + R1 20N read, M 20N
            // written = 80N — the flow they compute stays in registers and is never stored (counting
            // 28N + 68N charged 16N the kernel does not move).
            if (mfree) {
                double first = 48 * N;
                if (level < L) {
                    int pw, ph;
                    level_geometry(width, height, e->p.pyrScale, level + 1, &pw, &ph, &sg, &ks, &sc);
                    first += 8.0 * pw * ph;
                }
                return (first + (it - 1) * 56 * N) / it;
            }
            return it > 0 ? (28 * N + (it - 1) * 80 * N) / it : 0;
        default: return 0;
    }
}

// SURVEY.md 8(d) model of one pair (every named stage of the reference reads each input once and writes each output
// once: pyr 2(N0+4N), polyexp 48N, updateMatrices 68N and blur+solve 28N per iteration, flow init): the
// figure BASELINE.md prices the pair against (991.8 MB at 1080p with the default parameters).
double tw_algorithmic_bytes_pair(const tw_engine* e, int width, int height, int span)
{
    if (!e) return 0;
    const int L = tw_num_levels(e, width, height);
    const int it = e->p.pyrIterations;
    double tot = 0;
    for (int k = 0; k <= L; k++) {
        int w, h, ks;
        double sg, sc;
        level_geometry(width, height, e->p.pyrScale, k, &w, &h, &sg, &ks, &sc);
        const double N = (double)w * h, N0 = (double)width * height;
        double up = 8 * N;  // memset at the coarsest level
        if (k < L) {
            int pw, ph;
            level_geometry(width, height, e->p.pyrScale, k + 1, &pw, &ph, &sg, &ks, &sc);
            up = 8.0 * pw * ph + 8 * N;
        }
        tot += 2 * (N0 + 4 * N) + 48 * N + up + it * (68 + 28) * N;
    }
    if (span > 0) tot += 16.0 * tw_grid_capacity(width, height, span);
    return tot;
}

// What the kernels as built must move for one pair (fusion removed the flow stores between iterations): the sum of
// tw_algorithmic_bytes over all launches.  Reported beside the SURVEY figure; smaller than it.
double tw_min_traffic_bytes_pair(const tw_engine* e, int width, int height, int span)
{
    if (!e) return 0;
    const int L = tw_num_levels(e, width, height);
    const int it = e->p.pyrIterations;
    double tot = 0;
    for (int k = 0; k <= L; k++) {
        tot += tw_algorithmic_bytes(e, TW_K_PYR, k, width, height);
        tot += tw_algorithmic_bytes(e, TW_K_POLYEXP, k, width, height);
        tot += tw_algorithmic_bytes(e, TW_K_UPDATE_MATRICES, k, width, height);
        tot += it * tw_algorithmic_bytes(e, TW_K_BLUR_SOLVE, k, width, height);
    }
    if (span > 0) tot += 16.0 * tw_grid_capacity(width, height, span);
    return tot;
}

}  // extern "C"

// ---- per-stage entry points (tests) --------------------------------------------------------------------
namespace {
// (through the engine's page-locked bounce buffer, like tw_flow_u8: the runtime is never handed a pageable pointer for a
// pitched copy — round 4)
tw_status up_planes(tw_engine* e, float* d, int ld, long long ps, const float* h, int w, int hh, int planes)
{
    const size_t plane = (size_t)w * hh * 4;
    tw_status r = bounce_reserve(e, plane);
    if (r) return r;
    for (int c = 0; c < planes; c++) {
        memcpy(e->h_bounce, h + (size_t)c * w * hh, plane);
        TW_HIP(e, hipMemcpy2D(d + c * ps, (size_t)ld * 4, e->h_bounce, (size_t)w * 4, (size_t)w * 4, hh, hipMemcpyHostToDevice));
    }
    return TW_OK;
}
tw_status down_planes(tw_engine* e, float* h, const float* d, int ld, long long ps, int w, int hh, int planes)
{
    const size_t plane = (size_t)w * hh * 4;
    tw_status r = bounce_reserve(e, plane);
    if (r) return r;
    for (int c = 0; c < planes; c++) {
        TW_HIP(e, hipMemcpy2D(e->h_bounce, (size_t)w * 4, d + c * ps, (size_t)ld * 4, (size_t)w * 4, hh, hipMemcpyDeviceToHost));
        memcpy(h + (size_t)c * w * hh, e->h_bounce, plane);
    }
    return TW_OK;
}
}  // namespace


// diagnostic: the phase stamps of the last tw_pyr_taps launch (TW_DEBUG_STAMPS=1), 64 workgroups x 4 stamps
extern "C" int tw_debug_stamps(tw_engine* e, unsigned long long* out)
{
    if (!e || !e->dbg_stamps || !out) return 0;
    if (hipSetDevice(e->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return 0;
    return d2h_sync(e, out, e->dbg_stamps, 64 * 4 * sizeof(unsigned long long)) == TW_OK ? 256 : 0;
}

// the first n (<= 4096) stamps of the debug buffer: tw_flow_iter's phase stamps (variants library, TW_DEBUG_STAMPS=1):
// [workgroup 0..31][wave 0 / 9][step 40..47][stamp 0..7]
extern "C" int tw_debug_stamps_ex(tw_engine* e, unsigned long long* out, int n)
{
    if (!e || !e->dbg_stamps || !out || n < 1 || n > 4096) return 0;
    if (hipSetDevice(e->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return 0;
    return d2h_sync(e, out, e->dbg_stamps, (size_t)n * sizeof(unsigned long long)) == TW_OK ? n : 0;
}

// number of captured single-pair schedules this engine holds (tests: the graph path is really the one that ran)
extern "C" int tw_debug_graphs(tw_engine* e) { return e ? (int)e->lat_graphs.size() : -1; }

extern "C" int tw_debug_launch_counts(tw_engine* e, unsigned long long* counts, unsigned long long* last_z, int n, int reset)
{
    if (!e) return -1;
    for (int i = 0; i < n && i < TW_DF_COUNT; i++) {
        if (counts) counts[i] = e->launches[i];
        if (last_z) last_z[i] = e->last_grid_z[i];
    }
    if (reset) {
        memset(e->launches, 0, sizeof(e->launches));
        memset(e->last_grid_z, 0, sizeof(e->last_grid_z));
    }
    return TW_DF_COUNT;
}

extern "C" const char* tw_debug_family_name(int family)
{
    static const char* const names[TW_DF_COUNT] = {
        "tw_pyr_k3", "tw_pyr_k3f", "tw_pyr_23", "tw_pyr_taps", "tw_pyr_level", "tw_polyexp", "tw_update_matrices",
        "tw_flow_iter", "tw_flow_iter_ups", "tw_flow_iter_zero", "tw_blur_solve4", "tw_blur_solve4y", "tw_blur_solve8",
        "tw_blur_solve_pp", "tw_blur_solve_generic", "tw_blur_variant", "tw_blur_grid", "tw_box", "tw_twin",
        "tw_span_gather", "tw_span_scan", "tw_png_unfilter", "tw_span_scan_seg", "tw_blur_solve4q"};
    return (family >= 0 && family < TW_DF_COUNT) ? names[family] : nullptr;
}

extern "C" int tw_debug_memory(tw_engine* e, unsigned long long* out, int n)
{
    if (!e || !out) return -1;
    unsigned long long v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // device: the level workspace (I + 5 I for R + 2 x 2.5 I for M), the single-pair schedule's I / R, the box window's column
    // sums, flow planes, the dense grid, pointer / count tables, debug stamps, every context's images, filtered rows, PNG job
    // table and records, and the plans' tables (each allocation with the slack it was made with)
    if (e->I) v[0] += (e->ws_elems * 4 + 256) + (e->ws_elems * 5 * 4 + 256) + 2 * (e->ws_elems / 2 * 5 * 4 + 256);
    if (e->lat_I) v[0] += (e->lat_cap * 4 + 256) + (e->lat_cap * 5 * 4 + 256);
    if (e->Vd) v[0] += e->Vd_cap * sizeof(double) + 256;
    for (size_t k = 0; k < e->flow.size(); k++)
        if (e->flow[k]) v[0] += e->flow_cap[k] * 4 + 256;
    if (e->d_grid) v[0] += e->d_grid_cap * sizeof(float2) + 256;
    if (e->d_ptrs) v[0] += sizeof(void*) * 2 * (size_t)e->cap + 256;
    if (e->d_count) v[0] += sizeof(int) * (size_t)e->cap + 256;
    if (e->dbg_stamps) v[0] += 4096 * sizeof(unsigned long long);
    for (const Ctx& c : e->ctx) {
        if (c.d_img) v[0] += c.d_img_cap + 256;
        if (c.d_filt_raw) v[0] += c.d_filt_cap + 512;
        if (c.d_png) v[0] += sizeof(PngJob) * 2 * (size_t)e->cap + 256;
        if (c.d_rec) v[0] += c.d_rec_cap * sizeof(ScanRec) + 256;
        if (c.h_img) v[1] += c.h_img_cap;
        if (c.h_ptrs) v[1] += sizeof(void*) * 2 * (size_t)e->cap;
        if (c.h_count) v[1] += sizeof(int) * (size_t)e->cap;
        if (c.h_png) v[1] += sizeof(PngJob) * 2 * (size_t)e->cap;
        if (c.h_rec) v[1] += sizeof(ScanRec) * HOST_RECS * (size_t)e->cap;
    }
    for (const auto& kv : e->plans) v[0] += kv.second->owned_bytes;
    v[1] += e->h_bounce_cap;
    v[2] = e->plans.size();
    v[3] = e->h_bounce_cap;
    {
        PinRegistry& r = pin_registry();
        std::lock_guard<std::mutex> lk(r.m);
        v[4] = r.ranges.size();
        for (const auto& kv : r.ranges) v[5] += kv.second.bytes;
    }
    v[6] = e->prof_free.size();
    for (int i = 0; i < TW_K_COUNT; i++) v[6] += 2 * e->prof_pending[i].size();
    v[7] = e->lat_graphs.size();
    int i = 0;
    for (; i < n && i < 8; i++) out[i] = v[i];
    return i;
}

// occupancy report of the main kernels (workgroups per CU the runtime admits) — tools/kbench.py
extern "C" int tw_debug_occupancy(char* buf, int cap)
{
    int n = 0, o = 0;
    auto add = [&](const char* name, const void* fn, int threads, size_t dyn_lds) {
        int blocks = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, fn, threads, dyn_lds);
        o = snprintf(buf + n, cap - n, "%s: %d workgroups/CU of %d threads\n", name, blocks, threads);
        if (o > 0) n += o;
    };
    add("tw_blur_solve4<15,256>", (const void*)tw_blur_solve4<15, 256, 16, 8, true>, 256, 0);
#ifdef TW_VARIANTS
    add("tw_blur_solve8<15,256>", (const void*)tw_blur_solve8<15, 256, 16, 8, true, true>, 256, 0);
    add("tw_blur_solve8<15,128>", (const void*)tw_blur_solve8<15, 128, 16, 8, true, true>, 128, 0);
    add("tw_polyexp<7>", (const void*)tw_polyexp<7>, 256, 0);
    add("tw_polyexp_pk<7,16>", (const void*)tw_polyexp_pk<7, 16>, 256, 0);
#endif
    add("tw_polyexp_pk<7,8>", (const void*)tw_polyexp_pk<7, 8>, 256, 0);
    add("tw_update_matrices<true,2>", (const void*)tw_update_matrices<true, 2>, 256, 0);
    add("tw_pyr_k3<0>", (const void*)tw_pyr_k3<0>, 256, 0);
    return n;
}

// ---- the yardstick: a plain float4 device copy (SURVEY 8d "measured device-copy peak taken in the same run") -----------
// ONE 16-byte load and one 16-byte store per lane, no loop, one workgroup per 4 KiB: the shape that reaches what
// MI355X_MICROARCH.md quotes for "float4 copy" (6.29 TB/s there; 6.19 TB/s on round 5's lease, against 4.9 for a
// grid-stride loop, 4.8 for hipMemcpyDtoD and 5.5 for round 4's torch copy_: tools/ubench/copy_rate.hip,
// profiles/r05_copy_rate.txt), so that bench.py's `frac_of_measured_copy` is comparable with the guide.
__global__ __launch_bounds__(256) void tw_copy_f4(const float4* __restrict__ src, float4* __restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
extern "C" tw_status tw_debug_copy_rate(tw_engine* e, size_t bytes, int reps, double* gbps)
{
    if (!e || !gbps || bytes < 4096 || reps < 1) return TW_E_BAD_PARAMETER;
    TW_HIP(e, hipSetDevice(e->device));
    Tmp t;
    float4* a = (float4*)t.alloc<uint8_t>(bytes);
    float4* b = (float4*)t.alloc<uint8_t>(bytes);
    if (!a || !b) return TW_E_NOMEM;
    hipStream_t st = e->stream;
    TW_HIP(e, hipMemsetAsync(a, 1, bytes, st));
    const size_t n = bytes / 16;
    if ((n + 255) / 256 > 0x7fffffffu) return TW_E_BAD_PARAMETER;
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipEvent_t ea, eb;
    TW_HIP(e, hipEventCreate(&ea));
    TW_HIP(e, hipEventCreate(&eb));
    hipLaunchKernelGGL(tw_copy_f4, dim3(grid), dim3(256), 0, st, a, b, n);  // warm-up
    (void)hipEventRecord(ea, st);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(tw_copy_f4, dim3(grid), dim3(256), 0, st, a, b, n);
    (void)hipEventRecord(eb, st);
    hipError_t err = hipStreamSynchronize(st);
    float ms = 0.f;
    if (err == hipSuccess) err = hipEventElapsedTime(&ms, ea, eb);
    (void)hipEventDestroy(ea);
    (void)hipEventDestroy(eb);
    TW_HIP(e, err);
    TW_HIP(e, hipGetLastError());
    *gbps = 2.0 * (double)(n * 16) * reps / (ms * 1e-3) / 1e9;  // read + write counted
    return TW_OK;
}

// ---- isolated kernel timing on synthetic device data (bench.py, tools/kbench.py) -------------------------
extern "C" tw_status tw_bench_stage(tw_engine* e, int kclass, int width, int height, int level, int npairs,
                                    int iters, int flags, float* avg_us)
{
    if (!e || !avg_us || npairs < 1 || iters < 1 || kclass < 0 || kclass >= TW_K_COUNT) return TW_E_BAD_PARAMETER;
    TW_HIP(e, hipSetDevice(e->device));
    Plan* pl = nullptr;
    tw_status r = get_plan(e, width, height, &pl);
    if (r) return r;
    if (level < 0 || level > pl->levels) return TW_E_BAD_PARAMETER;
    const LevelPlan& L = pl->lv[level];
    const size_t ps = (size_t)L.ps, npx0 = (size_t)width * height;
    const size_t pps = level < pl->levels ? (size_t)pl->lv[level + 1].ps : 1;
    Tmp t;
    float* I = t.alloc<float>(ps * 2 * npairs);
    float* R = t.alloc<float>(ps * 10 * npairs);
    float* M0 = t.alloc<float>(ps * 5 * npairs);
    float* M1 = t.alloc<float>(ps * 5 * npairs);
    float* fl = t.alloc<float>(ps * 2 * npairs);
    float* pf = t.alloc<float>(pps * 2 * npairs);
    uint8_t* img = t.alloc<uint8_t>(npx0 * 2 * npairs);
    const uint8_t** tab = t.alloc<const uint8_t*>(2 * (size_t)npairs);
    const int span = 10;
    const size_t G = (size_t)tw_grid_capacity(width, height, span);
    int* cnt = t.alloc<int>(npairs);
    ScanRec* rec = t.alloc<ScanRec>(G * npairs);
    float2* grid = t.alloc<float2>(G * npairs);
    if (!grid || !I || !R || !M0 || !M1 || !fl || !pf || !img || !tab || !cnt || !rec) return TW_E_NOMEM;
    {
        // deterministic pseudo-random fill (host LCG), smooth small flow
        std::vector<float> hb(ps * 10);
        uint32_t sd = 12345u;
        auto rnd = [&]() { sd = sd * 1664525u + 1013904223u; return (float)((sd >> 8) & 0xFFFF) / 65536.f - 0.5f; };
        for (float& v : hb) v = rnd() * 20.f;
        // M: G = identity, h = small smooth field -> the solved flow is (h2, h1)/(1+1e-3), a few pixels
        std::vector<float> hm(ps * 5);
        for (size_t i = 0; i < ps; i++) {
            hm[i] = 1.f;
            hm[ps + i] = 0.f;
            hm[2 * ps + i] = 1.f;
            hm[3 * ps + i] = 2.5f * sinf((float)(i % 1931) * 0.004f);
            hm[4 * ps + i] = 2.0f * cosf((float)(i % 2477) * 0.003f);
        }
        for (int j = 0; j < npairs; j++) {
            TW_TRY(h2d_sync(e, R + ps * 10 * j, hb.data(), ps * 10 * 4));
            TW_TRY(h2d_sync(e, M0 + ps * 5 * j, hm.data(), ps * 5 * 4));
            TW_TRY(h2d_sync(e, M1 + ps * 5 * j, hm.data(), ps * 5 * 4));
            TW_TRY(h2d_sync(e, I + ps * 2 * j, hb.data() + 17, ps * 2 * 4));
        }
        std::vector<float> hf(std::max(ps, pps) * 2);
        for (size_t i = 0; i < hf.size(); i++) hf[i] = 3.f * sinf((float)(i % 977) * 0.01f) + (flags & 1 ? rnd() * 8.f : 0.f);
        std::vector<uint8_t> hi(npx0);
        for (uint8_t& v : hi) v = (uint8_t)((rnd() + 0.5f) * 255.f);
        std::vector<const uint8_t*> ht(2 * (size_t)npairs);
        for (int j = 0; j < npairs; j++) {
            TW_TRY(h2d_sync(e, fl + ps * 2 * j, hf.data(), ps * 2 * 4));
            TW_TRY(h2d_sync(e, pf + pps * 2 * j, hf.data(), pps * 2 * 4));
            for (int q = 0; q < 2; q++) {
                TW_TRY(h2d_sync(e, img + npx0 * (2 * j + q), hi.data(), npx0));
                ht[2 * j + q] = img + npx0 * (2 * j + q);
            }
        }
        TW_TRY(h2d_sync(e, (void*)tab, ht.data(), sizeof(void*) * ht.size()));
    }
    hipStream_t st = e->stream;
    hipEvent_t ea, eb;
    TW_HIP(e, hipEventCreate(&ea));
    TW_HIP(e, hipEventCreate(&eb));
    int saved[TW_K_COUNT];
    memcpy(saved, e->prof_level, sizeof(saved));
    for (int i = 0; i < TW_K_COUNT; i++) e->prof_level[i] = -2;
    auto once = [&](int i) -> tw_status {
        switch (kclass) {
            case TW_K_PYR:
                e->img_aligned4 = (npx0 % 4 == 0) ? 1 : 0;
                launch_pyr(e, st, pl, level, tab, width, I, 2 * npairs);
                break;
            case TW_K_POLYEXP: return launch_polyexp(e, st, L.w, L.h, L.ld, L.ps, I, R, 2 * npairs, level);
            case TW_K_UPDATE_MATRICES: launch_update(e, st, pl, level, R, fl, pf, M0, npairs); break;
            case TW_K_BLUR_SOLVE:
                if (flags & 12) {
                    // flags 4: one M-free iteration (tw_flow_iter) from the synthetic flow field into the M1 workspace;
                    // flags 8: the same with the upsample of the coarser level's flow fused (first iteration of a level)
                    if (!flow_iter_eligible(e, L.w, L.h)) return TW_E_UNSUPPORTED;
                    FlowUps ups;
                    const bool up = (flags & 8) && level < pl->levels;
                    if (up) {
                        const LevelPlan& Pv = pl->lv[level + 1];
                        ups = FlowUps{pf, Pv.w, Pv.h, Pv.ld, Pv.ps, L.d_uxofs, L.d_uyofs, L.d_ualpha, L.d_ubeta, L.uxmax,
                                      (float)(1. / e->p.pyrScale)};
                    }
                    launch_flow_iter(e, st, L.w, L.h, L.ld, L.ps, R, up ? nullptr : fl, L.ps, M1, L.ps, up ? &ups : nullptr,
                                     npairs, level);
                    break;
                }
                launch_blur(e, st, L.w, L.h, L.ld, L.ps, (i & 1) ? M1 : M0, (i & 1) ? M0 : M1, fl, R, (flags & 2) ? 0 : 1,
                            level, npairs);
                break;
            case TW_K_SCAN: {
                GatherArgs g;
                g.flow = fl; g.fzs = 2 * L.ps; g.fps = L.ps; g.ld = L.ld; g.span = span;
                g.gw = (L.w + span - 1) / span; g.gh = (L.h + span - 1) / span; g.g = grid;
                TW_LAUNCH(e, TW_DF_SPAN_GATHER, tw_span_gather, dim3((g.gw + 63) / 64, (g.gh + 3) / 4, npairs), dim3(256), 0, st, g);
                ScanArgs a;
                a.g = grid; a.span = span; a.gw = g.gw; a.gh = g.gh; a.thr2 = 4.0; a.count = cnt;
                a.rec = rec; a.rec_zs = (long long)a.gw * a.gh;
                a.flow = nullptr; a.fps = 0; a.ld = 0;
                TW_LAUNCH(e, TW_DF_SPAN_SCAN, tw_span_scan<false>, dim3(npairs), dim3(1024), 0, st, a);
            } break;
        }
        return TW_OK;
    };
    for (int i = 0; i < 2; i++)
        if ((r = once(i))) return r;
    TW_HIP(e, hipEventRecord(ea, st));
    for (int i = 0; i < iters; i++)
        if ((r = once(i))) return r;
    TW_HIP(e, hipEventRecord(eb, st));
    TW_HIP(e, hipGetLastError());
    TW_HIP(e, hipEventSynchronize(eb));
    float ms = 0;
    TW_HIP(e, hipEventElapsedTime(&ms, ea, eb));
    *avg_us = ms * 1e3f / iters;
    memcpy(e->prof_level, saved, sizeof(saved));
    (void)hipEventDestroy(ea);
    (void)hipEventDestroy(eb);
    return TW_OK;
}

extern "C" {

tw_status tw_stage_pyr_level(tw_engine* e, const uint8_t* img, int w0, int h0, int level, float* I, int* w, int* h)
{
    if (!e || !img || !I) return TW_E_BAD_PARAMETER;
    TW_HIP(e, hipSetDevice(e->device));
    Plan* pl = nullptr;
    tw_status r = get_plan(e, w0, h0, &pl);
    if (r) return r;
    if (level < 0 || level > pl->levels) return TW_E_BAD_PARAMETER;
    const LevelPlan& L = pl->lv[level];
    Tmp t;
    uint8_t* d_img = t.alloc<uint8_t>((size_t)w0 * h0);
    float* d_I = t.alloc<float>((size_t)L.ps);
    const uint8_t** d_tab = t.alloc<const uint8_t*>(1);
    if (!d_img || !d_I || !d_tab) return TW_E_NOMEM;
    TW_TRY(h2d_sync(e, d_img, img, (size_t)w0 * h0));
    const uint8_t* hp = d_img;
    TW_TRY(h2d_sync(e, (void*)d_tab, &hp, sizeof(hp)));
    hipStream_t st = e->stream;
    e->img_aligned4 = 1;  // hipMalloc'ed image, dense rows: the kernel itself checks stride % 4
    launch_pyr(e, st, pl, level, d_tab, w0, d_I, 1);
    TW_HIP(e, hipGetLastError());
    TW_HIP(e, hipStreamSynchronize(st));
    if ((r = down_planes(e, I, d_I, L.ld, L.ps, L.w, L.h, 1))) return r;
    if (w) *w = L.w;
    if (h) *h = L.h;
    return TW_OK;
}

tw_status tw_stage_pyr_fused23(tw_engine* e, const uint8_t* img, int w0, int h0, float* I3, float* I2)
{
    if (!e || !img || !I3 || !I2) return TW_E_BAD_PARAMETER;
    TW_HIP(e, hipSetDevice(e->device));
    Plan* pl = nullptr;
    tw_status r = get_plan(e, w0, h0, &pl);
    if (r) return r;
    if (!pl->fused23) {
        e->err = "tw_stage_pyr_fused23: levels 2 and 3 of this size are not exact reductions by 4 and 8";
        return TW_E_UNSUPPORTED;
    }
    const LevelPlan &L3 = pl->lv[3], &L2 = pl->lv[2];
    Tmp t;
    uint8_t* d_img = t.alloc<uint8_t>((size_t)w0 * h0);
    float* d_I3 = t.alloc<float>((size_t)L3.ps);
    float* d_I2 = t.alloc<float>((size_t)L2.ps);
    const uint8_t** d_tab = t.alloc<const uint8_t*>(1);
    if (!d_img || !d_I3 || !d_I2 || !d_tab) return TW_E_NOMEM;
    TW_TRY(h2d_sync(e, d_img, img, (size_t)w0 * h0));
    const uint8_t* hp = d_img;
    TW_TRY(h2d_sync(e, (void*)d_tab, &hp, sizeof(hp)));
    hipStream_t st = e->stream;
    e->img_aligned4 = 1;  // hipMalloc'ed image, dense rows: the kernel itself checks stride % 4
    launch_pyr23(e, st, pl, d_tab, w0, d_I3, d_I2, 1);
    TW_HIP(e, hipGetLastError());
    TW_HIP(e, hipStreamSynchronize(st));
    if ((r = down_planes(e, I3, d_I3, L3.ld, L3.ps, L3.w, L3.h, 1))) return r;
    if ((r = down_planes(e, I2, d_I2, L2.ld, L2.ps, L2.w, L2.h, 1))) return r;
    return TW_OK;
}

tw_status tw_stage_pyr_fused01(tw_engine* e, const uint8_t* img, int w0, int h0, float* I0, float* I1)
{
    if (!e || !img || !I0 || !I1) return TW_E_BAD_PARAMETER;
    TW_HIP(e, hipSetDevice(e->device));
    Plan* pl = nullptr;
    tw_status r = get_plan(e, w0, h0, &pl);
    if (r) return r;
    if (!pyr01_fusable(pl)) {
        e->err = "tw_stage_pyr_fused01: level 1 of this size / these parameters is not an exact halving with 3-tap smoothing";
        return TW_E_UNSUPPORTED;
    }
    const LevelPlan &L0 = pl->lv[0], &L1 = pl->lv[1];
    Tmp t;
    uint8_t* d_img = t.alloc<uint8_t>((size_t)w0 * h0);
    float* d_I0 = t.alloc<float>((size_t)L0.ps);
    float* d_I1 = t.alloc<float>((size_t)L1.ps);
    const uint8_t** d_tab = t.alloc<const uint8_t*>(1);
    if (!d_img || !d_I0 || !d_I1 || !d_tab) return TW_E_NOMEM;
    TW_TRY(h2d_sync(e, d_img, img, (size_t)w0 * h0));
    const uint8_t* hp = d_img;
    TW_TRY(h2d_sync(e, (void*)d_tab, &hp, sizeof(hp)));
    hipStream_t st = e->stream;
    e->img_aligned4 = 1;  // hipMalloc'ed image, dense rows: the kernel itself checks stride % 4
    launch_pyr01(e, st, pl, d_tab, w0, d_I1, d_I0, 1);
    TW_HIP(e, hipGetLastError());
    TW_HIP(e, hipStreamSynchronize(st));
    if ((r = down_planes(e, I0, d_I0, L0.ld, L0.ps, L0.w, L0.h, 1))) return r;
    if ((r = down_planes(e, I1, d_I1, L1.ld, L1.ps, L1.w, L1.h, 1))) return r;
    return TW_OK;
}

tw_status tw_stage_png_unfilter(tw_engine* e, const uint8_t* rows, int channels, int w, int h, int waves, uint8_t* gray)
{
    if (!e || !rows || !gray || channels < 1 || channels > 4) return TW_E_BAD_PARAMETER;
    tw_status r = check_dims(e, w, h);
    if (r) return r;
    TW_HIP(e, hipSetDevice(e->device));
    const size_t nb = png_rows_bytes(w, h, channels);
    for (int y = 0; y < h; y++)
        if (rows[(size_t)y * ((size_t)w * channels + 1)] > 4) {
            e->err = "bad PNG filter type";
            return TW_E_BAD_IMAGE_FORMAT;
        }
    if (waves == 0) waves = w <= PNG_LDS_PIXELS / 16 ? 16 : (w <= PNG_LDS_PIXELS / 4 ? 4 : 1);
    if ((waves != 1 && waves != 4 && waves != 16) || w > PNG_LDS_PIXELS / waves) return TW_E_BAD_PARAMETER;
    Tmp t;
    uint8_t* d_rows_raw = t.alloc<uint8_t>(nb + 512);  // the kernel may read a few bytes before / after the rows
    uint8_t* d_rows = d_rows_raw ? d_rows_raw + 256 : nullptr;
    uint8_t* d_gray = t.alloc<uint8_t>(staged_image_bytes(w, h));
    PngJob* d_job = t.alloc<PngJob>(1);
    if (!d_rows || !d_gray || !d_job) return TW_E_NOMEM;
    TW_TRY(h2d_sync(e, d_rows, rows, nb));
    PngJob pj;
    pj.src = d_rows;
    pj.dst = d_gray;
    pj.ch = channels;
    pj.pad = 0;
    TW_TRY(h2d_sync(e, d_job, &pj, sizeof(pj)));
    PngArgs pa;
    pa.jobs = d_job;
    pa.w = w;
    pa.h = h;
    hipStream_t st = e->stream;
    if (waves == 16) TW_LAUNCH(e, TW_DF_PNG_UNFILTER, tw_png_unfilter<16>, dim3(1), dim3(1024), 0, st, pa);
    else if (waves == 4) TW_LAUNCH(e, TW_DF_PNG_UNFILTER, tw_png_unfilter<4>, dim3(1), dim3(256), 0, st, pa);
    else TW_LAUNCH(e, TW_DF_PNG_UNFILTER, tw_png_unfilter<1>, dim3(1), dim3(64), 0, st, pa);
    TW_HIP(e, hipGetLastError());
    TW_HIP(e, hipStreamSynchronize(st));
    TW_TRY(d2h_sync(e, gray, d_gray, (size_t)w * h));
    return TW_OK;
}

tw_status tw_stage_polyexp(tw_engine* e, const float* I, int w, int h, float* R5)
{
    if (!e || !I || !R5 || w < 1 || h < 1) return TW_E_BAD_PARAMETER;
    TW_HIP(e, hipSetDevice(e->device));
    const int ld = round_up(w, 32);
    const long long ps = (long long)ld * h;
    Tmp t;
    float* d_I = t.alloc<float>((size_t)ps);
    float* d_R = t.alloc<float>((size_t)ps * 5);
    if (!d_I || !d_R) return TW_E_NOMEM;
    tw_status r;
    if ((r = up_planes(e, d_I, ld, ps, I, w, h, 1))) return r;
    hipStream_t st = e->stream;
    if ((r = launch_polyexp(e, st, w, h, ld, ps, d_I, d_R, 1, -1))) return r;
    TW_HIP(e, hipGetLastError());
    TW_HIP(e, hipStreamSynchronize(st));
    return down_planes(e, R5, d_R, ld, ps, w, h, 5);
}

tw_status tw_stage_update_matrices(tw_engine* e, const float* R0_5, const float* R1_5, const float* flow2, int w,
                                   int h, float* M5)
{
    if (!e || !R0_5 || !R1_5 || !flow2 || !M5 || w < 1 || h < 1) return TW_E_BAD_PARAMETER;
    TW_HIP(e, hipSetDevice(e->device));
    const int ld = round_up(w, 32);
    const long long ps = (long long)ld * h;
    Tmp t;
    float *d_R = t.alloc<float>(ps * 10), *d_M = t.alloc<float>(ps * 5), *d_f = t.alloc<float>(ps * 2);
    if (!d_R || !d_M || !d_f) return TW_E_NOMEM;
    tw_status r;
    if ((r = up_planes(e, d_R, ld, ps, R0_5, w, h, 5)) || (r = up_planes(e, d_R + 5 * ps, ld, ps, R1_5, w, h, 5)) ||
        (r = up_planes(e, d_f, ld, ps, flow2, w, h, 2)))
        return r;
    UpdArgs a;
    memset(&a, 0, sizeof(a));
    a.R = d_R;
    a.flow = d_f;
    a.M = d_M;
    a.w = w;
    a.h = h;
    a.ld = ld;
    a.ps = ps;
    a.fps = ps;
    hipStream_t st = e->stream;
    launch_upd_kernel<false>(e, st, w, h, 1, a);
    TW_HIP(e, hipGetLastError());
    TW_HIP(e, hipStreamSynchronize(st));
    return down_planes(e, M5, d_M, ld, ps, w, h, 5);
}

tw_status tw_stage_flow_upsample_update(tw_engine* e, const float* R0_5, const float* R1_5, const float* prevflow2,
                                        int pw, int ph, int w, int h, float* flow2, float* M5)
{
    if (!e || !R0_5 || !R1_5 || !prevflow2 || !flow2 || !M5 || w < 1 || h < 1 || pw < 1 || ph < 1)
        return TW_E_BAD_PARAMETER;
    TW_HIP(e, hipSetDevice(e->device));
    const int ld = round_up(w, 32), pld = round_up(pw, 32);
    const long long ps = (long long)ld * h, pps = (long long)pld * ph;
    ResizeTab u;
    make_resize_tab(pw, ph, w, h, u);
    if (u.mode == 2) return TW_E_UNSUPPORTED;
    Tmp t;
    float *d_R = t.alloc<float>(ps * 10), *d_M = t.alloc<float>(ps * 5), *d_f = t.alloc<float>(ps * 2),
          *d_p = t.alloc<float>(pps * 2);
    int *d_xo = t.alloc<int>(w), *d_yo = t.alloc<int>(h);
    float *d_al = t.alloc<float>(2 * (size_t)w), *d_be = t.alloc<float>(2 * (size_t)h);
    if (!d_R || !d_M || !d_f || !d_p || !d_xo || !d_yo || !d_al || !d_be) return TW_E_NOMEM;
    tw_status r;
    if ((r = up_planes(e, d_R, ld, ps, R0_5, w, h, 5)) || (r = up_planes(e, d_R + 5 * ps, ld, ps, R1_5, w, h, 5)) ||
        (r = up_planes(e, d_p, pld, pps, prevflow2, pw, ph, 2)))
        return r;
    TW_TRY(h2d_sync(e, d_xo, u.xofs.data(), (size_t)w * 4));
    TW_TRY(h2d_sync(e, d_yo, u.yofs.data(), (size_t)h * 4));
    TW_TRY(h2d_sync(e, d_al, u.alpha.data(), (size_t)w * 8));
    TW_TRY(h2d_sync(e, d_be, u.beta.data(), (size_t)h * 8));
    UpdArgs a;
    memset(&a, 0, sizeof(a));
    a.R = d_R;
    a.flow = d_f;
    a.M = d_M;
    a.w = w;
    a.h = h;
    a.ld = ld;
    a.ps = ps;
    a.fps = ps;
    a.prev = d_p;
    a.pw = pw;
    a.ph = ph;
    a.pld = pld;
    a.pfps = pps;
    a.xofs = d_xo;
    a.alpha = d_al;
    a.yofs = d_yo;
    a.beta = d_be;
    a.xmax = u.xmax;
    a.scale = (float)(1. / e->p.pyrScale);
    a.store_flow = 1;
    hipStream_t st = e->stream;
    launch_upd_kernel<true>(e, st, w, h, 1, a);
    TW_HIP(e, hipGetLastError());
    TW_HIP(e, hipStreamSynchronize(st));
    if ((r = down_planes(e, flow2, d_f, ld, ps, w, h, 2))) return r;
    return down_planes(e, M5, d_M, ld, ps, w, h, 5);
}

tw_status tw_stage_blur_solve(tw_engine* e, const float* R0_5, const float* R1_5, const float* M5, int w, int h,
                              int update_matrices, float* flow2, float* Mout5)
{
    if (!e || !R0_5 || !R1_5 || !M5 || !flow2 || w < 1 || h < 1) return TW_E_BAD_PARAMETER;
    if (update_matrices && !Mout5) return TW_E_BAD_PARAMETER;
    TW_HIP(e, hipSetDevice(e->device));
    const int ld = round_up(w, 32);
    const long long ps = (long long)ld * h;
    Tmp t;
    float *d_R = t.alloc<float>(ps * 10), *d_M = t.alloc<float>(ps * 5), *d_Mo = t.alloc<float>(ps * 5),
          *d_f = t.alloc<float>(ps * 2);
    if (!d_R || !d_M || !d_Mo || !d_f) return TW_E_NOMEM;
    tw_status r;
    if ((r = up_planes(e, d_R, ld, ps, R0_5, w, h, 5)) || (r = up_planes(e, d_R + 5 * ps, ld, ps, R1_5, w, h, 5)) ||
        (r = up_planes(e, d_M, ld, ps, M5, w, h, 5)))
        return r;
    hipStream_t st = e->stream;
    launch_blur(e, st, w, h, ld, ps, d_M, d_Mo, d_f, d_R, update_matrices ? 1 : 0, -1, 1);
    TW_HIP(e, hipGetLastError());
    TW_HIP(e, hipStreamSynchronize(st));
    if ((r = down_planes(e, flow2, d_f, ld, ps, w, h, 2))) return r;
    if (update_matrices) return down_planes(e, Mout5, d_Mo, ld, ps, w, h, 5);
    return TW_OK;
}

tw_status tw_stage_flow_iter(tw_engine* e, const float* R0_5, const float* R1_5, const float* flow_in2, const float* prev2,
                             int pw, int ph, int w, int h, float* flow_out2)
{
    if (!e || !R0_5 || !R1_5 || !flow_out2 || w < 1 || h < 1 || (flow_in2 && prev2)) return TW_E_BAD_PARAMETER;
    if (!flow_iter_eligible(e, w, h)) {
        e->err = "tw_stage_flow_iter: winSize 30/31 Gaussian window and a level of at least TW_MFREE_MIN_W (320) x 20 pixels";
        return TW_E_UNSUPPORTED;
    }
    TW_HIP(e, hipSetDevice(e->device));
    const int ld = round_up(w, 32);
    const long long ps = (long long)ld * h;
    const int pld = prev2 ? round_up(pw, 32) : 0;
    const long long pps = (long long)pld * ph;
    Tmp t;
    float *d_R = t.alloc<float>(ps * 10), *d_fi = t.alloc<float>(ps * 2), *d_fo = t.alloc<float>(ps * 2);
    if (!d_R || !d_fi || !d_fo) return TW_E_NOMEM;
    tw_status r;
    if ((r = up_planes(e, d_R, ld, ps, R0_5, w, h, 5)) || (r = up_planes(e, d_R + 5 * ps, ld, ps, R1_5, w, h, 5))) return r;
    if (flow_in2 && (r = up_planes(e, d_fi, ld, ps, flow_in2, w, h, 2))) return r;
    FlowUps ups;
    if (prev2) {
        if (pw < 1 || ph < 1) return TW_E_BAD_PARAMETER;
        ResizeTab u;
        make_resize_tab(pw, ph, w, h, u);
        if (u.mode == 2) return TW_E_UNSUPPORTED;
        float* d_p = t.alloc<float>(pps * 2);
        int *d_xo = t.alloc<int>(w), *d_yo = t.alloc<int>(h);
        float *d_al = t.alloc<float>(2 * (size_t)w), *d_be = t.alloc<float>(2 * (size_t)h);
        if (!d_p || !d_xo || !d_yo || !d_al || !d_be) return TW_E_NOMEM;
        if ((r = up_planes(e, d_p, pld, pps, prev2, pw, ph, 2))) return r;
        TW_TRY(h2d_sync(e, d_xo, u.xofs.data(), (size_t)w * 4));
        TW_TRY(h2d_sync(e, d_yo, u.yofs.data(), (size_t)h * 4));
        TW_TRY(h2d_sync(e, d_al, u.alpha.data(), (size_t)w * 8));
        TW_TRY(h2d_sync(e, d_be, u.beta.data(), (size_t)h * 8));
        ups = FlowUps{d_p, pw, ph, pld, pps, d_xo, d_yo, d_al, d_be, u.xmax, (float)(1. / e->p.pyrScale)};
    }
    hipStream_t st = e->stream;
    launch_flow_iter(e, st, w, h, ld, ps, d_R, flow_in2 ? d_fi : nullptr, ps, d_fo, ps, prev2 ? &ups : nullptr, 1, -1);
    TW_HIP(e, hipGetLastError());
    TW_HIP(e, hipStreamSynchronize(st));
    return down_planes(e, flow_out2, d_fo, ld, ps, w, h, 2);
}

}  // extern "C"
