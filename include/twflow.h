/*
 * twflow.h — C ABI of libtwflow.so, the MI355X (gfx950) Farneback image-diff engine.
 *
 * This is the drop-in boundary for tidal-wave's hot path.  It replaces
 *   - class OpticalFlow / OpticalFlowByCPU / OpticalFlowByGPU  (/root/reference/src/opticalflow.h:38-70,
 *     src/opticalflow.cpp:78-119) — the Farneback flow of one expect/target pair, and
 *   - the USE_GPU device selection and the span-grid threshold scan inside Consumer
 *     (/root/reference/src/consumer.cpp:18-32 and :60-76).
 * Everything is plain C: pointers and sizes, no C++/HIP/torch types.  No function throws or aborts
 * ("Node-gyp cannot use exceptions", src/opticalflow.h:18): every call returns a tw_status.
 *
 * Threading: a tw_engine is NOT thread-safe; create one per worker thread (the reference creates one
 * OpticalFlow per Consumer, src/consumer.cpp:27-35).  Different engines may be used concurrently from
 * different threads.  tw_engine_create binds the HIP device on the calling thread; every later call
 * re-binds it, so workers need not call hipSetDevice themselves (fixes the reference's main-thread
 * cv::gpu::setDevice, src/consumer.cpp:22).
 *
 * There is no CPU fallback in this library: without a HIP device every compute entry point returns
 * TW_E_DEVICE.
 */
#ifndef TWFLOW_H
#define TWFLOW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 3): + tw_device_pci_bus_id, tw_host_register / tw_host_unregister, tw_has_variants, TW_OPT_POLYEXP_F32.
 * 3 (round 4): + tw_submit_png8 (PNG scanline reconstruction + gray conversion on the device).
 * 4 (rounds 5-6): + tw_stage_pyr_fused23 / tw_stage_pyr_fused01 / tw_stage_flow_iter (per-stage test entry points),
 *    tw_algorithmic_bytes_launch, tw_level_runs_flow_iter; NARROWED: tw_host_register takes whole pages only (a range that is
 *    not page-aligned or not a multiple of the page size — e.g. a malloc / cv::Mat heap block, which version 3 accepted —
 *    now answers TW_E_BAD_PARAMETER; see the declaration).
 * Additive except for that one narrowing: a consumer built against version 1 that never called tw_host_register on heap
 * blocks runs unchanged; tw_abi_version() tells the two apart. */
#define TWFLOW_ABI_VERSION 4

/* Status codes.  The first four are enum ErrorCode of /root/reference/src/opticalflow.h:9-14. */
typedef enum tw_status {
    TW_OK = 0,
    TW_E_BAD_PARAMETER = 1,    /* BadParameter   */
    TW_E_BAD_IMAGE_FORMAT = 2, /* BadImageFormat */
    TW_E_DONT_MATCH_SIZE = 3,  /* DontMatchSize  */
    TW_E_DEVICE = 4,           /* HIP failure / no device; text via tw_last_error */
    TW_E_NOMEM = 5,
    TW_E_UNSUPPORTED = 6, /* e.g. pyrScale >= 1 (OpenCV asserts), polyN > 7 */
    TW_E_BUSY = 7         /* all in-flight slots taken: tw_wait a ticket first */
} tw_status;

/* struct OpticalFlowParameter, /root/reference/src/opticalflow.h:28-36 (same fields, same order). */
typedef struct tw_params {
    double pyrScale;
    int pyrLevels;
    int winSize;
    int pyrIterations;
    int polyN;
    double polySigma;
    int flags; /* 256 = OPTFLOW_FARNEBACK_GAUSSIAN (default, src/broker.cpp:117); 0 = box window */
} tw_params;

/* struct Vector, /root/reference/src/message_queue.h:20-25. */
typedef struct tw_vector {
    int x;
    int y;
    double dx;
    double dy;
} tw_vector;

typedef struct tw_engine tw_engine;
typedef int64_t tw_ticket;

/* Defaults of Broker::createInstance, /root/reference/src/broker.cpp:111-117. */
void tw_default_params(tw_params* p);

/* TWFLOW_ABI_VERSION of the library that was loaded (a consumer compares it with the header it was built against). */
int tw_abi_version(void);

/* 1 for a `make VARIANTS=1` build (tidal-wave_amd/libtwflow_variants.so: the default library plus the measured-slower
 * A/B kernels behind TW_BLUR_VARIANT / TW_POLY_VARIANT / TW_BLUR_SMALL / TW_UPD_NY — tests and tools only), 0 for the
 * product library, which carries only kernels a launch can reach with no environment variable set. */
int tw_has_variants(void);

/* cv::gpu::getCudaEnabledDeviceCount() of src/consumer.cpp:19-20: number of usable HIP devices
 * (0 when there is none or the runtime cannot initialise). */
int tw_device_count(void);

/* PCI bus id of a device ("0000:c1:00.0", NUL-terminated, cap >= 16) — lets the host layer place consumer i and its
 * page-locked buffers on the NUMA node of GPU i (SURVEY.md 8(e) "scaling risks"; the reference's consumer i <-> device i
 * mapping is src/consumer.cpp:18-24).  TW_E_DEVICE if the device does not exist. */
tw_status tw_device_pci_bus_id(int device, char* buf, int cap);

/* new OpticalFlowByGPU() + cv::gpu::setDevice(id), src/consumer.cpp:21-30.
 * `slots` = image pairs per batch (>= 1); up to three batches may be outstanding.  Parameters are fixed per engine like Consumer::parameter (src/consumer.cpp:97). */
tw_status tw_engine_create(int device, const tw_params* params, int slots, tw_engine** out);
void tw_engine_destroy(tw_engine* e);

/* Human-readable text for a status; the four reference codes have no fixed text of their own. */
const char* tw_strerror(tw_status s);
/* Message of the last failure on this engine (e.g. hipGetErrorString), "" if none. */
const char* tw_last_error(const tw_engine* e);

/* OpticalFlow::calculateInternal, /root/reference/src/opticalflow.h:49 / src/opticalflow.cpp:97-119:
 * dense flow of one 8-bit gray pair (row stride in bytes), planar flowx/flowy out (w*h floats each,
 * either may be NULL).  `seconds` = device compute time, the meaning of OpticalFlowStatus::time
 * (src/opticalflow.cpp:112-118).  Both images must already have equal size (the <=5 px reconcile of
 * src/opticalflow.cpp:52-68 is host-layer work). */
tw_status tw_flow_u8(tw_engine* e, const uint8_t* expect, const uint8_t* target, int width, int height,
                     ptrdiff_t stride, float* flowx, float* flowy, float* seconds);

/* Flow + the span-grid scan of src/consumer.cpp:60-76 in one call; only the flagged grid vectors are
 * copied back.  `out` has room for `cap` vectors; *n receives the number found (may exceed cap, then
 * only cap are written).  Vectors are in the reference's row-major order (y, then x). */
tw_status tw_diff_u8(tw_engine* e, const uint8_t* expect, const uint8_t* target, int width, int height,
                     ptrdiff_t stride, int span, double threshold, tw_vector* out, int cap, int* n,
                     float* seconds);

/* Asynchronous pair of the same operation, for batching (the Manager queue keeps `slots` jobs in
 * flight per GPU).  The host images are copied to pinned staging before tw_submit_u8 returns. */
tw_status tw_submit_u8(tw_engine* e, const uint8_t* expect, const uint8_t* target, int width, int height,
                       ptrdiff_t stride, int span, double threshold, tw_ticket* ticket);
/* The same pair handed over HALF-DECODED (ABI 3): what cv::imread(path, GRAYSCALE) of src/opticalflow.cpp:37,44 does to
 * an 8-bit non-interlaced PNG after the inflate — scanline reconstruction (filter types 0-4 of ISO/IEC 15948 §9) and
 * libpng 1.5's RGB -> gray — runs on the device, so the host's decode threads only inflate.  For each
 * image: channels 1 (gray), 2 (gray + alpha), 3 (RGB) or 4 (RGBA) = the inflated IDAT stream, `height` rows of
 * 1 + width * channels bytes, filter type byte first; channels 0 = a plain gray image, dense rows of `width` bytes (so a
 * pair may mix a PNG with an image the host decoded).  The result equals tw_submit_u8 on the host-decoded images bit for
 * bit.  A filter type above 4 answers TW_E_BAD_IMAGE_FORMAT.  Page-locked buffers are DMA-ed in place like
 * tw_submit_u8's (keep them until tw_wait); pageable ones are copied before the call returns. */
tw_status tw_submit_png8(tw_engine* e, const uint8_t* expect, int expect_channels, const uint8_t* target,
                         int target_channels, int width, int height, int span, double threshold, tw_ticket* ticket);
/* Same with the images already resident in device memory (HBM) of the engine's device. */
tw_status tw_submit_dev(tw_engine* e, const void* d_expect, const void* d_target, int width, int height,
                        ptrdiff_t stride, int span, double threshold, tw_ticket* ticket);
/* Submitted pairs are gathered into a batch of up to `slots` pairs and executed together, level by level
 * (every kernel launch covers as many pairs as it takes to fill the GPU).  A batch starts executing when it
 * is full, when tw_flush is called, or when one of its tickets is waited for.  `seconds` of a pair is the
 * device time of its batch divided by the pairs in it (exact for a batch of one).
 * Cold-start ramp (round 6; engines of >= 64 slots, host images): a batch that starts while nothing of an earlier
 * batch is still executing goes out in up to three pieces — the first quarter of `slots`, the second quarter, the
 * rest — each as soon as ITS uploads have arrived, instead of behind the batch's last upload; results, tickets and
 * their order are the same (environment TW_RAMP=0 turns it off). */
tw_status tw_flush(tw_engine* e);
tw_status tw_wait(tw_engine* e, tw_ticket ticket, tw_vector* out, int cap, int* n, float* seconds);

/* Number of grid points ceil(h/span)*ceil(w/span): the capacity that can never overflow. */
int tw_grid_capacity(int width, int height, int span);

/* Raw device memory helpers so a host without a HIP binding can keep inputs resident. */
tw_status tw_dev_alloc(tw_engine* e, size_t bytes, void** dptr);
tw_status tw_dev_free(tw_engine* e, void* dptr);
tw_status tw_dev_upload(tw_engine* e, void* dptr, const void* host, size_t bytes);

/* Engine options.
 * TW_OPT_SCAN_FUSED_FINAL (default 0): tw_submit_* / tw_diff_u8 with span 10 and winSize 30/31 evaluate the last
 *   window average + solve of level 0 only at the span-grid points the scan reads.  Status and vectors are
 *   bit-identical; the dense flow field of that last iteration is simply never materialised (tw_flow_u8 always
 *   computes the whole field).  Off by default because the reference's path — and bench.py's headline — produce
 *   the full field.
 * TW_OPT_POLYEXP_F32 (default 0): MEASUREMENT variant of the polynomial expansion with float instead of double
 *   horizontal accumulators (polyN 5 or 7; what OpenCV's CUDA module does).  NOT bit-identical to the CPU reference;
 *   exists so that "what would the kernel cost without its f64 half, and what would it do to the flow" is a number
 *   (bench.py `polyexp_f32_variant`, profiles/r03_polyexp_f32.md).  Value 2 (polyN 7) additionally fuses every
 *   multiply-add (v_pk_fma_f32).  Never on by default, never bench.py's `value`. */
enum { TW_OPT_SCAN_FUSED_FINAL = 1, TW_OPT_POLYEXP_F32 = 2 };
tw_status tw_set_option(tw_engine* e, int option, int value);

/* Page-locked host memory.  Images handed to tw_submit_u8 from a block tw_host_alloc returned (or that the caller
 * page-locked through tw_host_register) are DMA-ed to the device straight from the caller's buffer on the engine's
 * copy stream — no staging copy — and must therefore stay unchanged until tw_wait() of the ticket returns.
 * Any other memory is treated as pageable: it is staged through the engine's own pinned buffers and may be reused as
 * soon as tw_submit_u8 returns (memory page-locked behind the library's back, e.g. by a direct hipHostRegister, is
 * staged too — correct, one memcpy slower).  The blocks are usable by the engines of every device and by every
 * thread (the table of them is process-wide).  Either way the upload of batch j+1 overlaps the kernels of batch j
 * (BASELINE config 3: "pinned H2D/D2H overlapped on a side stream").
 * Releasing a block: tw_host_free / tw_host_unregister wait only for the uploads of the engine they are called on.
 * The caller must have collected (tw_wait) every ticket of EVERY engine that was handed images from the block before
 * it releases it.  A block is released the way it was made: tw_host_free on a registered block, or
 * tw_host_unregister on a tw_host_alloc block, answers TW_E_BAD_PARAMETER and releases nothing. */
tw_status tw_host_alloc(tw_engine* e, size_t bytes, void** hptr);
tw_status tw_host_free(tw_engine* e, void* hptr);
/* Page-lock / release memory the caller owns (hipHostRegister / hipHostUnregister + the table above).
 * WHOLE PAGES ONLY: `hptr` must be page-aligned and `bytes` a multiple of the page size (sysconf(_SC_PAGESIZE)),
 * i.e. memory the caller owns page-wise — mmap, aligned_alloc(page, n * page), posix_memalign.  Anything else
 * (a malloc / new block, a cv::Mat's heap buffer) answers TW_E_BAD_PARAMETER with a tw_last_error text and
 * page-locks nothing: a partial page is shared with the allocator's other blocks, and page-locking / releasing it
 * next to them is what a GPU memory-access fault of round 4 traced to.  Use tw_host_alloc for
 * buffers that need not live at a given address. */
tw_status tw_host_register(tw_engine* e, void* hptr, size_t bytes);
tw_status tw_host_unregister(tw_engine* e, void* hptr);

/* ---- instrumentation (bench.py / tests) ---------------------------------------------------------- */

/* Kernel classes, in data-flow order. */
enum {
    TW_K_PYR = 0,
    TW_K_POLYEXP = 1,
    TW_K_UPDATE_MATRICES = 2,
    TW_K_BLUR_SOLVE = 3,
    TW_K_SCAN = 4,
    TW_K_COUNT = 5
};

/* Every later launch of kernel class `kclass` at pyramid level `level` (-1: every level, -2: off) is
 * bracketed by hipEvents on the stream it is launched on; several classes may be on at once;
 * kclass -1 turns all of them off.  Totals accumulate per class until read. */
tw_status tw_prof_select(tw_engine* e, int kclass, int level);
/* Sum of event-measured milliseconds and number of launches of `kclass` since the last read; resets both.
 * Synchronises the device. */
tw_status tw_prof_read(tw_engine* e, int kclass, double* ms_total, int* launches);
/* Bytes one launch of `kclass` at `level` must move for a w x h pair: every input of the kernel AS BUILT read once,
 * every output written once (blur+solve: averaged over the pyrIterations launches of a level — a launch fused with
 * the matrix refresh moves 80 B/px, the last one 28 B/px; flows that never leave the registers are not counted). */
double tw_algorithmic_bytes(const tw_engine* e, int kclass, int level, int width, int height);
/* The same for a batch of `npairs` pairs: the model follows the schedule's own choices (one shared predicate in
 * csrc/twflow.hip) — a single pair, a batch too small to fill the chip with tw_flow_iter workgroups or a level whose launches
 * do not cover the batch keeps tw_update_matrices + tw_blur_solve* (UPDATE_MATRICES 60 B/px + the coarser flow, window launches
 * 80 / 28 B/px) where a full batch runs tw_flow_iter (56 B/px, no UPDATE_MATRICES launch).  tw_algorithmic_bytes is this with
 * npairs = the engine's slots. */
double tw_algorithmic_bytes_launch(const tw_engine* e, int kclass, int level, int width, int height, int npairs);
/* 1 if `level` of a batch of `npairs` pairs of width x height runs tw_flow_iter (whole iterations, no M in HBM), 0 if it runs
 * tw_update_matrices + tw_blur_solve* launches, -1 on a bad argument.  The scan-fused option is assumed with span 10. */
int tw_level_runs_flow_iter(const tw_engine* e, int width, int height, int level, int npairs);
/* SURVEY.md 8(d) model of one whole pair (each named stage of the reference reads its inputs once and writes its
 * outputs once; 991.8 MB at 1080p with the defaults): the figure BASELINE.md prices a pair against. */
double tw_algorithmic_bytes_pair(const tw_engine* e, int width, int height, int span);
/* Sum of tw_algorithmic_bytes over every launch of a pair: what the kernels as built must move for a full batch (611.8 MB at 1080p with the defaults; 859.6 before tw_flow_iter). */
double tw_min_traffic_bytes_pair(const tw_engine* e, int width, int height, int span);
/* Number of pyramid levels (= index of the coarsest level) the engine uses for w x h. */
int tw_num_levels(const tw_engine* e, int width, int height);
/* Image pairs one kernel launch covers at `level` (the level-major batch schedule), -1 on error. */
int tw_level_chunk(tw_engine* e, int width, int height, int level);

/* Isolated timing of one kernel class at `level` of a width x height pair, `npairs` pairs per launch, on
 * synthetic device-resident data: average microseconds per launch over `iters` back-to-back launches
 * (hipEvents on the engine's stream).  flags: 1 = rough (non-smooth) flow field, 2 = tw_blur_solve without
 * the fused matrix refresh. */
tw_status tw_bench_stage(tw_engine* e, int kclass, int width, int height, int level, int npairs, int iters,
                         int flags, float* avg_us);

/* ---- per-stage entry points (parity tests; host buffers, synchronous) --------------------------------
 * Layouts: images/planes are dense row-major; R and M are 5 planes [5][h][w]; flow is 2 planes. */
tw_status tw_stage_pyr_level(tw_engine* e, const uint8_t* img, int w0, int h0, int level, float* I, int* w,
                             int* h);
/* Levels 3 and 2 of one image from the batch path's one-read kernel (tw_pyr_23; round 5): I3 (w0/8 x h0/8) and I2
 * (w0/4 x h0/4).  TW_E_UNSUPPORTED when the size has no exact reductions by 4 and 8 (the engine then runs
 * tw_stage_pyr_level's kernels for those levels). */
tw_status tw_stage_pyr_fused23(tw_engine* e, const uint8_t* img, int w0, int h0, float* I3, float* I2);
/* Levels 0 and 1 of one image from one read (tw_pyr_k3f; round 5): I0 (w0 x h0) and I1 (w0/2 x h0/2).  TW_E_UNSUPPORTED
 * when level 1 is not an exact halving with 3-tap smoothing (pyrScale 0.5, even width and height). */
tw_status tw_stage_pyr_fused01(tw_engine* e, const uint8_t* img, int w0, int h0, float* I0, float* I1);
/* PNG scanline reconstruction + gray conversion of one image (tw_submit_png8's kernel): `rows` = h rows of
 * 1 + w * channels bytes, channels 1-4; `waves` = 0 (the engine's choice for this width), 1, 4 or 16 waves per image. */
tw_status tw_stage_png_unfilter(tw_engine* e, const uint8_t* rows, int channels, int w, int h, int waves,
                                uint8_t* gray);
tw_status tw_stage_polyexp(tw_engine* e, const float* I, int w, int h, float* R5);
tw_status tw_stage_update_matrices(tw_engine* e, const float* R0_5, const float* R1_5, const float* flow2,
                                   int w, int h, float* M5);
tw_status tw_stage_flow_upsample_update(tw_engine* e, const float* R0_5, const float* R1_5,
                                        const float* prevflow2, int pw, int ph, int w, int h, float* flow2,
                                        float* M5);
tw_status tw_stage_blur_solve(tw_engine* e, const float* R0_5, const float* R1_5, const float* M5, int w, int h,
                              int update_matrices, float* flow2, float* Mout5);
/* One whole iteration of the flow update WITHOUT M in memory (tw_flow_iter, round 5): flow_out = solve(window average of
 * FarnebackUpdateMatrices(R0, R1, flow)) where flow = flow_in2, or (prev2) the coarser level's flow prev2 (pw x ph)
 * resized INTER_LINEAR to w x h and scaled by 1/pyrScale, or zero (both null).  winSize 30/31 Gaussian window, levels of
 * at least 320 x 20 pixels; TW_E_UNSUPPORTED otherwise. */
tw_status tw_stage_flow_iter(tw_engine* e, const float* R0_5, const float* R1_5, const float* flow_in2, const float* prev2,
                             int pw, int ph, int w, int h, float* flow_out2);

#ifdef __cplusplus
}
#endif
#endif /* TWFLOW_H */
