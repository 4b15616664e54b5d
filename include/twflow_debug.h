/*
 * twflow_debug.h — DIAGNOSTIC entry points of libtwflow.so.  NOT part of the drop-in boundary (include/twflow.h): no
 * interface of the reference corresponds to them, a consumer of the library never needs them, and they may change
 * without an ABI version bump.  They exist for this repository's own tests and tools and are declared here so that
 * every symbol the product library exports is declared in include/ (hygiene).
 */
#ifndef TWFLOW_DEBUG_H
#define TWFLOW_DEBUG_H

#include "twflow.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Number of captured single-pair schedules (hipGraphs, TW_LAT_GRAPH=1) this engine holds; -1 for a null engine.
 * tests/test_gpu_parity.py uses it to prove that the graph path is what really ran. */
int tw_debug_graphs(tw_engine* e);

/* "kernel name: workgroups per CU" lines of the main kernels as the runtime admits them (tools/kbench.py).
 * Returns the number of bytes written to buf (at most cap, NUL-terminated). */
int tw_debug_occupancy(char* buf, int cap);

/* TW_DEBUG_STAMPS=1 runs only: the s_memtime phase stamps of the last tw_pyr_taps launch, 64 workgroups x 4 stamps
 * into out[256]; returns 256, or 0 when no stamps were taken.  No output value of the library depends on them. */
int tw_debug_stamps(tw_engine* e, unsigned long long* out);

/* The first n (<= 4096) entries of the same buffer: tw_flow_iter's phase stamps in the variants library
 * ([workgroup 0..31][wave 0 / 9][step 40..47][stamp 0..7]; tools/fi_stamps.py). */
int tw_debug_stamps_ex(tw_engine* e, unsigned long long* out, int n);

/* The yardstick of bench.py's `frac_of_measured_copy`: `reps` launches of a float4 device-to-device copy kernel over
 * `bytes` (16 B per lane per iteration, grid-stride; read + write counted) on the engine's stream, in GB/s. */
tw_status tw_debug_copy_rate(tw_engine* e, size_t bytes, int reps, double* gbps);

/* Kernel families of tw_debug_launch_counts: every kernel launch of the library is counted under exactly one of them
 * (the copy-rate yardstick excepted).  A test that claims to exercise a kernel asserts its family's count. */
enum tw_debug_family {
    TW_DF_PYR_K3 = 0,      /* tw_pyr_k3<0 / 2>: one 3-tap pyramid level */
    TW_DF_PYR_K3F,         /* tw_pyr_k3f: levels 0 + 1 from one read */
    TW_DF_PYR_23,          /* tw_pyr_23: levels 2 + 3 from one read */
    TW_DF_PYR_TAPS,        /* tw_pyr_taps<KS>: 7 .. 63-tap levels */
    TW_DF_PYR_LEVEL,       /* tw_pyr_level / tw_pyr_level_lds: the generic levels */
    TW_DF_POLYEXP,         /* tw_polyexp_pk / tw_polyexp / tw_polyexp_band */
    TW_DF_UPDATE_MATRICES, /* tw_update_matrices<UPSAMPLE, NY> */
    TW_DF_FLOW_ITER,       /* tw_flow_iter<15, 0>: a whole iteration, input flow from memory */
    TW_DF_FLOW_ITER_UPS,   /* tw_flow_iter<15, 1>: first iteration of a level, the coarser flow upsampled in place */
    TW_DF_FLOW_ITER_ZERO,  /* tw_flow_iter<15, 2>: first iteration of the coarsest level (zero flow) */
    TW_DF_BLUR_SOLVE4,     /* tw_blur_solve4<...>: 31-tap (and variants' 51-tap) window tiles */
    TW_DF_BLUR_SOLVE4Y,    /* tw_blur_solve4y: two sub-tiles per workgroup (51-tap default on wide levels) */
    TW_DF_BLUR_SOLVE8,     /* tw_blur_solve8: packed-f32 window kernel (51-tap narrow levels) */
    TW_DF_BLUR_PP,         /* tw_blur_solve_pp: plane-parallel small-grid kernel */
    TW_DF_BLUR_GENERIC,    /* tw_blur_solve_generic: any other window size */
    TW_DF_BLUR_VARIANT,    /* variants library only: tw_blur_solve4q / 4p / 6 */
    TW_DF_BLUR_GRID,       /* tw_blur_grid: scan-fused final iteration */
    TW_DF_BOX,             /* tw_box_vscan / tw_box_hscan_solve */
    TW_DF_TWIN,            /* tw_twin_*: two bodies in one launch (single-pair schedule; tw_twin_s4_poly since round 6) */
    TW_DF_SPAN_GATHER,
    TW_DF_SPAN_SCAN,
    TW_DF_PNG_UNFILTER,
    TW_DF_SPAN_SCAN_SEG,   /* tw_span_scan_seg: a single pair's ordered scan in 16 independent segments (round 6) */
    TW_DF_BLUR_SOLVE4Q,    /* tw_blur_solve4q: solve + refresh by the horizontal item's owner — a single pair's level-0 launches (round 6) */
    TW_DF_COUNT
};

/* Launches per kernel family on this engine since it was created or the counts were last reset: counts[0 .. min(n,
 * TW_DF_COUNT) - 1]; last_z (may be NULL) receives the grid z (pairs or images) of each family's latest launch.
 * reset != 0 zeroes the counters after copying them.  Returns TW_DF_COUNT, -1 for a null engine. */
int tw_debug_launch_counts(tw_engine* e, unsigned long long* counts, unsigned long long* last_z, int n, int reset);

/* Name of a family ("tw_flow_iter", ...), NULL past the end. */
const char* tw_debug_family_name(int family);

/* The library's own byte accounting of one engine (what it has allocated and still holds), for the leak test:
 *   out[0] device bytes (workspace, flow planes, batch contexts, tables, plans)      out[1] page-locked host bytes it allocated
 *   out[2] cached plans                                                              out[3] bounce-buffer capacity (in out[1])
 *   out[4] entries of the process-wide page-lock table (tw_host_alloc / tw_host_register)   out[5] their bytes
 *   out[6] pooled profiling events                                                   out[7] captured single-pair graphs
 * Returns the number of values written (at most n), -1 for a null engine. */
int tw_debug_memory(tw_engine* e, unsigned long long* out, int n);

#ifdef __cplusplus
}
#endif
#endif
