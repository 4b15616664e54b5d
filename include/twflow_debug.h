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

#ifdef __cplusplus
}
#endif
#endif
