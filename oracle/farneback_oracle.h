/* farneback_oracle.h — TEST INFRASTRUCTURE (see farneback_oracle.c header). */
#ifndef FARNEBACK_ORACLE_H
#define FARNEBACK_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* /root/reference/src/opticalflow.h:28-36 */
typedef struct {
    double pyrScale;
    int pyrLevels;
    int winSize;
    int pyrIterations;
    int polyN;
    double polySigma;
    int flags;
} orc_params;

/* /root/reference/src/message_queue.h:20-25 */
typedef struct {
    int x;
    int y;
    double dx;
    double dy;
} orc_vector;

typedef struct {
    int width, height, smooth_sz;
    double sigma, scale;
} orc_level;

int orc_level_plan(int w0, int h0, double pyr_scale, int levels, orc_level* out /* [levels+1] */);
void orc_gaussian_kernel(int n, double sigma, float* k);
void orc_pyr_level(const uint8_t* img, int w0, int h0, const orc_level* lv, float* I);
void orc_polyexp_setup(int n, double sigma, float* g_c, float* xg_c, float* xxg_c, double ig[4]);
void orc_polyexp(const float* src, int width, int height, int n, double sigma, float* dst_c5);
void orc_update_matrices(const float* R0, const float* R1, const float* flow_c2, float* M_c5, int width, int height,
                         int y0, int y1);
void orc_window_kernel(int block_size, float* kernel /* [block_size/2+1] */);
void orc_update_flow_gaussian(const float* R0, const float* R1, float* flow_c2, float* M_c5, int width, int height,
                              int block_size, int update_matrices);
void orc_update_flow_box(const float* R0, const float* R1, float* flow_c2, float* M_c5, int width, int height,
                         int block_size, int update_matrices);
void orc_flow_upsample(const float* prev_c2, int pw, int ph, float* flow_c2, int w, int h, double pyr_scale);
int orc_farneback(const uint8_t* prev, const uint8_t* next, int w, int h, const orc_params* p, float* flowx,
                  float* flowy);
/* /root/reference/src/opticalflow.cpp:52-68 (the <= 5 px size reconcile: 8-bit INTER_LINEAR resize of the target) */
void orc_resize_u8_linear(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh);
int orc_reconcile_target(const uint8_t* target, int tw, int th, int ew, int eh, uint8_t* target_out);
int orc_span_scan(const float* flowx, const float* flowy, int w, int h, int span, double threshold, orc_vector* out,
                  int cap);

#ifdef __cplusplus
}
#endif
#endif
