/* ORACLE -- test infrastructure only (see dbpost_oracle.c). */
#ifndef DBPOST_ORACLE_H
#define DBPOST_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { DBPO_OK = 0, DBPO_SKIP_NPTS = 1, DBPO_SKIP_SSID = 2, DBPO_SKIP_SCORE = 3, DBPO_SKIP_UNCLIP = 4, DBPO_SKIP_SSID2 = 5 };

/* one record per processed contour, in processing order (reference loop db_postprocess.cpp:254-314) */
typedef struct {
    int status, is_hole, trig_x, trig_y, start_x, start_y, npts, npix, npoly;
    float rect[5];       /* minAreaRect(contour): cx, cy, w, h, angle(deg) */
    float minibox[8];    /* GetMiniBoxes order TL,TR,BR,BL */
    float score, distance;
    float urect[5];      /* minAreaRect(unclip polygon) */
    int box[8];          /* final integer box */
} dbpost_oracle_dbg;

int dbpost_oracle_run(const float *pred, const uint8_t *bitmap, int H, int W, float box_thresh,
                      float unclip_ratio, int src_w, int src_h, int *boxes_out, int max_boxes,
                      dbpost_oracle_dbg *dbg, int dbg_cap, int *n_contours_out);
int dbpost_oracle_contours(const uint8_t *bitmap, int H, int W, int *npts, int *is_hole, int *trig_xy,
                           int cap, int *pts_xy, int pts_cap);
void dbpost_oracle_min_area_rect(const float *pts_xy, int n, float *rect5, float *minibox8, float *ssid);
int dbpost_oracle_clipper_offset(const long long *path_xy, int npts, double delta, long long *out_xy, int out_cap);
void dbpost_oracle_set_clipper_ref(void *fn);
void dbpost_oracle_set_padding_resize(int on);
void dbpost_oracle_dilate2x2(const uint8_t *in, int H, int W, uint8_t *out);
void dbpost_oracle_binarize(const float *pred, size_t n, float thresh, uint8_t *bitmap);

#ifdef __cplusplus
}
#endif
#endif
