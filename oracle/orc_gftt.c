/*
 * orc_gftt.c — CPU ORACLE (test infrastructure): Shi-Tomasi "good features to
 * track" with the reference's mask construction.
 *
 * Restates, for Frontend::DetectFeatures (src/frontend.cpp:36-70; detector
 * cv::GFTTDetector::create(num_features, 0.01, 20) at :24), the published
 * algorithm of OpenCV 4.5.4 (NOT in /root/reference):
 *   modules/imgproc/src/featureselect.cpp  goodFeaturesToTrack
 *   modules/imgproc/src/corner.cpp         cornerEigenValsVecs / calcMinEigenVal
 *   modules/imgproc/src/deriv.cpp          Sobel via sepFilter2D (8u -> 32f)
 *   modules/imgproc/src/box_filter.*       unnormalised 3x3, f64 accumulators
 * PARITY UNPINNED (see svs_oracle.h).  Declared float operation order (scalar
 * C++ path of OpenCV, no FMA):
 *   s = (float)(1/3060.)                        scale 1/(2^(3-1) * 3 * 255)
 *   Dx = (r0 + r2)*s + r1*(2s),  r_k = (float)p[y+k-1][x+1] - (float)p[y+k-1][x-1]
 *   Dy = c2 - c0,  c_k = (s*p[.][x-1] + (2s)*p[.][x]) + s*p[.][x+1]  on row y+k-1
 *   cov = (Dx*Dx, Dx*Dy, Dy*Dy);  box = (float)(f64 sum of the 3x3, three columns first, then the three rows)
 *   a = box_xx*0.5f, b = box_xy, c = box_yy*0.5f
 *   eig = (a + c) - sqrtf((a - c)*(a - c) + b*b)
 * Borders: BORDER_REFLECT_101 for Sobel (on the image) and for the box filter
 * (on the covariance maps).
 */
#include "svs_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <stddef.h>

static inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

void orc_min_eig_map(const uint8_t *img, int stride, int w, int h, float *eig)
{
    const float s1 = orc_whatif[3] ? (float)(1.0 / 12.0) : (float)(1.0 / 3060.0);
    const float s2 = orc_whatif[3] ? (float)(2.0 * (1.0 / 12.0)) : (float)(2.0 * (1.0 / 3060.0));
    size_t P = (size_t)w * h;
    float *cxx = (float *)malloc(sizeof(float) * P);
    float *cxy = (float *)malloc(sizeof(float) * P);
    float *cyy = (float *)malloc(sizeof(float) * P);
    for (int y = 0; y < h; ++y) {
        const uint8_t *r[3];
        for (int k = 0; k < 3; ++k) r[k] = img + (size_t)reflect101(y + k - 1, h) * stride;
        for (int x = 0; x < w; ++x) {
            int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
            /* Dx: row kernel [-1 0 1] (exact), column kernel [s 2s s] */
            float d0 = (float)r[0][xp] - (float)r[0][xm];
            float d1 = (float)r[1][xp] - (float)r[1][xm];
            float d2 = (float)r[2][xp] - (float)r[2][xm];
            float dx = (d0 + d2) * s1 + d1 * s2;
            /* Dy: row kernel [s 2s s] left-to-right, column kernel [-1 0 1] */
            float c0 = (s1 * (float)r[0][xm] + s2 * (float)r[0][x]) + s1 * (float)r[0][xp];
            float c2 = (s1 * (float)r[2][xm] + s2 * (float)r[2][x]) + s1 * (float)r[2][xp];
            float dy = c2 - c0;
            size_t i = (size_t)y * w + x;
            cxx[i] = dx * dx; cxy[i] = dx * dy; cyy[i] = dy * dy;
        }
    }
    for (int y = 0; y < h; ++y) {
        int ys[3] = { reflect101(y - 1, h), y, reflect101(y + 1, h) };
        for (int x = 0; x < w; ++x) {
            int xs[3] = { reflect101(x - 1, w), x, reflect101(x + 1, w) };
            /* f64 accumulators in box_filter's order: RowSum adds the three columns of a row, ColumnSum adds the rows
             * (the first window of its running sum; a later window's "+ new row - old row" is the same number whenever
             * the sum is exact, which it is but for ~1e-7 of the pixels — and there the f32 cast below hides it:
             * 0 differences in 5.6e7 sums on the synthetic frames, round 5).  Until round 5 this was one row-major sum. */
            double sxx = 0, sxy = 0, syy = 0;
            float fxx = 0.f, fxy = 0.f, fyy = 0.f;
            for (int j = 0; j < 3; ++j) {
                const size_t q0 = (size_t)ys[j] * w + xs[0], q1 = (size_t)ys[j] * w + xs[1], q2 = (size_t)ys[j] * w + xs[2];
                const double rxx = ((double)cxx[q0] + (double)cxx[q1]) + (double)cxx[q2];
                const double rxy = ((double)cxy[q0] + (double)cxy[q1]) + (double)cxy[q2];
                const double ryy = ((double)cyy[q0] + (double)cyy[q1]) + (double)cyy[q2];
                if (j == 0) { sxx = rxx; sxy = rxy; syy = ryy; }
                else { sxx += rxx; sxy += rxy; syy += ryy; }
                fxx += cxx[q0]; fxy += cxy[q0]; fyy += cyy[q0];
                fxx += cxx[q1]; fxy += cxy[q1]; fyy += cyy[q1];
                fxx += cxx[q2]; fxy += cxy[q2]; fyy += cyy[q2];
            }
            if (orc_whatif[5]) { sxx = fxx; sxy = fxy; syy = fyy; }
            float a = (float)sxx * 0.5f, b = (float)sxy, c = (float)syy * 0.5f;
            float t = a - c;
            eig[(size_t)y * w + x] = (a + c) - sqrtf(t * t + b * b);
        }
    }
    free(cxx); free(cxy); free(cyy);
}

void orc_gftt_mask(uint8_t *mask, int w, int h, const float *rect_xy, int nrect)
{
    /* src/frontend.cpp:42-47: mask = 255; cv::rectangle(mask, pt-(10,10),
     * pt+(10,10), 0, FILLED): Point2f -> Point rounds half-to-even, both
     * corners inclusive, clipped to the image. */
    memset(mask, 255, (size_t)w * h);
    for (int r = 0; r < nrect; ++r) {
        float fx = rect_xy[2 * r], fy = rect_xy[2 * r + 1];
        int x1 = (int)lrintf(fx - 10.f), y1 = (int)lrintf(fy - 10.f);
        int x2 = (int)lrintf(fx + 10.f), y2 = (int)lrintf(fy + 10.f);
        if (orc_whatif[4]) {
            x1 = (int)floorf(fx - 10.f + 0.5f); y1 = (int)floorf(fy - 10.f + 0.5f);
            x2 = (int)floorf(fx + 10.f + 0.5f); y2 = (int)floorf(fy + 10.f + 0.5f);
        }
        if (x1 < 0) x1 = 0;
        if (y1 < 0) y1 = 0;
        if (x2 > w - 1) x2 = w - 1;
        if (y2 > h - 1) y2 = h - 1;
        for (int y = y1; y <= y2; ++y)
            for (int x = x1; x <= x2; ++x) mask[(size_t)y * w + x] = 0;
    }
}

typedef struct { float v; int idx; } cand_t;

static int cand_cmp(const void *pa, const void *pb)
{
    /* greaterThanPtr: value descending, then address descending */
    const cand_t *a = (const cand_t *)pa, *b = (const cand_t *)pb;
    if (a->v > b->v) return -1;
    if (a->v < b->v) return 1;
    if (orc_whatif[6]) return a->idx < b->idx ? -1 : (a->idx > b->idx ? 1 : 0);
    return a->idx > b->idx ? -1 : (a->idx < b->idx ? 1 : 0);
}

int orc_gftt(const uint8_t *img, int stride, int w, int h, const float *rect_xy,
             int nrect, int max_corners, double quality, double min_dist,
             float *out_xy)
{
    size_t P = (size_t)w * h;
    float *eig = (float *)malloc(sizeof(float) * P);
    uint8_t *mask = (uint8_t *)malloc(P);
    orc_min_eig_map(img, stride, w, h, eig);
    orc_gftt_mask(mask, w, h, rect_xy, nrect);

    /* minMaxLoc(eig, 0, &maxVal, 0, 0, mask) */
    double maxVal = 0;
    int any = 0;
    for (size_t i = 0; i < P; ++i)
        if (mask[i]) {
            if (!any || eig[i] > maxVal) { maxVal = eig[i]; any = 1; }
        }
    if (!any) maxVal = 0;
    /* threshold(eig, eig, maxVal*quality, 0, THRESH_TOZERO): keep v > (float)thr;
     * dilate 3x3 + (val == dilated) on interior pixels == 3x3 local max of the
     * raw map for values above the threshold. */
    float thr = (float)(maxVal * quality);
    cand_t *cand = (cand_t *)malloc(sizeof(cand_t) * (P + 16));
    int nc = 0;
    for (int y = 1; y < h - 1; ++y)
        for (int x = 1; x < w - 1; ++x) {
            size_t i = (size_t)y * w + x;
            float v = eig[i];
            if (!(v > thr)) continue;          /* TOZERO */
            if (v == 0.f) continue;            /* val != 0 */
            if (!mask[i]) continue;
            int ismax = 1;
            for (int j = -1; j <= 1 && ismax; ++j)
                for (int k = -1; k <= 1; ++k) {
                    float u = eig[i + (ptrdiff_t)j * w + k];
                    float ut = u > thr ? u : 0.f;
                    if (ut > v) { ismax = 0; break; }
                }
            if (ismax) { cand[nc].v = v; cand[nc].idx = (int)i; ++nc; }
        }
    int ncorners = 0;
    if (nc > 0) {
        qsort(cand, (size_t)nc, sizeof(cand_t), cand_cmp);
        if (min_dist >= 1) {
            int cell = (int)lrint(min_dist);
            int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
            /* per-cell lists as linked chains */
            int *head = (int *)malloc(sizeof(int) * (size_t)gw * gh);
            int *nxt = (int *)malloc(sizeof(int) * (size_t)(max_corners > 0 ? max_corners : nc));
            float *acc = out_xy;
            for (int i = 0; i < gw * gh; ++i) head[i] = -1;
            float md2 = (float)(min_dist * min_dist);
            for (int i = 0; i < nc; ++i) {
                int y = cand[i].idx / w, x = cand[i].idx - y * w;
                int xc = x / cell, yc = y / cell;
                int x1 = xc - 1 < 0 ? 0 : xc - 1, y1 = yc - 1 < 0 ? 0 : yc - 1;
                int x2 = xc + 1 > gw - 1 ? gw - 1 : xc + 1, y2 = yc + 1 > gh - 1 ? gh - 1 : yc + 1;
                int good = 1;
                for (int yy = y1; yy <= y2 && good; ++yy)
                    for (int xx = x1; xx <= x2 && good; ++xx)
                        for (int j = head[yy * gw + xx]; j >= 0; j = nxt[j]) {
                            float dx = (float)x - acc[2 * j], dy = (float)y - acc[2 * j + 1];
                            if (dx * dx + dy * dy < md2) { good = 0; break; }
                        }
                if (good) {
                    acc[2 * ncorners] = (float)x; acc[2 * ncorners + 1] = (float)y;
                    nxt[ncorners] = head[yc * gw + xc];
                    head[yc * gw + xc] = ncorners;
                    ++ncorners;
                    if (max_corners > 0 && ncorners == max_corners) break;
                }
            }
            free(head); free(nxt);
        } else {
            for (int i = 0; i < nc; ++i) {
                int y = cand[i].idx / w, x = cand[i].idx - y * w;
                out_xy[2 * ncorners] = (float)x; out_xy[2 * ncorners + 1] = (float)y;
                ++ncorners;
                if (max_corners > 0 && ncorners == max_corners) break;
            }
        }
    }
    free(cand); free(eig); free(mask);
    return ncorners;
}
