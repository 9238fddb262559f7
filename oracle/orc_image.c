/*
 * orc_image.c — CPU ORACLE (test infrastructure): image pyramids, Scharr
 * derivatives and pyramidal Lucas-Kanade.
 *
 * Restates, for the call sites src/frontend.cpp:105-109 (FindFeaturesInRight)
 * and :353-357 (TrackLastFrame) of the reference, the published algorithm of
 * OpenCV 4.5.4 (README.md:29-35 pins the version; OpenCV is NOT in
 * /root/reference):
 *   modules/imgproc/src/pyramids.cpp   pyrDown, u8, 5x5 [1 4 6 4 1]^2
 *   modules/video/src/lkpyramid.cpp    buildOpticalFlowPyramid, calcSharrDeriv,
 *                                      LKTrackerInvoker::operator()
 * PARITY UNPINNED (see svs_oracle.h).  Build with -ffp-contract=off.
 */
#include "svs_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <stddef.h>

static inline int reflect101(int p, int len)
{
    /* cv::borderInterpolate(p, len, BORDER_REFLECT_101) */
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

/* ------------------------------------------------------------------ */
/* pyrDown (u8): dst = (sum_{i,j} k_i k_j src(2x+i-2, 2y+j-2) + 128) >> 8,    */
/* k = [1 4 6 4 1], REFLECT_101, dst size ((w+1)/2, (h+1)/2)                  */
void orc_pyrdown(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst,
                 int dstride)
{
    int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    int *rows = (int *)malloc(sizeof(int) * (size_t)dw * 5);
    int *tabx = (int *)malloc(sizeof(int) * (size_t)dw * 5);
    for (int x = 0; x < dw; ++x)
        for (int k = 0; k < 5; ++k)
            tabx[x * 5 + k] = reflect101(2 * x + k - 2, sw);
    for (int y = 0; y < dh; ++y) {
        for (int k = 0; k < 5; ++k) {
            int sy = reflect101(2 * y + k - 2, sh);
            const uint8_t *s = src + (size_t)sy * sstride;
            int *r = rows + k * dw;
            for (int x = 0; x < dw; ++x) {
                const int *t = tabx + x * 5;
                r[x] = s[t[0]] + 4 * s[t[1]] + 6 * s[t[2]] + 4 * s[t[3]] + s[t[4]];
            }
        }
        uint8_t *d = dst + (size_t)y * dstride;
        for (int x = 0; x < dw; ++x) {
            int v = rows[x] + 4 * rows[dw + x] + 6 * rows[2 * dw + x] +
                    4 * rows[3 * dw + x] + rows[4 * dw + x];
            d[x] = (uint8_t)((v + 128) >> 8);
        }
    }
    free(rows);
    free(tabx);
}

void orc_decimate(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst,
                  int dw, int dh, int dstride)
{
    /* cv::resize(src, dst, Size(), 0.5, 0.5, INTER_NEAREST):
     * sx = min(floor(x * 2), sw-1)   (src/dataset.cpp:126-129) */
    for (int y = 0; y < dh; ++y) {
        int sy = 2 * y < sh - 1 ? 2 * y : sh - 1;
        for (int x = 0; x < dw; ++x) {
            int sx = 2 * x < sw - 1 ? 2 * x : sw - 1;
            dst[(size_t)y * dstride + x] = src[(size_t)sy * sstride + sx];
        }
    }
}

static void plane_alloc(orc_plane *pl, int w, int h, int border)
{
    pl->w = w; pl->h = h; pl->border = border;
    pl->stride = w + 2 * border;
    pl->base = (uint8_t *)malloc((size_t)pl->stride * (h + 2 * border));
    pl->data = pl->base + (size_t)border * pl->stride + border;
}

static void plane_fill_border(orc_plane *pl)
{
    /* copyMakeBorder(..., BORDER_REFLECT_101) */
    int b = pl->border, w = pl->w, h = pl->h, st = pl->stride;
    for (int y = 0; y < h; ++y) {
        uint8_t *r = pl->data + (size_t)y * st;
        for (int x = 1; x <= b; ++x) {
            r[-x] = r[reflect101(-x, w)];
            r[w - 1 + x] = r[reflect101(w - 1 + x, w)];
        }
    }
    for (int y = 1; y <= b; ++y) {
        memcpy(pl->data + (size_t)(-y) * st - b,
               pl->data + (size_t)reflect101(-y, h) * st - b, (size_t)st);
        memcpy(pl->data + (size_t)(h - 1 + y) * st - b,
               pl->data + (size_t)reflect101(h - 1 + y, h) * st - b, (size_t)st);
    }
}

void orc_pyr_build(orc_pyr *p, const uint8_t *img, int stride, int w, int h,
                   int max_level, int win)
{
    /* buildOpticalFlowPyramid(img, pyr, winSize, maxLevel, withDerivatives=false,
     * pyrBorder=BORDER_REFLECT_101): every level carries a win-pixel border. */
    memset(p, 0, sizeof(*p));
    int lw = w, lh = h;
    for (int level = 0; level <= max_level && level < ORC_MAX_LEVELS; ++level) {
        plane_alloc(&p->lv[level], lw, lh, win);
        if (level == 0) {
            for (int y = 0; y < h; ++y)
                memcpy(p->lv[0].data + (size_t)y * p->lv[0].stride,
                       img + (size_t)y * stride, (size_t)w);
        } else {
            orc_pyrdown(p->lv[level - 1].data, p->lv[level - 1].w,
                        p->lv[level - 1].h, p->lv[level - 1].stride,
                        p->lv[level].data, p->lv[level].stride);
        }
        plane_fill_border(&p->lv[level]);
        p->nlevels = level + 1;
        lw = (lw + 1) / 2; lh = (lh + 1) / 2;
        if (lw <= win || lh <= win) break;
    }
}

void orc_pyr_free(orc_pyr *p)
{
    for (int i = 0; i < ORC_MAX_LEVELS; ++i) { free(p->lv[i].base); p->lv[i].base = 0; }
    p->nlevels = 0;
}

/* ------------------------------------------------------------------ */
/* calcSharrDeriv: t0 = 3(p[y-1]+p[y+1]) + 10 p[y], t1 = p[y+1]-p[y-1];     */
/* dx = t0[x+1]-t0[x-1], dy = 3(t1[x-1]+t1[x+1]) + 10 t1[x]; REFLECT_101     */
void orc_scharr(const uint8_t *src, int w, int h, int stride, int16_t *out)
{
    int *t0 = (int *)malloc(sizeof(int) * (size_t)(w + 2));
    int *t1 = (int *)malloc(sizeof(int) * (size_t)(w + 2));
    for (int y = 0; y < h; ++y) {
        const uint8_t *r0 = src + (size_t)reflect101(y - 1, h) * stride;
        const uint8_t *r1 = src + (size_t)y * stride;
        const uint8_t *r2 = src + (size_t)reflect101(y + 1, h) * stride;
        int *a = t0 + 1, *b = t1 + 1;
        for (int x = 0; x < w; ++x) {
            a[x] = (r0[x] + r2[x]) * 3 + r1[x] * 10;
            b[x] = r2[x] - r0[x];
        }
        a[-1] = a[reflect101(-1, w)]; a[w] = a[reflect101(w, w)];
        b[-1] = b[reflect101(-1, w)]; b[w] = b[reflect101(w, w)];
        int16_t *d = out + (size_t)y * w * 2;
        for (int x = 0; x < w; ++x) {
            d[2 * x] = (int16_t)(a[x + 1] - a[x - 1]);
            d[2 * x + 1] = (int16_t)((b[x + 1] + b[x - 1]) * 3 + b[x] * 10);
        }
    }
    free(t0); free(t1);
}

/* ------------------------------------------------------------------ */
#define W_BITS 14
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

int orc_whatif[ORC_WHATIF_N] = { 0 };       /* svs_oracle.h: what-if knobs, all 0 = the oracle as declared */
void orc_set_whatif(int which, int value) { if (which >= 0 && which < ORC_WHATIF_N) orc_whatif[which] = value; }

static inline int cv_round_f(float v) { return (int)lrintf(v); } /* half-to-even */
static inline int cv_floor_f(float v) { return (int)floorf(v); }

typedef struct { int16_t *data; int w, h, stride; int border; int16_t *base; } dplane;

static void lk_level(const orc_plane *I, const orc_plane *J, const dplane *dI,
                     int level, int max_level, int n, const float *prev_xy,
                     float *next_xy, uint8_t *status, float *err,
                     const orc_lk_params *prm, int max_count, double eps2)
{
    const int win = I->border; /* = 11 */
    const float half = (win - 1) * 0.5f;
    int16_t Ibuf[32 * 32], dIbuf[32 * 32 * 2];
    const float FLT_SCALE = 1.f / (1 << 20);
    const float lscale = (float)(1. / (1 << level));
    for (int pt = 0; pt < n; ++pt) {
        float px = prev_xy[2 * pt] * lscale, py = prev_xy[2 * pt + 1] * lscale;
        float nx, ny;
        if (level == max_level) {
            if (prm->use_initial_flow) {
                nx = next_xy[2 * pt] * lscale; ny = next_xy[2 * pt + 1] * lscale;
            } else { nx = px; ny = py; }
        } else {
            nx = next_xy[2 * pt] * 2.f; ny = next_xy[2 * pt + 1] * 2.f;
        }
        next_xy[2 * pt] = nx; next_xy[2 * pt + 1] = ny;

        px -= half; py -= half;
        int ipx = cv_floor_f(px), ipy = cv_floor_f(py);
        if (ipx < -win || ipx >= I->w || ipy < -win || ipy >= I->h) {
            if (level == 0) { status[pt] = 0; if (err) err[pt] = 0; }
            continue;
        }
        float a = px - ipx, b = py - ipy;
        int iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
        int iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
        int iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
        int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;

        long long sA11 = 0, sA12 = 0, sA22 = 0;
        float fA11 = 0.f, fA12 = 0.f, fA22 = 0.f;       /* what-if 0: f32 accumulators in pixel order */
        for (int y = 0; y < win; ++y) {
            const uint8_t *src = I->data + (ptrdiff_t)(y + ipy) * I->stride + ipx;
            const int16_t *ds = dI->data + ((ptrdiff_t)(y + ipy) * dI->stride + ipx) * 2;
            int sI = I->stride, sD = dI->stride * 2;
            for (int x = 0; x < win; ++x, ds += 2) {
                int ival = DESCALE(src[x] * iw00 + src[x + 1] * iw01 +
                                   src[x + sI] * iw10 + src[x + sI + 1] * iw11, W_BITS - 5);
                int ixval = DESCALE(ds[0] * iw00 + ds[2] * iw01 + ds[sD] * iw10 +
                                    ds[sD + 2] * iw11, W_BITS);
                int iyval = DESCALE(ds[1] * iw00 + ds[3] * iw01 + ds[sD + 1] * iw10 +
                                    ds[sD + 3] * iw11, W_BITS);
                Ibuf[y * win + x] = (int16_t)ival;
                dIbuf[(y * win + x) * 2] = (int16_t)ixval;
                dIbuf[(y * win + x) * 2 + 1] = (int16_t)iyval;
                sA11 += (long long)ixval * ixval;
                sA12 += (long long)ixval * iyval;
                sA22 += (long long)iyval * iyval;
                fA11 += (float)(ixval * ixval); fA12 += (float)(ixval * iyval); fA22 += (float)(iyval * iyval);
            }
        }
        /* declared order: exact integer sums, one conversion to float */
        float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE,
              A22 = (float)sA22 * FLT_SCALE;
        if (orc_whatif[0]) { A11 = fA11 * FLT_SCALE; A12 = fA12 * FLT_SCALE; A22 = fA22 * FLT_SCALE; }
        float D = A11 * A22 - A12 * A12;
        float dd = A11 - A22;
        float minEig = (A22 + A11 - sqrtf(dd * dd + 4.f * A12 * A12)) /
                       (float)(2 * win * win);
        if ((double)minEig < prm->min_eig_thr || D < FLT_EPSILON) {
            if (level == 0) status[pt] = 0;
            continue;
        }
        D = 1.f / D;
        nx -= half; ny -= half;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < max_count; ++j) {
            int inx = cv_floor_f(nx), iny = cv_floor_f(ny);
            if (inx < -win || inx >= J->w || iny < -win || iny >= J->h) {
                if (level == 0) status[pt] = 0;
                break;
            }
            a = nx - inx; b = ny - iny;
            iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
            iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
            iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            long long sb1 = 0, sb2 = 0;
            float fb1 = 0.f, fb2 = 0.f;
            for (int y = 0; y < win; ++y) {
                const uint8_t *Jp = J->data + (ptrdiff_t)(y + iny) * J->stride + inx;
                int sJ = J->stride;
                for (int x = 0; x < win; ++x) {
                    int diff = DESCALE(Jp[x] * iw00 + Jp[x + 1] * iw01 + Jp[x + sJ] * iw10 +
                                       Jp[x + sJ + 1] * iw11, W_BITS - 5) - Ibuf[y * win + x];
                    sb1 += (long long)diff * dIbuf[(y * win + x) * 2];
                    sb2 += (long long)diff * dIbuf[(y * win + x) * 2 + 1];
                    fb1 += (float)(diff * dIbuf[(y * win + x) * 2]); fb2 += (float)(diff * dIbuf[(y * win + x) * 2 + 1]);
                }
            }
            float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
            if (orc_whatif[0]) { b1 = fb1 * FLT_SCALE; b2 = fb2 * FLT_SCALE; }
            float dx = (A12 * b2 - A22 * b1) * D;
            float dy = (A12 * b1 - A11 * b2) * D;
            nx += dx; ny += dy;
            next_xy[2 * pt] = nx + half; next_xy[2 * pt + 1] = ny + half;
            if ((double)dx * dx + (double)dy * dy <= eps2) break;
            if (j > 0 && (double)fabsf(dx + pdx) < 0.01 && (double)fabsf(dy + pdy) < 0.01) {
                next_xy[2 * pt] -= dx * 0.5f; next_xy[2 * pt + 1] -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (status[pt] && err && level == 0) {
            float fx = next_xy[2 * pt] - half, fy = next_xy[2 * pt + 1] - half;
            int inx = cv_floor_f(fx), iny = cv_floor_f(fy);
            if (inx < -win || inx >= J->w || iny < -win || iny >= J->h) {
                status[pt] = 0;
                continue;
            }
            float aa = fx - inx, bb = fy - iny;
            iw00 = cv_round_f((1.f - aa) * (1.f - bb) * (1 << W_BITS));
            iw01 = cv_round_f(aa * (1.f - bb) * (1 << W_BITS));
            iw10 = cv_round_f((1.f - aa) * bb * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            long long serr = 0; /* |diff| summed exactly, converted once */
            for (int y = 0; y < win; ++y) {
                const uint8_t *Jp = J->data + (ptrdiff_t)(y + iny) * J->stride + inx;
                int sJ = J->stride;
                for (int x = 0; x < win; ++x) {
                    int diff = DESCALE(Jp[x] * iw00 + Jp[x + 1] * iw01 + Jp[x + sJ] * iw10 +
                                       Jp[x + sJ + 1] * iw11, W_BITS - 5) - Ibuf[y * win + x];
                    serr += diff < 0 ? -diff : diff;
                }
            }
            err[pt] = (float)serr * 1.f / (float)(32 * win * win);
        }
    }
}

void orc_lk_pyr(const orc_pyr *prev, const orc_pyr *next, int n,
                const float *prev_xy, float *next_xy, uint8_t *status,
                float *err, const orc_lk_params *prm)
{
    int max_level = prm->max_level;
    if (max_level > prev->nlevels - 1) max_level = prev->nlevels - 1;
    if (max_level > next->nlevels - 1) max_level = next->nlevels - 1;
    int max_count = prm->max_iter < 0 ? 0 : (prm->max_iter > 100 ? 100 : prm->max_iter);
    double eps = prm->epsilon < 0 ? 0 : (prm->epsilon > 10. ? 10. : prm->epsilon);
    double eps2 = eps * eps;
    for (int i = 0; i < n; ++i) { status[i] = 1; if (err) err[i] = 0; }
    const int win = prev->lv[0].border;
    for (int level = max_level; level >= 0; --level) {
        const orc_plane *I = &prev->lv[level];
        /* derivative plane with zero border (BORDER_CONSTANT) */
        dplane d;
        d.w = I->w; d.h = I->h; d.border = win; d.stride = I->w + 2 * win;
        d.base = (int16_t *)calloc((size_t)d.stride * (I->h + 2 * win) * 2, sizeof(int16_t));
        d.data = d.base + ((size_t)win * d.stride + win) * 2;
        int16_t *tight = (int16_t *)malloc(sizeof(int16_t) * 2 * (size_t)I->w * I->h);
        orc_scharr(I->data, I->w, I->h, I->stride, tight);
        for (int y = 0; y < I->h; ++y)
            memcpy(d.data + (size_t)y * d.stride * 2, tight + (size_t)y * I->w * 2,
                   sizeof(int16_t) * 2 * (size_t)I->w);
        free(tight);
        lk_level(I, &next->lv[level], &d, level, max_level, n, prev_xy, next_xy,
                 status, err, prm, max_count, eps2);
        free(d.base);
    }
}

void orc_lk(const uint8_t *prev, int pstride, const uint8_t *next, int nstride,
            int w, int h, int n, const float *prev_xy, float *next_xy,
            uint8_t *status, float *err, const orc_lk_params *p)
{
    /* like cv::calcOpticalFlowPyrLK on two plain images: both pyramids are
     * rebuilt on every call (the reference never passes prebuilt pyramids) */
    orc_pyr P, N;
    orc_pyr_build(&P, prev, pstride, w, h, p->max_level, 11);
    orc_pyr_build(&N, next, nstride, w, h, p->max_level, 11);
    orc_lk_pyr(&P, &N, n, prev_xy, next_xy, status, err, p);
    orc_pyr_free(&P);
    orc_pyr_free(&N);
}
