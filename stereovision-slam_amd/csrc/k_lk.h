// k_lk.h — pyramidal Lucas-Kanade tracker, one 64-lane wavefront per point.
// Replaces cv::calcOpticalFlowPyrLK (OpenCV lkpyramid.cpp LKTrackerInvoker) at
// reference src/frontend.cpp:353-357 (TrackLastFrame) and :105-109
// (FindFeaturesInRight).  Mirrors oracle/orc_image.c:lk_level operation for
// operation; integer patches and the exact-integer normal-equation sums make
// the result bit-exact against the oracle.
//
// Mapping: window 11x11 = 121 pixels -> 2 pixels per lane.  Per level the wave
//   1. stages the 14x14 I neighbourhood (HBM -> LDS, rows of the stored-border
//      pyramid, no index arithmetic),
//   2. computes the Scharr derivatives of the 12x12 inner positions on the fly
//      (the reference materialises a full int16x2 derivative image per level —
//      5.3 bytes/pixel of HBM traffic that is never needed),
//   3. interpolates its two I/Ix/Iy samples (Q14 weights) into registers and
//      wave-reduces A11,A12,A22 (int32, exact),
//   4. stages a 32x32 J search region into LDS with aligned dword loads and
//      iterates entirely out of LDS (re-staging only if the window leaves the
//      region), wave-reducing b1,b2 as exact int64.
// All control flow is wave-uniform; waves never synchronise with each other.
#pragma once
#include "dev_common.h"

struct LkJob { int prev_slot, next_slot, pt_ofs, npts; };
struct LkParams {
    int max_level;
    int max_count;
    double eps2;
    double min_eig_thr;
    int use_initial_flow;
};

#define LK_WIN 11
#define LK_NPIX 121
#define LK_W_BITS 14
#define LK_DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))
#define LK_REG 32
#define LK_WAVES_PER_BLOCK 4

__device__ __forceinline__ void lk_weights(float a, float b, int &w00, int &w01, int &w10, int &w11)
{
    // cvRound == round-half-even == rintf
    // the inputs are wave-uniform: keep the weights in SGPRs
    w00 = __builtin_amdgcn_readfirstlane((int)rintf((1.f - a) * (1.f - b) * (float)(1 << LK_W_BITS)));
    w01 = __builtin_amdgcn_readfirstlane((int)rintf(a * (1.f - b) * (float)(1 << LK_W_BITS)));
    w10 = __builtin_amdgcn_readfirstlane((int)rintf((1.f - a) * b * (float)(1 << LK_W_BITS)));
    w11 = (1 << LK_W_BITS) - w00 - w01 - w10;
}

__device__ __forceinline__ void lk_stage_J(uint32_t *sJ, const uint8_t *J0, int pitch, int w, int h,
                                           int cx, int cy, int lane, int &rx0, int &ry0)
{
    rx0 = __builtin_amdgcn_readfirstlane((cx - 10) & ~3);
    ry0 = __builtin_amdgcn_readfirstlane(cy - 10);
    const int gxmax = (w + SVS_BORDER - 4) & ~3;
#pragma unroll
    for (int i = lane; i < LK_REG * (LK_REG / 4); i += 64) {
        int r = i >> 3, c4 = i & 7;
        int gy = ry0 + r, gx = rx0 + c4 * 4;
        gy = max(-SVS_BORDER, min(gy, h + SVS_BORDER - 1));
        gx = max(-SVS_BORDER, min(gx, gxmax));
        sJ[i] = *reinterpret_cast<const uint32_t *>(J0 + (ptrdiff_t)gy * pitch + gx);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int lk_sample_J(const uint8_t *sJb, int o, int w00, int w01, int w10, int w11)
{
    int v = sJb[o] * w00 + sJb[o + 1] * w01 + sJb[o + LK_REG] * w10 + sJb[o + LK_REG + 1] * w11;
    return LK_DESCALE(v, LK_W_BITS - 5);
}

__global__ void __launch_bounds__(64 * LK_WAVES_PER_BLOCK)
k_lk(const LkJob *jobs, const uint8_t *pyr, PyrGeom g, const float2 *prev_xy, float2 *next_xy,
     uint8_t *status, float *err, LkParams prm)
{
    __shared__ uint8_t sI_all[LK_WAVES_PER_BLOCK][14 * 16];
    __shared__ uint32_t sD_all[LK_WAVES_PER_BLOCK][144];
    __shared__ uint32_t sJ_all[LK_WAVES_PER_BLOCK][LK_REG * LK_REG / 4];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const LkJob jb = jobs[blockIdx.y];
    const int pi = blockIdx.x * LK_WAVES_PER_BLOCK + wave;
    if (pi >= jb.npts) return;
    const int pt = jb.pt_ofs + pi;
    uint8_t *sI = sI_all[wave];
    uint32_t *sD = sD_all[wave];
    uint32_t *sJ = sJ_all[wave];
    const uint8_t *sJb = reinterpret_cast<const uint8_t *>(sJ);

    const uint8_t *slotI = pyr + (size_t)jb.prev_slot * g.slot_bytes;
    const uint8_t *slotJ = pyr + (size_t)jb.next_slot * g.slot_bytes;

    const float2 prevp = prev_xy[pt];
    float2 nextp = next_xy[pt];
    bool st = true;
    float errv = 0.f;

    // this lane's two window pixels
    const int p0 = lane, p1 = lane + 64;
    const int wy0 = p0 / LK_WIN, wx0 = p0 - wy0 * LK_WIN;
    const int wy1 = p1 / LK_WIN, wx1 = p1 - wy1 * LK_WIN;
    const bool has1 = p1 < LK_NPIX;
    const float FLT_SCALE = 1.f / (float)(1 << 20);

    int max_level = prm.max_level;
    if (max_level > g.nlevels - 1) max_level = g.nlevels - 1;

    for (int level = max_level; level >= 0; --level) {
        const int w = g.w[level], h = g.h[level], pitch = g.pitch[level];
        const uint8_t *I0 = lvl_origin(slotI, g, level);
        const uint8_t *J0 = lvl_origin(slotJ, g, level);
        const float lscale = 1.f / (float)(1 << level);
        float px = prevp.x * lscale, py = prevp.y * lscale;
        float nx, ny;
        if (level == max_level) {
            if (prm.use_initial_flow) { nx = nextp.x * lscale; ny = nextp.y * lscale; }
            else { nx = px; ny = py; }
        } else { nx = nextp.x * 2.f; ny = nextp.y * 2.f; }
        nextp.x = nx; nextp.y = ny;

        px -= 5.f; py -= 5.f;
        const int ipx = __builtin_amdgcn_readfirstlane((int)floorf(px)), ipy = __builtin_amdgcn_readfirstlane((int)floorf(py));
        if (ipx < -LK_WIN || ipx >= w || ipy < -LK_WIN || ipy >= h) {
            if (level == 0) { st = false; errv = 0.f; }
            continue;
        }
        int iw00, iw01, iw10, iw11;
        lk_weights(px - (float)ipx, py - (float)ipy, iw00, iw01, iw10, iw11);

        // 1. stage I neighbourhood: rows ipy-1..ipy+12, cols ipx-1..ipx+12
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < 14 * 14; i += 64) {
            int r = i / 14, c = i - r * 14;
            sI[r * 16 + c] = I0[(ptrdiff_t)(ipy - 1 + r) * pitch + (ipx - 1 + c)];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // 2. Scharr at the 12x12 positions (ipx+c, ipy+r); zero outside the image
        for (int i = lane; i < 144; i += 64) {
            int r = i / 12, c = i - r * 12;
            const uint8_t *q = sI + r * 16 + c; // top-left of the 3x3 around (r+1,c+1)
            int a0 = q[0], a1 = q[1], a2 = q[2];
            int b0 = q[16], b2 = q[18];
            int c0 = q[32], c1 = q[33], c2 = q[34];
            int t0m = (a0 + c0) * 3 + b0 * 10;
            int t0p = (a2 + c2) * 3 + b2 * 10;
            int t1m = c0 - a0, t1c = c1 - a1, t1p = c2 - a2;
            int dx = t0p - t0m;
            int dy = (t1p + t1m) * 3 + t1c * 10;
            int gx = ipx + c, gy = ipy + r;
            if (gx < 0 || gx >= w || gy < 0 || gy >= h) { dx = 0; dy = 0; }
            sD[i] = ((uint32_t)dx & 0xffffu) | ((uint32_t)dy << 16);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // 3. interpolate the two samples of this lane
        int iv0, ix0, iy0, iv1 = 0, ix1 = 0, iy1 = 0;
        {
            const uint8_t *q = sI + (wy0 + 1) * 16 + wx0 + 1;
            iv0 = LK_DESCALE(q[0] * iw00 + q[1] * iw01 + q[16] * iw10 + q[17] * iw11, LK_W_BITS - 5);
            uint32_t d00 = sD[wy0 * 12 + wx0], d01 = sD[wy0 * 12 + wx0 + 1];
            uint32_t d10 = sD[(wy0 + 1) * 12 + wx0], d11 = sD[(wy0 + 1) * 12 + wx0 + 1];
            ix0 = LK_DESCALE((int)(short)(d00 & 0xffff) * iw00 + (int)(short)(d01 & 0xffff) * iw01 +
                             (int)(short)(d10 & 0xffff) * iw10 + (int)(short)(d11 & 0xffff) * iw11, LK_W_BITS);
            iy0 = LK_DESCALE(((int)d00 >> 16) * iw00 + ((int)d01 >> 16) * iw01 +
                             ((int)d10 >> 16) * iw10 + ((int)d11 >> 16) * iw11, LK_W_BITS);
        }
        if (has1) {
            const uint8_t *q = sI + (wy1 + 1) * 16 + wx1 + 1;
            iv1 = LK_DESCALE(q[0] * iw00 + q[1] * iw01 + q[16] * iw10 + q[17] * iw11, LK_W_BITS - 5);
            uint32_t d00 = sD[wy1 * 12 + wx1], d01 = sD[wy1 * 12 + wx1 + 1];
            uint32_t d10 = sD[(wy1 + 1) * 12 + wx1], d11 = sD[(wy1 + 1) * 12 + wx1 + 1];
            ix1 = LK_DESCALE((int)(short)(d00 & 0xffff) * iw00 + (int)(short)(d01 & 0xffff) * iw01 +
                             (int)(short)(d10 & 0xffff) * iw10 + (int)(short)(d11 & 0xffff) * iw11, LK_W_BITS);
            iy1 = LK_DESCALE(((int)d00 >> 16) * iw00 + ((int)d01 >> 16) * iw01 +
                             ((int)d10 >> 16) * iw10 + ((int)d11 >> 16) * iw11, LK_W_BITS);
        }
        const int sA11 = wave_sum_i32(ix0 * ix0 + ix1 * ix1);
        const int sA12 = wave_sum_i32(ix0 * iy0 + ix1 * iy1);
        const int sA22 = wave_sum_i32(iy0 * iy0 + iy1 * iy1);
        const float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float dd = A11 - A22;
        const float minEig = (A22 + A11 - sqrtf(dd * dd + 4.f * A12 * A12)) / (float)(2 * LK_WIN * LK_WIN);
        if ((double)minEig < prm.min_eig_thr || D < 1.1920928955078125e-07f) {
            if (level == 0) st = false;
            continue;
        }
        D = 1.f / D;
        nx -= 5.f; ny -= 5.f;
        float pdx = 0.f, pdy = 0.f;
        int rx0, ry0;
        lk_stage_J(sJ, J0, pitch, w, h, (int)floorf(nx), (int)floorf(ny), lane, rx0, ry0);

        for (int j = 0; j < prm.max_count; ++j) {
            const int inx = __builtin_amdgcn_readfirstlane((int)floorf(nx)), iny = __builtin_amdgcn_readfirstlane((int)floorf(ny));
            if (inx < -LK_WIN || inx >= w || iny < -LK_WIN || iny >= h) {
                if (level == 0) st = false;
                break;
            }
            int ox = inx - rx0, oy = iny - ry0;
            if (ox < 0 || ox > LK_REG - 12 || oy < 0 || oy > LK_REG - 12) {
                lk_stage_J(sJ, J0, pitch, w, h, inx, iny, lane, rx0, ry0);
                ox = inx - rx0; oy = iny - ry0;
            }
            lk_weights(nx - (float)inx, ny - (float)iny, iw00, iw01, iw10, iw11);
            int d0 = lk_sample_J(sJb, (oy + wy0) * LK_REG + ox + wx0, iw00, iw01, iw10, iw11) - iv0;
            int pb1 = d0 * ix0, pb2 = d0 * iy0;
            if (has1) {
                int d1 = lk_sample_J(sJb, (oy + wy1) * LK_REG + ox + wx1, iw00, iw01, iw10, iw11) - iv1;
                pb1 += d1 * ix1; pb2 += d1 * iy1;
            }
            const long long sb1 = wave_sum_i64((long long)pb1);
            const long long sb2 = wave_sum_i64((long long)pb2);
            const float b1 = (float)(double)sb1 * FLT_SCALE, b2 = (float)(double)sb2 * FLT_SCALE;
            const float dx = (A12 * b2 - A22 * b1) * D;
            const float dy = (A12 * b1 - A11 * b2) * D;
            nx += dx; ny += dy;
            nextp.x = nx + 5.f; nextp.y = ny + 5.f;
            if ((double)dx * (double)dx + (double)dy * (double)dy <= prm.eps2) break;
            if (j > 0 && (double)fabsf(dx + pdx) < 0.01 && (double)fabsf(dy + pdy) < 0.01) {
                nextp.x -= dx * 0.5f; nextp.y -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }

        if (st && level == 0) {
            // level-0 residual ("err" output); can still clear status
            const float fx = nextp.x - 5.f, fy = nextp.y - 5.f;
            const int inx = (int)floorf(fx), iny = (int)floorf(fy);
            if (inx < -LK_WIN || inx >= w || iny < -LK_WIN || iny >= h) {
                st = false;
                continue;
            }
            int ox = inx - rx0, oy = iny - ry0;
            if (ox < 0 || ox > LK_REG - 12 || oy < 0 || oy > LK_REG - 12) {
                lk_stage_J(sJ, J0, pitch, w, h, inx, iny, lane, rx0, ry0);
                ox = inx - rx0; oy = iny - ry0;
            }
            lk_weights(fx - (float)inx, fy - (float)iny, iw00, iw01, iw10, iw11);
            int d0 = lk_sample_J(sJb, (oy + wy0) * LK_REG + ox + wx0, iw00, iw01, iw10, iw11) - iv0;
            int e = d0 < 0 ? -d0 : d0;
            if (has1) {
                int d1 = lk_sample_J(sJb, (oy + wy1) * LK_REG + ox + wx1, iw00, iw01, iw10, iw11) - iv1;
                e += d1 < 0 ? -d1 : d1;
            }
            const int serr = wave_sum_i32(e);
            errv = (float)serr * 1.f / (float)(32 * LK_WIN * LK_WIN);
        }
    }
    if (lane == 0) {
        next_xy[pt] = nextp;
        status[pt] = st ? 1 : 0;
        if (err) err[pt] = errv;
    }
}
