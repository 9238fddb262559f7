// k_gftt.h — Shi-Tomasi corner detection with the reference's feature mask.
// Replaces Frontend::DetectFeatures' mask construction + cv::GFTTDetector
// (OpenCV goodFeaturesToTrack: cornerMinEigenVal -> masked max -> threshold ->
// 3x3 NMS -> sort -> greedy min-distance) at reference src/frontend.cpp:42-51.
// Mirrors oracle/orc_gftt.c; the f32 operation order is the declared one
// (-ffp-contract=off, IEEE sqrt), the 3x3 box sums in f64 follow its order too (GeCov below), so the
// corner list (coordinates, order, count) is bit-exact against the oracle.
//
// Two kernels per batch (grid.z / grid.x = job); the image is read ONCE and nothing
// image-sized is ever written:
//   k_gftt_eig3   one wave = 64 image columns walking down a strip of rows: Sobel ->
//                 covariance -> 3x3 f64 box -> min-eigenvalue -> 3x3 non-max test, all on
//                 register windows (three rows each of pixels, covariance products and
//                 eigenvalues; horizontal neighbours by DPP wave shifts).  The feature mask of
//                 src/frontend.cpp:42-47 is never rasterised: the few 21x21 squares that touch
//                 a wave's strip become one bit per row in two VGPRs per lane.  Outputs: the
//                 masked maximum (one atomic per wave) and the local maxima as 64-bit keys
//                 (ordered value << 32 | pixel index), buffered in LDS and appended with one
//                 atomic per wave, already thinned by the lower bound of the quality threshold
//                 that the maxima seen so far imply.
//   k_gftt_select2 one workgroup per job: exact threshold, then top-K selection instead of
//                 a full sort — histogram of the candidates' values (2048 log-spaced bins
//                 between the threshold and the maximum), the best <= 2048 are sorted
//                 (bitonic, LDS) and fed to the order-dependent greedy min-distance pass;
//                 only if that pass runs dry before max_corners does the next slice follow.
//                 Resets the job's counters for the next call.
#pragma once
#include "dev_common.h"

#define GF_CNT_STRIDE 32
struct GfttJob { int slot, rect_ofs, nrect; };

struct GfttWork {            // per-job scratch in HBM
    unsigned long long *keys;// [jobs][cap] candidate keys
    unsigned int *counters;  // [jobs][GF_CNT_STRIDE]: 0 = ordered masked max, 1 = ncand (one 128 B line per job);
                             // zero between calls (k_gftt_select2 leaves them so)
    int cap;                 // key capacity per job (power of two >= w*h)
    long long *prof;         // development: phase ticks of job 0's selection (SVSLAM_GFTT_PROF), else null
};

__device__ __forceinline__ unsigned int f32_ordered(float v)
{
    unsigned int b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float f32_from_ordered(unsigned int k)
{
    unsigned int b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(b);
}
// (float)(maxVal * qualityLevel) of goodFeaturesToTrack as an ordered key; mk = ordered masked max (0: none)
__device__ __forceinline__ unsigned int gf_threshold_ordered(unsigned int mk, double quality)
{
    const double maxVal = mk ? (double)f32_from_ordered(mk) : 0.0;
    return f32_ordered((float)(maxVal * quality));
}

// ---- fused min-eigenvalue + non-max suppression + mask -------------------------------------
// One wave carries 64 image columns (58 outputs + 3 halo columns each side: 1 Sobel, 1 box, 1 NMS)
// and walks down GE_ROWS + 6 rows.  Horizontal neighbours come from DPP wave shifts (gfx9
// wave_shr / wave_shl), vertical neighbours from three-row register windows.  Same arithmetic, same
// order of operations as the oracle; REFLECT_101 of the covariance map (OpenCV's boxFilter border)
// is evaluated at the reflected pixel: with the stored border that is the mirrored neighbourhood,
// i.e. left/right or top/bottom swapped.
#define GE_COLS 58
#define GE_ROWS 48
#define GE_CBUF 256                 // candidate keys buffered per wave between flushes
#define SVS_DPP_WAVE_SHR1 0x138
#define SVS_DPP_WAVE_SHL1 0x130
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
struct GePix { float d, hs; };      // per pixel row: right - left, and (s*left + 2s*mid) + s*right (the two Sobel row passes)
// covariance products of (left, this, right) summed in f64, in that order: box_filter's RowSum; the three rows are then added
// top to bottom (its ColumnSum) — the order oracle/orc_gftt.c declares, so the f32 eigenvalue is the oracle's bit for bit.
// (Round 5; before, both sides summed the nine products row-major: 24 f64 additions per pixel and a window of 27 doubles
// instead of 12 and 9.  The sums are exact — and any order the same — for all but ~1e-7 of the pixels, where a dy that is
// a rounding residue sits beside large products.)
struct GeCov { double xx, xy, yy; };
struct GeEig { float m, hm; };      // value, max over (left, value, right)
template <int CTRL> __device__ __forceinline__ double dpp_f64x(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

template <bool STORE_EIG>
__global__ void __launch_bounds__(64)
k_gftt_eig3(const GfttJob *jobs, int njobs, const uint8_t *pyr, PyrGeom g, GfttWork wk, const float2 *rect_xy, double quality,
            float *eig_out)
{
    __shared__ unsigned long long sKeys[GE_CBUF + 64];
    const int w = g.w[0], h = g.h[0], pitch = g.pitch[0];
    // grid: 8 x strips-per-image x ceil(jobs / 8) workgroups; consecutive ids go to consecutive XCDs, so job j lives
    // on XCD j % 8 and its strips find each other's halos in that L2
    const int nsx = (w + GE_COLS - 1) / GE_COLS, nsy = (h + GE_ROWS - 1) / GE_ROWS, per = nsx * nsy;
    const int slot = blockIdx.x >> 3, round = slot / per, strip = slot - round * per;
    const int job = round * 8 + (blockIdx.x & 7);
    if (job >= njobs) return;
    const int bx = strip % nsx, by = strip / nsx;
    const GfttJob jb = jobs[job];
    const uint8_t *img = lvl_origin(pyr + (size_t)jb.slot * g.slot_bytes, g, 0);
    unsigned int *ctr = wk.counters + (size_t)job * GF_CNT_STRIDE;
    unsigned long long *gkeys = wk.keys + (size_t)job * wk.cap;
    const int lane = threadIdx.x;
    const int x0 = bx * GE_COLS, y0 = by * GE_ROWS;
    const int gx = x0 - 3 + lane;                               // column of this lane's pixel / covariance / eigenvalue
    const uint8_t *colp = img + min(gx, w + SVS_BORDER - 1);
    const bool col_out = gx < 0 || gx >= w;
    const bool col_zero = gx > w;                               // beyond the 1-px covariance halo
    const bool col_img = gx >= 0 && gx < w && lane >= 2 && lane < 62;        // eigenvalue valid and inside the image
    const bool col_own = lane >= 3 && lane < 3 + GE_COLS && gx < w;          // this wave's output columns
    const bool col_cand = col_own && gx >= 1 && gx <= w - 2;
    const float s1 = (float)(1.0 / 3060.0);
    const float s2 = (float)(2.0 * (1.0 / 3060.0));

    // ---- feature mask (src/frontend.cpp:42-47: cv::rectangle(mask, pt - (10,10), pt + (10,10), 0, FILLED),
    // Point2f -> Point rounds half to even) as row bits: bit r of (mlo, mhi) set <=> pixel (gx, y0 - 1 + r) is masked
    uint32_t mlo = 0, mhi = 0;
    {
        const int tx1 = x0 - 3, tx2 = x0 + 60, ty1 = y0 - 1, ty2 = y0 + GE_ROWS;
        for (int b = 0; b < jb.nrect; b += 64) {
            const int ri = b + lane;
            int x1 = 1, x2 = 0, y1 = 1, y2 = 0;
            if (ri < jb.nrect) {
                const float2 c = rect_xy[jb.rect_ofs + ri];
                x1 = (int)rintf(c.x - 10.f); y1 = (int)rintf(c.y - 10.f);
                x2 = (int)rintf(c.x + 10.f); y2 = (int)rintf(c.y + 10.f);
            }
            const bool hit = ri < jb.nrect && x1 <= tx2 && x2 >= tx1 && y1 <= ty2 && y2 >= ty1;
            unsigned long long m = __ballot(hit);
            while (m) {
                const int l = __ffsll((long long)m) - 1;
                m &= m - 1;
                const int rx1 = __builtin_amdgcn_readlane(x1, l), rx2 = __builtin_amdgcn_readlane(x2, l);
                const int ry1 = __builtin_amdgcn_readlane(y1, l), ry2 = __builtin_amdgcn_readlane(y2, l);
                const int r1 = max(ry1 - ty1, 0), r2 = min(ry2 - ty1, GE_ROWS + 1);      // inclusive rows of the strip
                const unsigned long long bits = (((r2 - r1 + 1) >= 64 ? ~0ull : ((1ull << (r2 - r1 + 1)) - 1ull)) << r1);
                if (gx >= rx1 && gx <= rx2) { mlo |= (uint32_t)bits; mhi |= (uint32_t)(bits >> 32); }
            }
        }
    }
    auto masked = [&](int r) -> bool {                           // r wave-uniform
        const uint32_t word = r < 32 ? mlo : mhi;
        return (word >> (r & 31)) & 1u;
    };

    GePix Pw[3];
    GeCov Cw[3];
    GeEig Ew[3];
    float best = -INFINITY;                                      // masked maximum of this lane, -inf = nothing yet
    int cnt = 0;                                                 // buffered candidates (wave-uniform)

    // append the buffered candidates that can still pass the quality threshold
    auto flush = [&]() {
        unsigned int wb = best == -INFINITY ? 0u : f32_ordered(best);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) wb = max(wb, (unsigned int)__shfl_xor((int)wb, o, 64));
        unsigned int seen = 0;
        if (lane == 0) seen = wb ? max(atomicMax(&ctr[0], wb), wb) : __hip_atomic_load(&ctr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        seen = (unsigned int)__builtin_amdgcn_readfirstlane((int)seen);
        // every maximum seen so far bounds the final one from below, hence the final threshold too
        const unsigned int lo = gf_threshold_ordered(seen, quality);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int kept = 0;
        for (int i0 = 0; i0 < cnt; i0 += 64) {
            const bool k = i0 + lane < cnt && (unsigned int)(sKeys[i0 + lane] >> 32) > lo;
            kept += __popcll(__ballot(k));
        }
        if (kept) {
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(&ctr[1], (unsigned int)kept);
            base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
            for (int i0 = 0; i0 < cnt; i0 += 64) {
                unsigned long long key = 0;
                bool k = false;
                if (i0 + lane < cnt) { key = sKeys[i0 + lane]; k = (unsigned int)(key >> 32) > lo; }
                const unsigned long long bm = __ballot(k);
                const unsigned int pos = base + __builtin_amdgcn_mbcnt_hi((unsigned int)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)bm, 0u));
                if (k && pos < (unsigned int)wk.cap) gkeys[pos] = key;
                base += (unsigned int)__popcll(bm);
            }
        }
        cnt = 0;
        __builtin_amdgcn_wave_barrier();
    };

    auto step = [&](int i, GePix &Pnew, const GePix &Ptop, const GePix &Pmid, GeCov &Cnew, const GeCov &Ctop, const GeCov &Cmid,
                    GeEig &Enew, const GeEig &Etop, const GeEig &Emid) {
        // pixel row y0 - 3 + i: the horizontal halves of the two Sobel passes, once per row
        const int ry = min(y0 - 3 + i, h + SVS_BORDER - 1);
        const float p = (float)colp[(ptrdiff_t)ry * pitch];
        float l = dpp_f32<SVS_DPP_WAVE_SHR1>(p), r = dpp_f32<SVS_DPP_WAVE_SHL1>(p);
        if (col_out) { const float t = l; l = r; r = t; }
        Pnew.d = r - l;
        Pnew.hs = (s1 * l + s2 * p) + s1 * r;
        if (i < 2) return;
        // covariance row cy = y0 - 4 + i from pixel rows (Ptop, Pmid, Pnew); an out-of-image row is the
        // mirrored neighbourhood, i.e. top and bottom swapped (only the sign of dy notices)
        const int cy = y0 - 4 + i;
        const bool row_out = cy < 0 || cy >= h;
        float dx = (Ptop.d + Pnew.d) * s1 + Pmid.d * s2;
        float dy = row_out ? Ptop.hs - Pnew.hs : Pnew.hs - Ptop.hs;
        if (col_zero || cy > h) { dx = 0.f; dy = 0.f; }
        // products in f32 (the reference's order), widened to f64 once; the 3x3 box sums below are f64
        const double xx = (double)(dx * dx), xy = (double)(dx * dy), yy = (double)(dy * dy);
        Cnew.xx = (dpp_f64x<SVS_DPP_WAVE_SHR1>(xx) + xx) + dpp_f64x<SVS_DPP_WAVE_SHL1>(xx);
        Cnew.xy = (dpp_f64x<SVS_DPP_WAVE_SHR1>(xy) + xy) + dpp_f64x<SVS_DPP_WAVE_SHL1>(xy);
        Cnew.yy = (dpp_f64x<SVS_DPP_WAVE_SHR1>(yy) + yy) + dpp_f64x<SVS_DPP_WAVE_SHL1>(yy);
        if (i < 4) return;
        // eigenvalue row oy = y0 - 5 + i from covariance rows (Ctop, Cmid, Cnew): the exact f64 box sum
        const int oy = y0 - 5 + i;
        const double sxx = (Ctop.xx + Cmid.xx) + Cnew.xx, sxy = (Ctop.xy + Cmid.xy) + Cnew.xy, syy = (Ctop.yy + Cmid.yy) + Cnew.yy;
        const float a = (float)sxx * 0.5f, b = (float)sxy, cc = (float)syy * 0.5f;
        const float t = a - cc;
        const float e = (a + cc) - sqrtf(t * t + b * b);
        Enew.m = e;
        Enew.hm = fmaxf(fmaxf(dpp_f32<SVS_DPP_WAVE_SHR1>(e), dpp_f32<SVS_DPP_WAVE_SHL1>(e)), e);
        if (oy >= 0 && oy < h) {                                 // wave-uniform
            if (STORE_EIG && col_own && oy >= y0 && oy < y0 + GE_ROWS) eig_out[(size_t)oy * w + gx] = e;
            // cv::minMaxLoc(eig, 0, &maxVal, 0, 0, mask): every unmasked pixel counts (halo lanes repeat a
            // neighbour's pixels, which a maximum does not mind)
            if (col_img && !masked(i - 4)) best = fmaxf(best, e);
        }
        if (i < 6) return;
        // 3x3 non-max test of row ny = y0 - 6 + i: after cv::threshold(TOZERO) + cv::dilate a pixel
        // survives iff it is above the threshold and no 3x3 neighbour is larger (goodFeaturesToTrack)
        const int ny = y0 - 6 + i;
        const float v = Emid.m;
        bool pass = col_cand && ny >= 1 && ny <= h - 2 && v != 0.f && v >= Etop.hm && v >= Emid.hm && v >= Enew.hm;
        if (pass) pass = !masked(i - 5);
        const unsigned long long bm = __ballot(pass);
        if (bm) {
            if (pass) {
                const unsigned int pos = cnt + __builtin_amdgcn_mbcnt_hi((unsigned int)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)bm, 0u));
                sKeys[pos] = ((unsigned long long)f32_ordered(v) << 32) | (unsigned int)((size_t)ny * w + gx);
            }
            cnt += __popcll(bm);
            if (cnt > GE_CBUF) flush();
        }
    };
    static_assert((GE_ROWS + 6) % 3 == 0, "row loop is unrolled by the window depth");
    static_assert(GE_ROWS + 2 <= 64, "one mask bit per strip row");
    for (int i0 = 0; i0 < GE_ROWS + 6; i0 += 3) {
        // window slot of pixel row i: i % 3; covariance row (i - 2): (i + 1) % 3; eigenvalue row (i - 4): (i + 2) % 3
        step(i0 + 0, Pw[0], Pw[1], Pw[2], Cw[1], Cw[2], Cw[0], Ew[2], Ew[0], Ew[1]);
        step(i0 + 1, Pw[1], Pw[2], Pw[0], Cw[2], Cw[0], Cw[1], Ew[0], Ew[1], Ew[2]);
        step(i0 + 2, Pw[2], Pw[0], Pw[1], Cw[0], Cw[1], Cw[2], Ew[1], Ew[2], Ew[0]);
    }
    flush();                                                     // also publishes this wave's maximum
}

#define GF_MAX_CORNERS 1024
#define GS_THREADS 256
#define GS_BINS 2048
#ifndef GS_CHUNK
#define GS_CHUNK 2048
#endif

__device__ __forceinline__ void bitonic_desc(unsigned long long *a, int n2, int tid, int nthreads)
{
    // sort n2 (power of two) keys descending; a may be LDS or global
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n2; i += nthreads) {
                int ixj = i ^ j;
                if (ixj > i) {
                    unsigned long long x = a[i], y = a[ixj];
                    bool desc = ((i & k) == 0);
                    if (desc ? (x < y) : (x > y)) { a[i] = y; a[ixj] = x; }
                }
            }
            __threadfence_block();
            __syncthreads();
        }
    }
}

// Bitonic sort (descending) of n2 >= 8 keys in LDS, three compare-exchange strides per pass: a thread
// loads the 8 keys that differ in three consecutive index bits, runs up to three network steps on them in
// registers and stores them back — 26 LDS round trips and barriers for 2048 keys instead of 66.
__device__ __forceinline__ void bitonic_desc_lds8(unsigned long long *a, int n2, int tid, int nthreads)
{
    int log2n = 0;
    while ((1 << log2n) < n2) ++log2n;
    for (int lk = 1; lk <= log2n; ++lk) {
        const int k = 1 << lk;
        for (int lj = lk - 1; lj >= 0; lj -= 3) {             // strides 2^lj, 2^(lj-1), 2^(lj-2) (as far as they exist)
            const int nst = min(3, lj + 1);
            const int b3 = min(lj - (nst - 1), log2n - 3);    // lowest of the three varying bits
            const int mtop = 1 << (lj - b3);                  // r-mask of the first stride of this pass
            for (int g = tid; g < (n2 >> 3); g += nthreads) {
                const int base = ((g >> b3) << (b3 + 3)) | (g & ((1 << b3) - 1));
                unsigned long long v[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = a[base + (r << b3)];
#pragma unroll
                for (int mk = 4; mk >= 1; mk >>= 1) {
                    if (mk <= mtop && mk > (mtop >> nst)) {
#pragma unroll
                        for (int r = 0; r < 8; ++r)
                            if ((r & mk) == 0) {
                                // both keys of a pair share bit k of their index (the stride is below k)
                                const bool desc = ((base + (r << b3)) & k) == 0;
                                const unsigned long long x = v[r], y = v[r | mk];
                                const bool sw = desc ? (x < y) : (x > y);
                                v[r] = sw ? y : x; v[r | mk] = sw ? x : y;
                            }
                    }
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) a[base + (r << b3)] = v[r];
            }
            __syncthreads();
        }
    }
}

// The order-dependent greedy pass of goodFeaturesToTrack over keys[0..m) (descending), executed by one
// wave, 64 candidates per step; continues a selection of nacc corners; returns true when max_corners is
// reached.  `lo`: candidates at or below this ordered value end the list (under the quality threshold).
// Accepted corners are kept twice: in acceptance order (accx/accy, the output order) and — like OpenCV's
// own implementation — in a grid of cells of cvRound(minDistance) pixels, so that a candidate is tested
// against the corners of the 3x3 cells around it instead of against all of them.  A cell of that size can
// hold two corners (opposite corners of the cell); three slots are kept, and should a cell ever fill up
// the pass falls back to the full list (same answers, only slower).
#define GF_GRID_CELLS 1280
#define GF_GRID_SLOTS 3
struct GfGrid {
    uint32_t *slots;      // [cells][GF_GRID_SLOTS]  x | y << 16
    uint8_t *count;       // [cells]
    int cell, gw, gh;     // cell size in pixels, grid size; cell == 0: no grid (too many cells or no min-distance)
};
__device__ __forceinline__ bool gf_greedy(const unsigned long long *keys, int m, unsigned int lo, int w, int max_corners,
                                          bool use_dist, double md2, float *accx, float *accy, const GfGrid &gr, bool &grid_ok,
                                          float2 *out, int &nacc, int lane)
{
    for (int base = 0; base < m; base += 64) {
        const int c = base + lane;
        bool alive = c < m;
        float x = 0.f, y = 0.f;
        int xi = 0, yi = 0;
        if (alive) {
            const unsigned long long key = keys[c];
            if ((unsigned int)(key >> 32) <= lo) alive = false;
            const unsigned int idx = (unsigned int)(key & 0xffffffffull);
            yi = idx / w; xi = idx - yi * w;
            x = (float)xi; y = (float)yi;
        }
        int cxi = 0, cyi = 0;
        if (use_dist && grid_ok && gr.cell > 0) {
            cxi = xi / gr.cell; cyi = yi / gr.cell;
            if (alive) {
                const int cx1 = max(cxi - 1, 0), cx2 = min(cxi + 1, gr.gw - 1), cy1 = max(cyi - 1, 0), cy2 = min(cyi + 1, gr.gh - 1);
                for (int yy = cy1; yy <= cy2 && alive; ++yy)
                    for (int xx = cx1; xx <= cx2; ++xx) {
                        const int ci = yy * gr.gw + xx;
                        const int cn = gr.count[ci];
                        for (int k = 0; k < cn; ++k) {
                            const uint32_t pk = gr.slots[ci * GF_GRID_SLOTS + k];
                            const float dx = x - (float)(pk & 0xffffu), dy = y - (float)(pk >> 16);
                            if ((double)(dx * dx + dy * dy) < md2) alive = false;
                        }
                    }
            }
        } else if (use_dist) {
            for (int j = 0; j < nacc; ++j) {
                const float dx = x - accx[j], dy = y - accy[j];
                if ((double)(dx * dx + dy * dy) < md2) alive = false;
            }
        }
        unsigned long long mm = __ballot(alive);
        while (mm) {
            const int l = __ffsll((long long)mm) - 1;
            const float bx = __shfl(x, l, 64), by = __shfl(y, l, 64);
            if (lane == 0) { accx[nacc] = bx; accy[nacc] = by; out[nacc] = make_float2(bx, by); }
            if (use_dist && gr.cell > 0) {
                // insert into the grid (kept up to date even after a fall-back: harmless)
                const int ci = __shfl(cyi, l, 64) * gr.gw + __shfl(cxi, l, 64);
                const int cn = grid_ok ? gr.count[ci] : GF_GRID_SLOTS;
                if (cn < GF_GRID_SLOTS) {
                    if (lane == 0) { gr.slots[ci * GF_GRID_SLOTS + cn] = (uint32_t)bx | ((uint32_t)by << 16); gr.count[ci] = (uint8_t)(cn + 1); }
                } else grid_ok = false;                          // wave-uniform
            }
            ++nacc;
            if (max_corners > 0 && nacc == max_corners) return true;
            if (lane == l) alive = false;
            else if (alive && use_dist) {
                const float dx = x - bx, dy = y - by;
                if ((double)(dx * dx + dy * dy) < md2) alive = false;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            mm = __ballot(alive);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    return false;
}

// The same pass with the exclusion test turned inside out: every accepted corner rasterises its
// exclusion disc (dx^2 + dy^2 < minDistance^2, one lane per disc row, fire-and-forget LDS atomics) into a
// one-bit-per-pixel image, and a candidate's fate is ONE bit instead of a search through its neighbours.
// xy: the slice's positions (x | y << 16), precomputed by the whole workgroup.  Needs minDistance <= 31
// (a disc row per lane) and w*h bits of LDS; otherwise gf_greedy above runs.
#define GF_BITMAP_WORDS 4096
__device__ __forceinline__ bool gf_greedy_bitmap(const unsigned long long *keys, const uint32_t *xy, int m, unsigned int lo, int w, int h,
                                                 int max_corners, double md2, uint32_t *bitmap, int wp, float *accx, float *accy,
                                                 float2 *out, int &nacc, int lane)
{
    // this lane's disc row: dy = lane - R, half width hw (-1: row not part of the disc)
    int R = 0;
    while ((double)(float)((R + 1) * (R + 1)) < md2) ++R;
    const int dyl = lane - R;
    int hw = -1;
    if (lane <= 2 * R)
        for (int k = 0; k <= R; ++k) if ((double)(float)(k * k + dyl * dyl) < md2) hw = k;
    for (int base = 0; base < m; base += 64) {
        const int c = base + lane;
        bool alive = c < m;
        int xi = 0, yi = 0;
        if (alive) {
            if ((unsigned int)(keys[c] >> 32) <= lo) alive = false;
            const uint32_t p = xy[c];
            xi = (int)(p & 0xffffu); yi = (int)(p >> 16);
            if ((bitmap[yi * wp + (xi >> 5)] >> (xi & 31)) & 1u) alive = false;
        }
        const float x = (float)xi, y = (float)yi;
        unsigned long long mm = __ballot(alive);
        while (mm) {
            const int l = __ffsll((long long)mm) - 1;
            const int bxi = __builtin_amdgcn_readlane(xi, l), byi = __builtin_amdgcn_readlane(yi, l);
            const float bx = (float)bxi, by = (float)byi;
            if (lane == 0) { accx[nacc] = bx; accy[nacc] = by; out[nacc] = make_float2(bx, by); }
            ++nacc;
            if (max_corners > 0 && nacc == max_corners) return true;
            // rasterise the disc for the candidates of later steps
            const int yy = byi + dyl;
            if (hw >= 0 && yy >= 0 && yy < h) {
                const int x1 = max(bxi - hw, 0), x2 = min(bxi + hw, w - 1);
                for (int wd = x1 >> 5; wd <= (x2 >> 5); ++wd) {
                    const int b1 = max(x1 - (wd << 5), 0), b2 = min(x2 - (wd << 5), 31);
                    const uint32_t bits = (b2 - b1 == 31 ? ~0u : ((1u << (b2 - b1 + 1)) - 1u)) << b1;
                    atomicOr(&bitmap[yy * wp + wd], bits);
                }
            }
            // ... and test the rest of this step directly
            if (lane == l) alive = false;
            else if (alive) {
                const float dx = x - bx, dy = y - by;
                if ((double)(dx * dx + dy * dy) < md2) alive = false;
            }
            mm = __ballot(alive);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    return false;
}

__global__ void __launch_bounds__(GS_THREADS)
k_gftt_select2(GfttWork wk, int w, int h_img, int max_corners, double quality, double min_dist, float2 *out_xy, int *out_n,
               int out_stride)
{
    __shared__ unsigned int sHist[GS_BINS];
    __shared__ __attribute__((aligned(16))) unsigned long long sKeys[GS_CHUNK];
    __shared__ float accx[GF_MAX_CORNERS], accy[GF_MAX_CORNERS];
    // exclusion structure of the greedy pass: one bit per pixel (the usual case), or OpenCV's cell grid
    __shared__ uint32_t sExcl[GF_BITMAP_WORDS > GF_GRID_CELLS * GF_GRID_SLOTS ? GF_BITMAP_WORDS : GF_GRID_CELLS * GF_GRID_SLOTS];
    __shared__ uint8_t sGridCnt[GF_GRID_CELLS];
    __shared__ unsigned int sScan[GS_THREADS];
    __shared__ int sCut, sCnt, sDone;

    const int job = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    unsigned int *ctr = wk.counters + (size_t)job * GF_CNT_STRIDE;
    unsigned int n = ctr[1];
    const unsigned int mk = ctr[0];
    __syncthreads();
    if (tid == 0) { ctr[0] = 0; ctr[1] = 0; sDone = 0; }          // the next call finds clean counters
    if (n > (unsigned int)wk.cap) n = wk.cap;
    unsigned long long *gkeys = wk.keys + (size_t)job * wk.cap;
    float2 *out = out_xy + (size_t)job * out_stride;
    const unsigned int lo = gf_threshold_ordered(mk, quality);   // a corner needs value > threshold  <=>  key_hi > lo
    if (n == 0 || mk <= lo) { if (tid == 0) out_n[job] = 0; return; }
    const unsigned int span = mk - lo;                           // key_hi - lo - 1 in [0, span - 1]
    int s = 0;
    while (((span - 1) >> s) >= GS_BINS) ++s;
    const bool use_dist = min_dist >= 1.0;
    const double md2 = min_dist * min_dist;
    unsigned long long ub = (unsigned long long)mk + 1;          // this slice: lo < key_hi < ub
    int nacc = 0;
    GfGrid gr;
    gr.slots = sExcl; gr.count = sGridCnt; gr.cell = 0; gr.gw = gr.gh = 0;
    const int wp = (w + 31) >> 5;
    const bool use_bitmap = use_dist && min_dist <= 31.0 && wp * h_img <= GF_BITMAP_WORDS && w < 65536 && h_img < 65536;
    uint32_t *sXY = sHist;                                       // the histogram is idle while the greedy pass runs
    static_assert(GS_CHUNK <= GS_BINS, "positions of a slice live in the histogram's memory");
    if (use_bitmap) for (int i = tid; i < wp * h_img; i += GS_THREADS) sExcl[i] = 0;
    if (use_dist && !use_bitmap) {
        const int cell = (int)rint(min_dist);                    // cvRound
        const int gw_ = (w + cell - 1) / cell, gh_ = (h_img + cell - 1) / cell;
        if (gw_ * gh_ <= GF_GRID_CELLS) { gr.cell = cell; gr.gw = gw_; gr.gh = gh_; }
    }
    bool grid_ok = true;
    for (int i = tid; i < GF_GRID_CELLS; i += GS_THREADS) sGridCnt[i] = 0;

    long long pt = wall_clock64();
    auto tick = [&](int k) { if (wk.prof && job == 0 && tid == 0) { const long long t = wall_clock64(); wk.prof[k] += t - pt; pt = t; } };
    if (wk.prof && job == 0 && tid == 0) { wk.prof[5] += n; wk.prof[7] += 1; }
    for (;;) {
        for (int i = tid; i < GS_BINS; i += GS_THREADS) sHist[i] = 0;
        if (tid == 0) { sCnt = 0; sCut = -1; }
        __syncthreads();
        tick(8);
        for (unsigned int i = tid; i < n; i += GS_THREADS) {
            const unsigned int kh = (unsigned int)(gkeys[i] >> 32);
            if (kh > lo && kh < ub) atomicAdd(&sHist[(kh - lo - 1) >> s], 1u);
        }
        __syncthreads();
        tick(0);
        // suffix sums over the bins (thread t owns bins 8t .. 8t+7)
        unsigned int own = 0;
#pragma unroll
        for (int b = 0; b < GS_BINS / GS_THREADS; ++b) own += sHist[tid * (GS_BINS / GS_THREADS) + b];
        sScan[tid] = own;
        __syncthreads();
        for (int d = 1; d < GS_THREADS; d <<= 1) {
            const unsigned int add = tid + d < GS_THREADS ? sScan[tid + d] : 0u;
            __syncthreads();
            sScan[tid] += add;
            __syncthreads();
        }
        const unsigned int total = sScan[0];
        const unsigned int above = sScan[tid] - own;
        if (total == 0) break;                                   // nothing left above the threshold
        if (total <= GS_CHUNK) { if (tid == 0) sCut = 0; }
        else if (above <= GS_CHUNK && above + own > GS_CHUNK) {
            unsigned int c = above;
            int cut = 0;
            for (int b = GS_BINS / GS_THREADS - 1; b >= 0; --b) {
                const unsigned int hb = sHist[tid * (GS_BINS / GS_THREADS) + b];
                if (c + hb > GS_CHUNK) { cut = tid * (GS_BINS / GS_THREADS) + b + 1; break; }
                c += hb;
            }
            // the best non-empty bin alone overflows a slice: no progress possible by binning
            sCut = c == 0 ? GS_BINS : cut;
        }
        __syncthreads();
        tick(1);
        const int cut = sCut;
        if (cut >= GS_BINS) {
            // One bin alone holds more than a slice (thousands of equal or nearly equal values, e.g. a
            // synthetic lattice): sort everything in global memory instead.  Rare, correct, slow.
            int n2 = 1;
            while (n2 < (int)n) n2 <<= 1;
            for (int i = n + tid; i < n2; i += GS_THREADS) gkeys[i] = 0ull;
            __threadfence_block();
            __syncthreads();
            bitonic_desc(gkeys, n2, tid, GS_THREADS);
            if (tid < 64) {
                // keys at or above ub were consumed by earlier slices
                int first = 0;
                for (int base = 0; base < (int)n; base += 64) {
                    const bool old = base + lane < (int)n && (gkeys[base + lane] >> 32) >= ub;
                    const int c = __popcll(__ballot(old));
                    first += c;
                    if (c < 64) break;
                }
                gf_greedy(gkeys + first, (int)n - first, lo, w, max_corners, use_dist, md2, accx, accy, gr, grid_ok, out, nacc, lane);
            }
            break;
        }
        for (unsigned int i = tid; i < n; i += GS_THREADS) {
            const unsigned long long key = gkeys[i];
            const unsigned int kh = (unsigned int)(key >> 32);
            if (kh > lo && kh < ub && (int)((kh - lo - 1) >> s) >= cut) sKeys[atomicAdd(&sCnt, 1)] = key;
        }
        __syncthreads();
        tick(2);
        const int m = sCnt;
        if (wk.prof && job == 0 && tid == 0) wk.prof[6] += m;
        int n2 = 8;
        while (n2 < m) n2 <<= 1;
        for (int i = m + tid; i < n2; i += GS_THREADS) sKeys[i] = 0ull;
        __syncthreads();
        bitonic_desc_lds8(sKeys, n2, tid, GS_THREADS);
        if (use_bitmap) {
            for (int i = tid; i < m; i += GS_THREADS) {
                const unsigned int idx = (unsigned int)(sKeys[i] & 0xffffffffull);
                const unsigned int yi = idx / (unsigned int)w;
                sXY[i] = (idx - yi * (unsigned int)w) | (yi << 16);
            }
            __syncthreads();
        }
        tick(3);
        if (tid < 64) {
            const bool done = use_bitmap
                ? gf_greedy_bitmap(sKeys, sXY, m, lo, w, h_img, max_corners, md2, sExcl, wp, accx, accy, out, nacc, lane)
                : gf_greedy(sKeys, m, lo, w, max_corners, use_dist, md2, accx, accy, gr, grid_ok, out, nacc, lane);
            if (lane == 0) sDone = done ? 1 : 0;
        }
        __syncthreads();
        tick(4);
        if (sDone || cut == 0) break;
        ub = (unsigned long long)lo + 1ull + ((unsigned long long)cut << s);
    }
    if (tid == 0) out_n[job] = nacc;
}
