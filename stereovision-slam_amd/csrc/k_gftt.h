// k_gftt.h — Shi-Tomasi corner detection with the reference's feature mask.
// Replaces Frontend::DetectFeatures' mask construction + cv::GFTTDetector
// (OpenCV goodFeaturesToTrack: cornerMinEigenVal -> masked max -> threshold ->
// 3x3 NMS -> sort -> greedy min-distance) at reference src/frontend.cpp:42-51.
// Mirrors oracle/orc_gftt.c; the f32 operation order is the declared one
// (-ffp-contract=off, IEEE sqrt), the 3x3 box sums are exact in f64, so the
// corner list (coordinates, order, count) is bit-exact against the oracle.
//
// Kernels (all batched over jobs in grid.z / grid.y):
//   k_gftt_mask   rasterise the 21x21 exclusion squares of the existing features
//   k_gftt_eig2   fused Sobel -> covariance -> 3x3 box -> min-eigenvalue on DPP wave
//                 shifts + register windows (the reference makes ~10 unfused passes),
//                 plus the masked global maximum (wave reduce + 1 atomic/wave)
//   k_gftt_cand   threshold + 3x3 non-max suppression + mask -> compacted
//                 64-bit keys (ordered value << 32 | pixel index)
//   k_gftt_select one workgroup per job: bitonic sort of the keys (LDS, or
//                 global memory when they do not fit) then the order-dependent
//                 greedy min-distance pass, 64 candidates per step
#pragma once
#include "dev_common.h"

#define GF_CNT_STRIDE 32
struct GfttJob { int slot, rect_ofs, nrect; };

struct GfttWork {            // per-job scratch in HBM
    float *eig;              // [jobs][w*h]
    uint8_t *mask;           // [jobs][w*h]
    unsigned long long *keys;// [jobs][cap]
    unsigned int *counters;  // [jobs][GF_CNT_STRIDE]: 0 = ordered max, 1 = ncand (one 128 B line per job)
    int cap;                 // key capacity per job (power of two >= w*h)
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

__global__ void k_gftt_init(GfttWork wk, int w, int h, int njobs)
{
    // mask = 255, counters = 0
    const int job = blockIdx.y;
    const size_t P = (size_t)w * h;
    uint32_t *m = reinterpret_cast<uint32_t *>(wk.mask + (size_t)job * ((P + 3) & ~(size_t)3));
    const size_t n4 = (P + 3) >> 2;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        m[i] = 0xffffffffu;
    if (blockIdx.x == 0 && threadIdx.x < 4) wk.counters[job * GF_CNT_STRIDE + threadIdx.x] = 0;
}

// one block per rectangle (grid.x = max nrect over jobs, grid.y = job)
__global__ void k_gftt_mask(const GfttJob *jobs, GfttWork wk, const float2 *rect_xy, int w, int h)
{
    const GfttJob jb = jobs[blockIdx.y];
    if ((int)blockIdx.x >= jb.nrect) return;
    const size_t P = (size_t)w * h;
    uint8_t *m = wk.mask + (size_t)blockIdx.y * ((P + 3) & ~(size_t)3);
    const float2 c = rect_xy[jb.rect_ofs + blockIdx.x];
    // cv::rectangle(mask, pt-(10,10), pt+(10,10), 0, FILLED); Point2f->Point = cvRound
    int x1 = (int)rintf(c.x - 10.f), y1 = (int)rintf(c.y - 10.f);
    int x2 = (int)rintf(c.x + 10.f), y2 = (int)rintf(c.y + 10.f);
    x1 = max(x1, 0); y1 = max(y1, 0); x2 = min(x2, w - 1); y2 = min(y2, h - 1);
    const int rw = x2 - x1 + 1, rh = y2 - y1 + 1;
    if (rw <= 0 || rh <= 0) return;
    for (int i = threadIdx.x; i < rw * rh; i += blockDim.x) {
        int yy = i / rw, xx = i - yy * rw;
        m[(size_t)(y1 + yy) * w + x1 + xx] = 0;
    }
}

// ---- min-eigenvalue map without LDS ------------------------------------------------------
// One wave carries 64 image columns (60 outputs + 2 halo columns each side) and walks down
// GE_ROWS + 4 rows.  Horizontal neighbours come from DPP wave shifts (gfx9 wave_shr / wave_shl),
// vertical neighbours from a three-row register window, for the pixels and again for the
// covariance products, so the 3x3 box sum reads registers where k_gftt_eig read 27 LDS words
// per pixel (that kernel is LDS-read bound, ~5x off its VALU time).  Same arithmetic, same
// order of operations, same REFLECT_101 treatment (the covariance of an out-of-image position
// is evaluated at the reflected pixel: with the stored border that is the mirrored
// neighbourhood, i.e. left/right or top/bottom swapped).
#define GE_COLS 60
#define GE_ROWS 32
#define SVS_DPP_WAVE_SHR1 0x138
#define SVS_DPP_WAVE_SHL1 0x130
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
struct GePix { float l, m, r; };
struct GeCov { float xxl, xxm, xxr, xyl, xym, xyr, yyl, yym, yyr; };

__global__ void __launch_bounds__(64)
k_gftt_eig2(const GfttJob *jobs, const uint8_t *pyr, PyrGeom g, GfttWork wk)
{
    const int job = blockIdx.z;
    const GfttJob jb = jobs[job];
    const int w = g.w[0], h = g.h[0], pitch = g.pitch[0];
    const uint8_t *img = lvl_origin(pyr + (size_t)jb.slot * g.slot_bytes, g, 0);
    const size_t P = (size_t)w * h;
    float *eig = wk.eig + (size_t)job * P;
    const uint8_t *mask = wk.mask + (size_t)job * ((P + 3) & ~(size_t)3);
    const int lane = threadIdx.x;
    const int x0 = blockIdx.x * GE_COLS, y0 = blockIdx.y * GE_ROWS;
    const int gx = x0 - 2 + lane;                               // column of this lane's pixel / covariance
    const uint8_t *colp = img + min(gx, w + SVS_BORDER - 1);
    const bool col_out = gx < 0 || gx >= w;
    const bool col_zero = gx > w;                               // beyond the 1-px covariance halo
    const bool store_lane = lane >= 2 && lane < 2 + GE_COLS && gx < w;
    const float s1 = (float)(1.0 / 3060.0);
    const float s2 = (float)(2.0 * (1.0 / 3060.0));
    GePix Pw[3];
    GeCov Cw[3];
    unsigned int best = 0;

    auto step = [&](int i, GePix &Pnew, const GePix &Ptop, const GePix &Pmid, GeCov &Cnew, const GeCov &Ctop, const GeCov &Cmid) {
        // pixel row y0 - 2 + i
        const int ry = min(y0 - 2 + i, h + SVS_BORDER - 1);
        const float p = (float)colp[(ptrdiff_t)ry * pitch];
        float l = dpp_f32<SVS_DPP_WAVE_SHR1>(p), r = dpp_f32<SVS_DPP_WAVE_SHL1>(p);
        if (col_out) { const float t = l; l = r; r = t; }
        Pnew.l = l; Pnew.m = p; Pnew.r = r;
        if (i < 2) return;
        // covariance row cy = y0 - 3 + i from pixel rows (Ptop, Pmid, Pnew)
        const int cy = y0 - 3 + i;
        const bool row_out = cy < 0 || cy >= h;
        const GePix &T = row_out ? Pnew : Ptop, &B = row_out ? Ptop : Pnew;
        const float d0 = T.r - T.l, d1 = Pmid.r - Pmid.l, d2 = B.r - B.l;
        float dx = (d0 + d2) * s1 + d1 * s2;
        const float c0 = (s1 * T.l + s2 * T.m) + s1 * T.r;
        const float c2 = (s1 * B.l + s2 * B.m) + s1 * B.r;
        float dy = c2 - c0;
        if (col_zero || cy > h) { dx = 0.f; dy = 0.f; }
        const float xx = dx * dx, xy = dx * dy, yy = dy * dy;
        Cnew.xxm = xx; Cnew.xym = xy; Cnew.yym = yy;
        Cnew.xxl = dpp_f32<SVS_DPP_WAVE_SHR1>(xx); Cnew.xxr = dpp_f32<SVS_DPP_WAVE_SHL1>(xx);
        Cnew.xyl = dpp_f32<SVS_DPP_WAVE_SHR1>(xy); Cnew.xyr = dpp_f32<SVS_DPP_WAVE_SHL1>(xy);
        Cnew.yyl = dpp_f32<SVS_DPP_WAVE_SHR1>(yy); Cnew.yyr = dpp_f32<SVS_DPP_WAVE_SHL1>(yy);
        if (i < 4) return;
        // output row oy = y0 - 4 + i from covariance rows (Ctop, Cmid, Cnew), row-major f64 sum from zero
        const int oy = y0 - 4 + i;
        double sxx = 0, sxy = 0, syy = 0;
        sxx += (double)Ctop.xxl; sxy += (double)Ctop.xyl; syy += (double)Ctop.yyl;
        sxx += (double)Ctop.xxm; sxy += (double)Ctop.xym; syy += (double)Ctop.yym;
        sxx += (double)Ctop.xxr; sxy += (double)Ctop.xyr; syy += (double)Ctop.yyr;
        sxx += (double)Cmid.xxl; sxy += (double)Cmid.xyl; syy += (double)Cmid.yyl;
        sxx += (double)Cmid.xxm; sxy += (double)Cmid.xym; syy += (double)Cmid.yym;
        sxx += (double)Cmid.xxr; sxy += (double)Cmid.xyr; syy += (double)Cmid.yyr;
        sxx += (double)Cnew.xxl; sxy += (double)Cnew.xyl; syy += (double)Cnew.yyl;
        sxx += (double)Cnew.xxm; sxy += (double)Cnew.xym; syy += (double)Cnew.yym;
        sxx += (double)Cnew.xxr; sxy += (double)Cnew.xyr; syy += (double)Cnew.yyr;
        const float a = (float)sxx * 0.5f, b = (float)sxy, cc = (float)syy * 0.5f;
        const float t = a - cc;
        const float e = (a + cc) - sqrtf(t * t + b * b);
        if (store_lane && oy < h) {
            const size_t pi = (size_t)oy * w + gx;
            eig[pi] = e;
            if (mask[pi]) best = max(best, f32_ordered(e));
        }
    };
    for (int i0 = 0; i0 < GE_ROWS + 4; i0 += 3) {
        // window slot of pixel row i: i % 3; covariance row (i - 2): (i - 2) % 3 = (i + 1) % 3
        step(i0 + 0, Pw[0], Pw[1], Pw[2], Cw[1], Cw[2], Cw[0]);
        step(i0 + 1, Pw[1], Pw[2], Pw[0], Cw[2], Cw[0], Cw[1]);
        step(i0 + 2, Pw[2], Pw[0], Pw[1], Cw[0], Cw[1], Cw[2]);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) best = max(best, (unsigned int)__shfl_xor((int)best, o, 64));
    if (lane == 0 && best) atomicMax(&wk.counters[job * GF_CNT_STRIDE + 0], best);
}

// thread = 4 consecutive pixels of one row; candidates are compacted with one LDS atomic per
// thread and ONE global atomic per block (a per-candidate global atomic serialises ~3000
// same-address operations per image).  The key order is irrelevant: k_gftt_select sorts.
__global__ void __launch_bounds__(256)
k_gftt_cand(GfttWork wk, int w, int h, double quality)
{
    __shared__ unsigned int sCnt, sBase;
    const int job = blockIdx.z;
    const size_t P = (size_t)w * h;
    const float *eig = wk.eig + (size_t)job * P;
    const uint8_t *mask = wk.mask + (size_t)job * ((P + 3) & ~(size_t)3);
    const unsigned int mk = wk.counters[job * GF_CNT_STRIDE + 0];
    const double maxVal = mk ? (double)f32_from_ordered(mk) : 0.0;
    const float thr = (float)(maxVal * quality);
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (tid == 0) sCnt = 0;
    __syncthreads();
    float v[4] = { 0.f, 0.f, 0.f, 0.f };
    unsigned int pm = 0;
    if (y >= 1 && y < h - 1 && x < w) {
        // thresholded 3 x 6 neighbourhood (columns x-1 .. x+4, clamped loads; the clamped
        // positions are only ever neighbours of pixels that are skipped anyway)
        float t[3][6];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int xx = min(max(x - 1 + k, 0), w - 1);
                const float u = eig[(size_t)(y - 1 + j) * w + xx];
                t[j][k] = u > thr ? u : 0.f;
                if (j == 1 && k >= 1 && k <= 4) v[k - 1] = u;
            }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int xx = x + k;
            bool pass = xx >= 1 && xx < w - 1 && v[k] > thr && v[k] != 0.f;
            if (pass) pass = mask[(size_t)y * w + xx] != 0;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int q = 0; q < 3; ++q) pass = pass && !(t[j][k + q] > v[k]);
            pm |= (unsigned int)pass << k;
        }
    }
    unsigned int off = 0;
    const unsigned int np = __popc(pm);
    if (np) off = atomicAdd(&sCnt, np);
    __syncthreads();
    if (tid == 0) sBase = sCnt ? atomicAdd(&wk.counters[job * GF_CNT_STRIDE + 1], sCnt) : 0u;
    __syncthreads();
    if (np) {
        unsigned int slot = sBase + off;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if ((pm >> k) & 1u) {
                if (slot < (unsigned int)wk.cap)
                    wk.keys[(size_t)job * wk.cap + slot] =
                        ((unsigned long long)f32_ordered(v[k]) << 32) | (unsigned int)((size_t)y * w + x + k);
                ++slot;
            }
    }
}

#define GF_SEL_THREADS 1024
#define GF_LDS_KEYS 16384
#define GF_MAX_CORNERS 1024
#define GF_SEL_LDS_BYTES (GF_LDS_KEYS * 8 + GF_MAX_CORNERS * 8)

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
            __syncthreads();
        }
    }
}

__global__ void __launch_bounds__(GF_SEL_THREADS)
k_gftt_select(GfttWork wk, int w, int max_corners, double min_dist, float2 *out_xy, int *out_n,
              int out_stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // all LDS scratch lives in the dynamic region (keeps the base 16-B aligned)
    unsigned long long *skeys = reinterpret_cast<unsigned long long *>(smem);
    float *accx = reinterpret_cast<float *>(smem + (size_t)GF_LDS_KEYS * 8);
    float *accy = accx + GF_MAX_CORNERS;

    const int job = blockIdx.x;
    const int tid = threadIdx.x;
    unsigned int n = wk.counters[job * GF_CNT_STRIDE + 1];
    if (n > (unsigned int)wk.cap) n = wk.cap;
    unsigned long long *gkeys = wk.keys + (size_t)job * wk.cap;
    float2 *out = out_xy + (size_t)job * out_stride;
    if (n == 0) { if (tid == 0) out_n[job] = 0; return; }
    int n2 = 1;
    while (n2 < (int)n) n2 <<= 1;
    unsigned long long *keys;
    if (n2 <= GF_LDS_KEYS) {
        for (int i = tid; i < n2; i += GF_SEL_THREADS) skeys[i] = i < (int)n ? gkeys[i] : 0ull;
        keys = skeys;
    } else {
        for (int i = n + tid; i < n2; i += GF_SEL_THREADS) gkeys[i] = 0ull;
        keys = gkeys;
    }
    __syncthreads();
    bitonic_desc(keys, n2, tid, GF_SEL_THREADS);

    // greedy min-distance selection (order dependent): wave 0 only
    if (tid < 64) {
        const int lane = tid;
        const bool use_dist = min_dist >= 1.0;
        const double md2 = min_dist * min_dist;
        int nacc = 0;
        bool done = false;
        for (int base = 0; base < (int)n && !done; base += 64) {
            const int c = base + lane;
            bool alive = c < (int)n;
            float x = 0.f, y = 0.f;
            if (alive) {
                unsigned int idx = (unsigned int)(keys[c] & 0xffffffffull);
                int yi = idx / w, xi = idx - yi * w;
                x = (float)xi; y = (float)yi;
            }
            if (use_dist) {
                for (int j = 0; j < nacc; ++j) {
                    float dx = x - accx[j], dy = y - accy[j];
                    if ((double)(dx * dx + dy * dy) < md2) alive = false;
                }
            }
            unsigned long long m = __ballot(alive);
            while (m) {
                const int l = __ffsll((long long)m) - 1;
                const float bx = __shfl(x, l, 64), by = __shfl(y, l, 64);
                if (lane == 0) { accx[nacc] = bx; accy[nacc] = by; out[nacc] = make_float2(bx, by); }
                ++nacc;
                if (max_corners > 0 && nacc == max_corners) { done = true; break; }
                if (lane == l) alive = false;
                else if (alive && use_dist) {
                    float dx = x - bx, dy = y - by;
                    if ((double)(dx * dx + dy * dy) < md2) alive = false;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                m = __ballot(alive);
            }
        }
        if (lane == 0) out_n[job] = nacc;
    }
}
