// k_pyramid.h — u8 image pyramid with stored REFLECT_101 border.
// Replaces cv::buildOpticalFlowPyramid / pyrDown as executed inside
// cv::calcOpticalFlowPyrLK (reference src/frontend.cpp:105-109, 353-357) and,
// in the decimating variant, the 1/2 INTER_NEAREST resize of
// Dataset::NextFrame (src/dataset.cpp:126-129).
// Integer arithmetic only -> bit-exact against oracle/orc_image.c.
//
// HBM layout: slot = 4 padded levels back to back; level l is
// (h_l + 2*16) rows of pitch_l bytes (pitch multiple of 64), pixel (0,0) at
// (16,16).  The border is the REFLECT_101 continuation, so LK windows, pyrDown
// taps and the GFTT stencils read it without any index arithmetic.
//
// Fast kernels (levels with w,h >= 18): a thread produces 16 (level 0) or 4
// (pyrDown) interior pixels from aligned dword loads and one aligned vector store,
// and ALSO writes the mirror images of its pixels into the border (x in [1,16] ->
// -x, x in [w-17,w-2] -> 2(w-1)-x, same for rows), so no thread is spent on
// recomputing border pixels.  Generic kernels cover tiny levels.
#pragma once
#include "dev_common.h"
#include "k_rt.h"

struct PyrJob {
    const uint8_t *src;   // level-0 source (device), tight or strided
    int src_stride;
    int slot;
};

// ---------------------------------------------------------------- generic (any size)
template <bool DECIMATE>
__global__ void __launch_bounds__(256)
k_pyr_level0(const PyrJob *jobs, uint8_t *pyr, PyrGeom g, int src_w, int src_h)
{
    const PyrJob jb = jobs[blockIdx.z];
    uint8_t *dst = pyr + (size_t)jb.slot * g.slot_bytes + g.ofs[0];
    const int w = g.w[0], h = g.h[0], pitch = g.pitch[0];
    const int pw4 = (w + 2 * SVS_BORDER + 3) >> 2;
    const int x4 = blockIdx.x * blockDim.x + threadIdx.x;
    const int py = blockIdx.y * blockDim.y + threadIdx.y;
    if (x4 >= pw4 || py >= h + 2 * SVS_BORDER) return;
    const int sy = reflect101(py - SVS_BORDER, h);
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int px = x4 * 4 + k;
        int sx = reflect101(px - SVS_BORDER, w);
        if (px >= w + 2 * SVS_BORDER) sx = 0;
        int rx = sx, ry = sy;
        if (DECIMATE) {
            // cv::resize(..., 0.5, 0.5, INTER_NEAREST): sx = min(2x, src_w-1)
            rx = min(2 * sx, src_w - 1);
            ry = min(2 * sy, src_h - 1);
        }
        out |= (uint32_t)jb.src[(size_t)ry * jb.src_stride + rx] << (8 * k);
    }
    *reinterpret_cast<uint32_t *>(dst + (size_t)py * pitch + x4 * 4) = out;
}

__global__ void __launch_bounds__(256)
k_pyr_down(const PyrJob *jobs, uint8_t *pyr, PyrGeom g, int l)
{
    const PyrJob jb = jobs[blockIdx.z];
    uint8_t *slot = pyr + (size_t)jb.slot * g.slot_bytes;
    const uint8_t *src = lvl_origin((const uint8_t *)slot, g, l - 1);
    uint8_t *dst = slot + g.ofs[l];
    const int sp = g.pitch[l - 1];
    const int w = g.w[l], h = g.h[l], pitch = g.pitch[l];
    const int px = blockIdx.x * blockDim.x + threadIdx.x;
    const int py = blockIdx.y * blockDim.y + threadIdx.y;
    if (px >= w + 2 * SVS_BORDER || py >= h + 2 * SVS_BORDER) return;
    const int x = reflect101(px - SVS_BORDER, w);
    const int y = reflect101(py - SVS_BORDER, h);
    const uint8_t *s = src + (ptrdiff_t)(2 * y - 2) * sp + (2 * x - 2);
    int acc = 0;
    const int kw[5] = { 1, 4, 6, 4, 1 };
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const uint8_t *r = s + (ptrdiff_t)j * sp;
        int row = r[0] + 4 * r[1] + 6 * r[2] + 4 * r[3] + r[4];
        acc += kw[j] * row;
    }
    dst[(size_t)py * pitch + px] = (uint8_t)((acc + 128) >> 8);
}

// ---------------------------------------------------------------- fast path
__device__ __forceinline__ bool pyr_fast_ok(int w, int h) { return w >= 18 && h >= 18; }

// byte k (compile-time) of a 16-byte register block
#define PYR_BYTE(v, k) (uint8_t)(((k) < 4 ? (v).x >> (8 * ((k) & 3)) : (k) < 8 ? (v).y >> (8 * ((k) & 3)) : \
                                  (k) < 12 ? (v).z >> (8 * ((k) & 3)) : (v).w >> (8 * ((k) & 3))) & 0xff)

// store the interior bytes [x0, x0+n) of padded row `row` (held in v, byte k = column x0+k)
// plus their column mirror images; all indexing is compile-time (no scratch)
__device__ __forceinline__ void store16_with_col_mirrors(uint8_t *lvl, int pitch, int w, int row, int x0, int n, uint4 v)
{
    uint8_t *r = lvl + (size_t)(row + SVS_BORDER) * pitch + SVS_BORDER;
    if (n == 16) *reinterpret_cast<uint4 *>(r + x0) = v;
    else {
#pragma unroll
        for (int k = 0; k < 16; ++k) if (k < n) r[x0 + k] = PYR_BYTE(v, k);
    }
    if (x0 <= SVS_BORDER) {
#pragma unroll
        for (int k = 0; k < 16; ++k) { const int x = x0 + k; if (k < n && x >= 1 && x <= SVS_BORDER) r[-x] = PYR_BYTE(v, k); }
    }
    if (x0 + n - 1 >= w - 1 - SVS_BORDER) {
#pragma unroll
        for (int k = 0; k < 16; ++k) { const int x = x0 + k; if (k < n && x >= w - 1 - SVS_BORDER && x <= w - 2) r[2 * (w - 1) - x] = PYR_BYTE(v, k); }
    }
}

// Level 0: thread = 16 interior bytes of one row (optionally 2x nearest decimated).
template <bool DECIMATE>
__global__ void __launch_bounds__(256)
k_pyr_level0_fast(const PyrJob *jobs, uint8_t *pyr, PyrGeom g, int src_w, int src_h)
{
    const PyrJob jb = jobs[blockIdx.z];
    uint8_t *lvl = pyr + (size_t)jb.slot * g.slot_bytes + g.ofs[0];
    const int w = g.w[0], h = g.h[0], pitch = g.pitch[0];
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 16;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x0 >= w || y >= h) return;
    const int n = min(16, w - x0);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (!DECIMATE) {
        const uint8_t *s = jb.src + (size_t)y * jb.src_stride + x0;
        if (n == 16 && ((reinterpret_cast<uintptr_t>(s) & 3) == 0)) {
            const uint32_t *s4 = reinterpret_cast<const uint32_t *>(s);
            v = make_uint4(s4[0], s4[1], s4[2], s4[3]);
        } else {
            uint32_t q[4] = { 0, 0, 0, 0 };
#pragma unroll
            for (int k = 0; k < 16; ++k) if (k < n) q[k >> 2] |= (uint32_t)s[k] << (8 * (k & 3));
            v = make_uint4(q[0], q[1], q[2], q[3]);
        }
    } else {
        // dst(x,y) = src(min(2x, src_w-1), min(2y, src_h-1))
        const int ry = min(2 * y, src_h - 1);
        const uint8_t *s = jb.src + (size_t)ry * jb.src_stride;
        if (n == 16 && 2 * (x0 + 15) <= src_w - 1 && ((reinterpret_cast<uintptr_t>(s + 2 * x0) & 3) == 0)) {
            const uint32_t *s4 = reinterpret_cast<const uint32_t *>(s + 2 * x0);
            uint32_t q[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t a = s4[2 * i], b = s4[2 * i + 1];   // bytes 0,2 of each dword are the even columns
                q[i] = (a & 0xff) | ((a >> 8) & 0xff00) | ((b & 0xff) << 16) | ((b << 8) & 0xff000000u);
            }
            v = make_uint4(q[0], q[1], q[2], q[3]);
        } else {
            uint32_t q[4] = { 0, 0, 0, 0 };
#pragma unroll
            for (int k = 0; k < 16; ++k) if (k < n) q[k >> 2] |= (uint32_t)s[min(2 * (x0 + k), src_w - 1)] << (8 * (k & 3));
            v = make_uint4(q[0], q[1], q[2], q[3]);
        }
    }
    store16_with_col_mirrors(lvl, pitch, w, y, x0, n, v);
    if (y >= 1 && y <= SVS_BORDER) store16_with_col_mirrors(lvl, pitch, w, -y, x0, n, v);
    if (y >= h - 1 - SVS_BORDER && y <= h - 2) store16_with_col_mirrors(lvl, pitch, w, 2 * (h - 1) - y, x0, n, v);
}

__device__ __forceinline__ void store4_with_col_mirrors(uint8_t *lvl, int pitch, int w, int row, int x0, int n, uint32_t v)
{
    uint8_t *r = lvl + (size_t)(row + SVS_BORDER) * pitch + SVS_BORDER;
    if (n == 4) *reinterpret_cast<uint32_t *>(r + x0) = v;
    else {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < n) r[x0 + k] = (uint8_t)((v >> (8 * k)) & 0xff);
    }
    if (x0 <= SVS_BORDER) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int x = x0 + k; if (k < n && x >= 1 && x <= SVS_BORDER) r[-x] = (uint8_t)((v >> (8 * k)) & 0xff); }
    }
    if (x0 + n - 1 >= w - 1 - SVS_BORDER) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int x = x0 + k; if (k < n && x >= w - 1 - SVS_BORDER && x <= w - 2) r[2 * (w - 1) - x] = (uint8_t)((v >> (8 * k)) & 0xff); }
    }
}

// pyrDown: thread = 4 interior output pixels of one row; 5 source rows x 16 bytes by aligned
// dword loads (the stored border of the source supplies every out-of-image tap).
__global__ void __launch_bounds__(256)
k_pyr_down_fast(const PyrJob *jobs, uint8_t *pyr, PyrGeom g, int l)
{
    const PyrJob jb = jobs[blockIdx.z];
    uint8_t *slot = pyr + (size_t)jb.slot * g.slot_bytes;
    const uint8_t *src = lvl_origin((const uint8_t *)slot, g, l - 1);
    uint8_t *lvl = slot + g.ofs[l];
    const int sp = g.pitch[l - 1];
    const int w = g.w[l], h = g.h[l], pitch = g.pitch[l];
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x0 >= w || y >= h) return;
    const int n = min(4, w - x0);
    int acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int kwj = (j == 0 || j == 4) ? 1 : (j == 2 ? 6 : 4);
        // source columns 2*x0-4 .. 2*x0+11 of row 2y-2+j (16-byte window, 4-byte aligned)
        const uint32_t *r4 = reinterpret_cast<const uint32_t *>(src + (ptrdiff_t)(2 * y - 2 + j) * sp + (2 * x0 - 4));
        const uint32_t v0 = r4[0], v1 = r4[1], v2 = r4[2], v3 = r4[3];
#define PB(v, k) (int)(((v) >> (8 * (k))) & 0xff)
        // window column c (0..15): c<4 -> v0, <8 -> v1, <12 -> v2, else v3; output o uses c = 2o+2 .. 2o+6
        acc0 += kwj * (PB(v0, 2) + 4 * PB(v0, 3) + 6 * PB(v1, 0) + 4 * PB(v1, 1) + PB(v1, 2));
        acc1 += kwj * (PB(v1, 0) + 4 * PB(v1, 1) + 6 * PB(v1, 2) + 4 * PB(v1, 3) + PB(v2, 0));
        acc2 += kwj * (PB(v1, 2) + 4 * PB(v1, 3) + 6 * PB(v2, 0) + 4 * PB(v2, 1) + PB(v2, 2));
        acc3 += kwj * (PB(v2, 0) + 4 * PB(v2, 1) + 6 * PB(v2, 2) + 4 * PB(v2, 3) + PB(v3, 0));
#undef PB
    }
    const uint32_t v = (uint32_t)((acc0 + 128) >> 8) | ((uint32_t)((acc1 + 128) >> 8) << 8) |
                       ((uint32_t)((acc2 + 128) >> 8) << 16) | ((uint32_t)((acc3 + 128) >> 8) << 24);
    store4_with_col_mirrors(lvl, pitch, w, y, x0, n, v);
    if (y >= 1 && y <= SVS_BORDER) store4_with_col_mirrors(lvl, pitch, w, -y, x0, n, v);
    if (y >= h - 1 - SVS_BORDER && y <= h - 2) store4_with_col_mirrors(lvl, pitch, w, 2 * (h - 1) - y, x0, n, v);
}

// ---------------------------------------------------------------- all levels in one launch
// One workgroup builds a horizontal strip of EVERY level of one image: the strip's level-0 rows (plus
// the halo the coarser levels need: 2 rows per side per level, compounded) are read from the source
// once, staged in LDS, and each pyrDown level is computed from the level below in LDS — no level is
// re-read from HBM, one launch instead of four dependent ones.
// A row lives in LDS exactly as it lives in the slot: 16 border bytes, the w pixels, 16 border bytes
// (REFLECT_101, filled in LDS).  So (a) the 5-tap window of 4 outputs is four aligned dwords and
// needs no index arithmetic, and (b) a finished row — and its mirror image among the slot's border
// rows — goes out as plain 16-byte chunks; no byte-granular global store exists in this kernel.
// The 5x5 binomial runs on v_dot4_u32_u8: one output pixel of one source row is two dot products
// (weights (1,4,6,4)*k_row on four bytes, k_row on the fifth), accumulated across the five rows.
// Strips overlap only in what they READ.  Host side (pyr_fused_plan): strip height in coarsest-level
// rows, LDS offsets and pitches.
#ifndef PF_THREADS
// Round 6: 512 (was 1024).  With the fill no longer bound by the L1's line rate (PF_UA_LOADS below) the smaller workgroup is
// the faster one alone (150 against 159 us per launch of the headline configuration) and in the line (595-602 k against
// 591 k frames/s); 256 threads: 185 us.  profiles/r6_ab_pyramid_unaligned_loads.txt
#define PF_THREADS 512
#endif
#define PF_PAD 16                   // == SVS_BORDER: LDS rows and slot rows share one layout
#ifndef PF_MAXR
#define PF_MAXR 3                   // 16-byte source loads a thread keeps in flight
#endif
struct PyrFusedPlan {
    int rows_top;                   // coarsest-level rows per strip
    int nstrips;
    int pitch[SVS_LEVELS];          // LDS row pitch per level (multiple of 16, >= w + 32)
    int ofs[SVS_LEVELS];            // LDS byte offset per level
    int cap[SVS_LEVELS];            // LDS rows reserved per level
    int lds_bytes;
    // floor(t / d) == (t * magic) >> 32 for the task indices below (t < 2^20, d < 2^11): d = 16-byte chunks of a
    // level-0 row, 16-byte chunks of a padded row, 4-pixel groups of a row
    uint32_t magic_fill, magic_store[SVS_LEVELS], magic_quad[SVS_LEVELS];
    long long *prof;                // development: phase ticks of workgroup (1, 0) (SVSLAM_PYR_PROF), else null
};
__host__ __device__ inline uint32_t pyr_magic(int d) { return (uint32_t)(0xFFFFFFFFull / (unsigned)d) + 1u; }
__device__ __forceinline__ int pyr_div(int t, uint32_t magic) { return (int)__umulhi((uint32_t)t, magic); }

// real rows [lo, hi] of every level that strip [c0, c1) of the coarsest level has to hold
__host__ __device__ inline void pyr_strip_rows(const PyrGeom &g, int c0, int c1, int *own_lo, int *own_hi, int *need_lo, int *need_hi)
{
    const int L = g.nlevels - 1;
    for (int l = 0; l <= L; ++l) {
        own_lo[l] = c0 << (L - l);
        const int e = c1 << (L - l);
        own_hi[l] = e < g.h[l] ? e : g.h[l];                    // exclusive
    }
    need_lo[L] = c0; need_hi[L] = c1 - 1;
    for (int l = L - 1; l >= 0; --l) {
        int lo = 2 * need_lo[l + 1] - 2, hi = 2 * need_hi[l + 1] + 2;
        const int h = g.h[l];
        // REFLECT_101 folds rows -2,-1 onto 2,1 and h,h+1 onto h-2,h-3: keep the images inside the range
        if (lo < 0) { if (hi < 2) hi = 2; lo = 0; }
        if (hi > h - 1) { if (lo > h - 3) lo = h - 3; hi = h - 1; }
        if (own_lo[l] < lo) lo = own_lo[l];
        if (own_hi[l] - 1 > hi) hi = own_hi[l] - 1;
        if (lo < 0) lo = 0;
        need_lo[l] = lo; need_hi[l] = hi;
    }
}

typedef const __attribute__((address_space(1))) uint32_t *pyr_gptr32;
struct PyrRaw16 { uint32_t d[5]; int m; };
// issue: five aligned dword loads covering 16 source bytes at p (any alignment), of which the first n are
// needed (branch-free: a dword past the needed ones re-reads the last needed one)
__device__ __forceinline__ PyrRaw16 pyr_load16_issue(const uint8_t *p, int n)
{
    PyrRaw16 r;
    r.m = (int)(reinterpret_cast<uintptr_t>(p) & 3);
    // the images live in device (global) memory: say so, or the loads become flat_load
    pyr_gptr32 q = (pyr_gptr32)(reinterpret_cast<uintptr_t>(p) - (uintptr_t)r.m);
    const int last = ((n + r.m + 3) >> 2) - 1;                  // last dword that holds a needed byte (<= 4)
    r.d[0] = q[0]; r.d[1] = q[min(1, last)]; r.d[2] = q[min(2, last)]; r.d[3] = q[min(3, last)]; r.d[4] = q[min(4, last)];
    return r;
}
// use: realign to the byte the caller asked for
__device__ __forceinline__ uint4 pyr_load16_finish(const PyrRaw16 &r)
{
    return make_uint4(__builtin_amdgcn_alignbyte(r.d[1], r.d[0], r.m), __builtin_amdgcn_alignbyte(r.d[2], r.d[1], r.m),
                      __builtin_amdgcn_alignbyte(r.d[3], r.d[2], r.m), __builtin_amdgcn_alignbyte(r.d[4], r.d[3], r.m));
}
// Round 6: the source fill by UNALIGNED 16-byte loads.  A 1241-byte row pitch puts every row at another alignment, and the
// five aligned dwords above cost five wave-wide loads whose lanes sit 16 (32 when decimating) bytes apart — each touches 8 (16)
// cache lines for 256 bytes of payload, and the fill was bound by the L1's line rate, not by HBM.  gfx950 serves a misaligned
// global_load_dwordx4 in hardware (the code object runs in unaligned access mode, which is why the compiler emits it for an
// align-1 vector type); consecutive lanes then read consecutive 16-byte (32-byte) pieces of the row: one (two) loads per task.
// The window may run past the bytes the task needs, into the next row; only a window that would leave the image's LAST row is
// not issued (pyr_ua_unsafe: the task is gathered bytewise instead — at most one task per image).
#ifndef PF_UA_LOADS
#define PF_UA_LOADS 1
#endif
typedef uint32_t pyr_u4v __attribute__((ext_vector_type(4)));
typedef pyr_u4v pyr_u4v_ua __attribute__((aligned(1)));
typedef const __attribute__((address_space(1))) pyr_u4v_ua *pyr_gptr128;
__device__ __forceinline__ pyr_u4v pyr_load16_ua(const uint8_t *p) { return *(pyr_gptr128)(reinterpret_cast<uintptr_t>(p)); }
// bytes 0,2 of two consecutive dwords -> one dword (2:1 column decimation)
__device__ __forceinline__ uint32_t pyr_even_bytes(uint32_t a, uint32_t b)
{
    return __builtin_amdgcn_perm(b, a, 0x06040200u);
}
__device__ __forceinline__ uint32_t pyr_bswap(uint32_t v) { return __builtin_amdgcn_perm(0u, v, 0x00010203u); }

// REFLECT_101 continuation of one LDS row (pixel 0 at r): 16 bytes left of 0 (mirror of 1..16), 16 right of w-1
__device__ __forceinline__ void pyr_fill_row_border(uint8_t *r, int w, int side)
{
    if (side == 0) {
        // columns -16..-1 = columns 16..1: bytes 1..16 sit one byte off alignment -> 5 dwords, realign, reverse
        const uint32_t *q = reinterpret_cast<const uint32_t *>(r);
        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
        const uint32_t a = __builtin_amdgcn_alignbyte(d1, d0, 1), b = __builtin_amdgcn_alignbyte(d2, d1, 1),
                       c = __builtin_amdgcn_alignbyte(d3, d2, 1), d = __builtin_amdgcn_alignbyte(d4, d3, 1);   // columns 1..4, 5..8, 9..12, 13..16
        uint32_t *o = reinterpret_cast<uint32_t *>(r - 16);
        o[0] = pyr_bswap(d); o[1] = pyr_bswap(c); o[2] = pyr_bswap(b); o[3] = pyr_bswap(a);
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) r[w + k] = r[w - 2 - k];
    }
}

// rows [y0, y1) of a level: LDS -> slot, 16-byte chunks of the whole padded row, plus the copies that
// make up the slot's top / bottom border rows (row -y is row y for y in 1..16, row h-1+k is row h-1-k)
__device__ __forceinline__ void pyr_store_rows(uint8_t *lvl, int pitch, int w, int h, const uint8_t *S, int sp, int lds_row0,
                                               int y0, int y1, int tid, uint32_t magic)
{
    const int chunks = (w + 2 * SVS_BORDER + 15) >> 4;
    const int ntask = (y1 - y0) * chunks;
    for (int t = tid; t < ntask; t += PF_THREADS) {
        const int ro = pyr_div(t, magic), ck = t - ro * chunks;
        const int y = y0 + ro;
        const uint4 v = *reinterpret_cast<const uint4 *>(S + (y - lds_row0) * sp + (ck << 4));
        *reinterpret_cast<uint4 *>(lvl + (size_t)(y + SVS_BORDER) * pitch + (ck << 4)) = v;
        if (y >= 1 && y <= SVS_BORDER) *reinterpret_cast<uint4 *>(lvl + (size_t)(SVS_BORDER - y) * pitch + (ck << 4)) = v;
        if (y >= h - 1 - SVS_BORDER && y <= h - 2) *reinterpret_cast<uint4 *>(lvl + (size_t)(2 * (h - 1) - y + SVS_BORDER) * pitch + (ck << 4)) = v;
    }
}

template <bool DECIMATE>
__global__ void __launch_bounds__(PF_THREADS)
k_pyr_fused(const PyrJob *jobs, int njobs, uint8_t *pyr, PyrGeom g, int src_w, int src_h, PyrFusedPlan pl, RtGatherArgs ga)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t pf_lds[];
    // Workgroup b runs on XCD b % 8 (observed dispatch order; only speed depends on it): keep the strips of
    // one image on one XCD, so the rows two neighbouring strips both read come out of that XCD's L2.
    const int bid = blockIdx.x;
    // workgroups past the pyramid's own: the feature gather of resident tracking (k_rt.h) rides on this launch — it reads
    // nothing the pyramid writes, and as a launch of its own it would stand between the pyramid and LK
    const int pyr_blocks = ((njobs + 7) >> 3) * 8 * pl.nstrips;
    if (bid >= pyr_blocks) {
        const int gb = bid - pyr_blocks, gj = gb / ga.chunks;
        if (gj < ga.njobs) {
            const RtJob rj = ga.jobs[gj];                 // a copy: the array may be pinned host memory
            const int i = (gb - gj * ga.chunks) * PF_THREADS + (int)threadIdx.x;
            if (i < rj.npts) rt_gather_point(rj, ga.rs, ga.cam, i, ga.prev_xy, ga.next_xy, ga.has_mp, ga.xyz);
        }
        return;
    }
    const int job = (bid / (8 * pl.nstrips)) * 8 + (bid & 7);
    const int strip = (bid >> 3) % pl.nstrips;
    if (job >= njobs) return;
    const PyrJob jb = jobs[job];
    uint8_t *slot = pyr + (size_t)jb.slot * g.slot_bytes;
    const int tid = threadIdx.x;
    const int L = g.nlevels - 1;
    const int c0 = strip * pl.rows_top, c1 = min(c0 + pl.rows_top, g.h[L]);
    int own_lo[SVS_LEVELS], own_hi[SVS_LEVELS], need_lo[SVS_LEVELS], need_hi[SVS_LEVELS];
    pyr_strip_rows(g, c0, c1, own_lo, own_hi, need_lo, need_hi);
    long long pt = wall_clock64();
    auto tick = [&](int k) { if (pl.prof && strip == 1 && job == 0 && tid == 0) { const long long t = wall_clock64(); pl.prof[k] += t - pt; pt = t; } };

    // ---- level 0: source -> LDS
    {
        const int w = g.w[0];
        uint8_t *S = pf_lds + pl.ofs[0];
        const int sp = pl.pitch[0];
        const int chunks = (w + 15) >> 4;
        const int nrows = need_hi[0] - need_lo[0] + 1;
        // all of a thread's source loads are issued before the first use (PF_MAXR x 16 bytes in flight
        // per thread): the strip's fill is one memory latency, not one per 16 bytes
        const int ntasks = nrows * chunks;
#if PF_UA_LOADS
        for (int tb = 0; tb < ntasks; tb += PF_MAXR * PF_THREADS) {
            pyr_u4v ra[PF_MAXR], rb[DECIMATE ? PF_MAXR : 1];
            // bytes of the source row a task's window covers: 16, or 32 when decimating and more than 16 are needed
#pragma unroll
            for (int k = 0; k < PF_MAXR; ++k) {
                const int t = min(tb + k * PF_THREADS + tid, ntasks - 1);      // surplus threads repeat the last task
                const int rr = pyr_div(t, pl.magic_fill), x0 = (t - rr * chunks) << 4;
                const int y = need_lo[0] + rr;
                if (!DECIMATE) {
                    const bool unsafe = y == src_h - 1 && x0 + 16 > src_w;
                    ra[k] = pyr_load16_ua(unsafe ? jb.src : jb.src + (size_t)y * jb.src_stride + x0);
                } else {
                    const int sy = min(2 * y, src_h - 1);
                    const int nb = min(2 * min(16, w - x0) - 1, src_w - 2 * x0);          // source bytes needed
                    const bool unsafe = sy == src_h - 1 && 2 * x0 + (nb > 16 ? 32 : 16) > src_w;
                    const uint8_t *s = unsafe ? jb.src : jb.src + (size_t)sy * jb.src_stride + 2 * x0;
                    ra[k] = pyr_load16_ua(s);
                    rb[k] = pyr_load16_ua(nb > 16 && !unsafe ? s + 16 : s);
                }
            }
#pragma unroll
            for (int k = 0; k < PF_MAXR; ++k) {
                const int t = tb + k * PF_THREADS + tid;
                if (t < ntasks) {
                    const int rr = pyr_div(t, pl.magic_fill), x0 = (t - rr * chunks) << 4;
                    const int y = need_lo[0] + rr, n = min(16, w - x0);
                    uint4 v;
                    if (!DECIMATE) {
                        v = make_uint4(ra[k].x, ra[k].y, ra[k].z, ra[k].w);
                        if (y == src_h - 1 && x0 + 16 > src_w) {
                            const uint8_t *s = jb.src + (size_t)y * jb.src_stride + x0;
                            uint32_t q[4] = { 0, 0, 0, 0 };
                            for (int i = 0; i < n; ++i) q[i >> 2] |= (uint32_t)s[i] << (8 * (i & 3));
                            v = make_uint4(q[0], q[1], q[2], q[3]);
                        }
                    } else {
                        v = make_uint4(pyr_even_bytes(ra[k].x, ra[k].y), pyr_even_bytes(ra[k].z, ra[k].w),
                                       pyr_even_bytes(rb[k].x, rb[k].y), pyr_even_bytes(rb[k].z, rb[k].w));
                        const int sy = min(2 * y, src_h - 1);
                        const int nb = min(2 * n - 1, src_w - 2 * x0);
                        if (sy == src_h - 1 && 2 * x0 + (nb > 16 ? 32 : 16) > src_w) {
                            const uint8_t *s = jb.src + (size_t)sy * jb.src_stride + 2 * x0;
                            uint32_t q[4] = { 0, 0, 0, 0 };
                            for (int i = 0; i < n; ++i) q[i >> 2] |= (uint32_t)s[min(2 * i, nb - 1)] << (8 * (i & 3));
                            v = make_uint4(q[0], q[1], q[2], q[3]);
                        }
                    }
                    *reinterpret_cast<uint4 *>(S + rr * sp + PF_PAD + x0) = v;
                }
            }
        }
#else
        for (int tb = 0; tb < ntasks; tb += PF_MAXR * PF_THREADS) {
            PyrRaw16 ra[PF_MAXR], rb[DECIMATE ? PF_MAXR : 1];
#pragma unroll
            for (int k = 0; k < PF_MAXR; ++k) {
                const int t = min(tb + k * PF_THREADS + tid, ntasks - 1);      // surplus threads repeat the last task
                const int rr = pyr_div(t, pl.magic_fill), x0 = (t - rr * chunks) << 4;
                const int y = need_lo[0] + rr;
                const int n = min(16, w - x0);
                if (!DECIMATE) ra[k] = pyr_load16_issue(jb.src + (size_t)y * jb.src_stride + x0, n);
                else {
                    // dst(x,y) = src(2x, 2y)  (cv::resize INTER_NEAREST at 1/2; the clamps of the general
                    // formula never bind for the sizes svslam_set_source_size accepts)
                    const uint8_t *s = jb.src + (size_t)min(2 * y, src_h - 1) * jb.src_stride + 2 * x0;
                    const int nb = min(2 * n - 1, src_w - 2 * x0);          // source bytes needed
                    ra[k] = pyr_load16_issue(s, min(nb, 16));
                    rb[k] = pyr_load16_issue(nb > 16 ? s + 16 : s, nb > 16 ? nb - 16 : 1);
                }
            }
#pragma unroll
            for (int k = 0; k < PF_MAXR; ++k) {
                const int t = tb + k * PF_THREADS + tid;
                if (t < ntasks) {
                    const int rr = pyr_div(t, pl.magic_fill), x0 = (t - rr * chunks) << 4;
                    uint4 v = pyr_load16_finish(ra[k]);
                    if (DECIMATE) {
                        const uint4 b = pyr_load16_finish(rb[k]);
                        v = make_uint4(pyr_even_bytes(v.x, v.y), pyr_even_bytes(v.z, v.w), pyr_even_bytes(b.x, b.y), pyr_even_bytes(b.z, b.w));
                    }
                    *reinterpret_cast<uint4 *>(S + rr * sp + PF_PAD + x0) = v;
                }
            }
        }
#endif
        __syncthreads();
        tick(0);
    }
    // ---- every level: border in LDS, rows out to the slot, next level from LDS
    for (int l = 0; l <= L; ++l) {
        const int w = g.w[l], h = g.h[l];
        uint8_t *S = pf_lds + pl.ofs[l];
        const int sp = pl.pitch[l];
        const int nrows = need_hi[l] - need_lo[l] + 1;
        for (int t = tid; t < 2 * nrows; t += PF_THREADS) pyr_fill_row_border(S + (t >> 1) * sp + PF_PAD, w, t & 1);
        __syncthreads();
        pyr_store_rows(slot + g.ofs[l], g.pitch[l], w, h, S, sp, need_lo[l], own_lo[l], own_hi[l], tid, pl.magic_store[l]);
        if (l == L) break;
        // pyrDown: level l+1 rows need_lo..need_hi from this level's LDS rows
        const int wn = g.w[l + 1];
        uint8_t *Sn = pf_lds + pl.ofs[l + 1];
        const int spn = pl.pitch[l + 1];
        const int quads = (wn + 3) >> 2;
        const int nr = need_hi[l + 1] - need_lo[l + 1] + 1;
        for (int t = tid; t < nr * quads; t += PF_THREADS) {
            const int rr = pyr_div(t, pl.magic_quad[l + 1]), x0 = (t - rr * quads) << 2;
            const int y = need_lo[l + 1] + rr;
            uint32_t acc0 = 128, acc1 = 128, acc2 = 128, acc3 = 128;    // + 128: rounding of the >> 8
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const uint32_t kwj = (j == 0 || j == 4) ? 1u : (j == 2 ? 6u : 4u);
                const uint32_t W4 = 0x04060401u * kwj;                   // (1,4,6,4) * k_row, each <= 36
                const int sy = reflect101(2 * y - 2 + j, h) - need_lo[l];
                // source columns 2*x0-4 .. 2*x0+11 (16-byte window, 4-byte aligned): output o taps 2o-2 .. 2o+2
                const uint32_t *r4 = reinterpret_cast<const uint32_t *>(S + sy * sp + PF_PAD + 2 * x0 - 4);
                const uint32_t v0 = r4[0], v1 = r4[1], v2 = r4[2], v3 = r4[3];
                const uint32_t a02 = __builtin_amdgcn_alignbyte(v1, v0, 2), a12 = __builtin_amdgcn_alignbyte(v2, v1, 2);
                acc0 = __builtin_amdgcn_udot4(a02, W4, acc0, false); acc0 = __builtin_amdgcn_udot4(v1, kwj << 16, acc0, false);
                acc1 = __builtin_amdgcn_udot4(v1, W4, acc1, false);  acc1 = __builtin_amdgcn_udot4(v2, kwj, acc1, false);
                acc2 = __builtin_amdgcn_udot4(a12, W4, acc2, false); acc2 = __builtin_amdgcn_udot4(v2, kwj << 16, acc2, false);
                acc3 = __builtin_amdgcn_udot4(v2, W4, acc3, false);  acc3 = __builtin_amdgcn_udot4(v3, kwj, acc3, false);
            }
            const uint32_t v = (acc0 >> 8) | ((acc1 >> 8) << 8) | ((acc2 >> 8) << 16) | ((acc3 >> 8) << 24);
            *reinterpret_cast<uint32_t *>(Sn + rr * spn + PF_PAD + x0) = v;
        }
        __syncthreads();
        tick(1 + l);
    }
    if (pl.prof && strip == 1 && job == 0 && tid == 0) { tick(5); pl.prof[7] += 1; }
}

// strip height and LDS layout for a geometry; returns false when the fused kernel does not apply
// (a level too small for a 16-pixel mirrored border, or no strip fits the LDS budget)
inline bool pyr_fused_plan(const PyrGeom &g, int lds_budget, PyrFusedPlan &pl)
{
    if (g.nlevels < 2) return false;
    for (int l = 0; l < g.nlevels; ++l) if (g.w[l] < 18 || g.h[l] < 18) return false;
    const int L = g.nlevels - 1;
    for (int l = 0; l < SVS_LEVELS; ++l) { pl.pitch[l] = 0; pl.ofs[l] = 0; pl.cap[l] = 0; }
    for (int l = 0; l <= L; ++l) {
        pl.pitch[l] = (g.w[l] + 2 * PF_PAD + 15) & ~15;
        pl.magic_store[l] = pyr_magic((g.w[l] + 2 * SVS_BORDER + 15) >> 4);
        pl.magic_quad[l] = pyr_magic((g.w[l] + 3) >> 2);
    }
    pl.magic_fill = pyr_magic((g.w[0] + 15) >> 4);
    for (int rt = g.h[L]; rt >= 1; --rt) {
        int cap[SVS_LEVELS] = { 0, 0, 0, 0 };
        const int ns = (g.h[L] + rt - 1) / rt;
        for (int s = 0; s < ns; ++s) {
            int ol[SVS_LEVELS], oh[SVS_LEVELS], nl[SVS_LEVELS], nh[SVS_LEVELS];
            const int c0 = s * rt, c1 = c0 + rt < g.h[L] ? c0 + rt : g.h[L];
            pyr_strip_rows(g, c0, c1, ol, oh, nl, nh);
            for (int l = 0; l <= L; ++l) if (nh[l] - nl[l] + 1 > cap[l]) cap[l] = nh[l] - nl[l] + 1;
        }
        int off = 0;
        for (int l = 0; l <= L; ++l) { pl.ofs[l] = off; pl.cap[l] = cap[l]; off += cap[l] * pl.pitch[l]; off = (off + 15) & ~15; }
        if (off + 64 <= lds_budget) { pl.rows_top = rt; pl.nstrips = ns; pl.lds_bytes = off + 64; pl.prof = nullptr; return true; }
    }
    return false;
}
