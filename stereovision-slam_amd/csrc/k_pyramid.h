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
