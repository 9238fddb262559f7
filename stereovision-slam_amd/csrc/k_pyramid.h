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
#pragma once
#include "dev_common.h"

struct PyrJob {
    const uint8_t *src;   // level-0 source (device), tight or strided
    int src_stride;
    int slot;
};

// Level 0: copy (optionally 2x nearest decimate) the source into the padded
// level, border included.  One thread writes 4 consecutive padded bytes.
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

// pyrDown level l-1 -> l, border included:
//   dst(x,y) = (sum_{i,j} k_i k_j src(2x+i-2, 2y+j-2) + 128) >> 8, k=[1 4 6 4 1]
// Border pixels of dst are computed at their reflected coordinate, reading the
// source's stored border for the taps that leave the image.
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
