/*
 * synth_scene.h — procedural stereo scene used as the synthetic KITTI-00-shaped
 * input stream (there is no KITTI data on either box).
 *
 * A closed "tunnel" (ground, two walls, ceiling) carrying a world-attached,
 * footprint-antialiased multi-scale tile texture is ray-cast per pixel, so the
 * left/right images and consecutive frames are photometrically consistent views
 * of one rigid 3-D scene with known camera poses (ground truth for ATE).
 * The function is shared by the HIP generator kernel (synth.hip) and the CPU
 * generator (synth_cpu.c); it is input generation, not part of the hot path.
 *
 * The image is rendered directly at the pixels the reference's 1/2 nearest
 * decimation keeps (src/dataset.cpp:126-129: dst(x,y) = src(2x,2y)), i.e. with
 * the halved intrinsics of src/dataset.cpp:73.
 */
#ifndef SVS_SYNTH_SCENE_H
#define SVS_SYNTH_SCENE_H
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define SVS_HD __host__ __device__ __forceinline__
#else
#define SVS_HD static inline
#endif

typedef struct svs_synth_view {
    float fx, fy, cx, cy; /* intrinsics of the rendered image                 */
    float R[9];           /* camera -> world rotation, row-major              */
    double C[3];          /* camera centre in world (double: z grows to km)   */
    uint32_t seed;        /* scene seed                                       */
    uint32_t noise_seed;  /* per-image sensor-noise seed                      */
    float scale;          /* pixel step in rendered-image units (1 = as is)   */
} svs_synth_view;

SVS_HD uint32_t svs_hash3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t h = a * 0x9E3779B1u ^ (b * 0x85EBCA77u) ^ (c * 0xC2B2AE3Du);
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return h;
}

SVS_HD float svs_clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

SVS_HD float svs_tile(double s, double t, float cell, float fp, uint32_t seed)
{
    /* lattice values blended with an edge whose width follows the pixel
     * footprint: sharp tiles up close, smooth value noise far away */
    double u = s / (double)cell, v = t / (double)cell;
    double fu = floor(u), fv = floor(v);
    int iu = (int)fu, iv = (int)fv;
    float a = (float)(u - fu), b = (float)(v - fv);
    float w = svs_clampf(fp / cell, 0.06f, 0.5f);
    float ba = svs_clampf((a - 0.5f) / (2.f * w) + 0.5f, 0.f, 1.f);
    float bb = svs_clampf((b - 0.5f) / (2.f * w) + 0.5f, 0.f, 1.f);
    ba = ba * ba * (3.f - 2.f * ba);
    bb = bb * bb * (3.f - 2.f * bb);
    float v00 = (float)(svs_hash3((uint32_t)iu, (uint32_t)iv, seed) & 0xFFFF) * (1.f / 65535.f);
    float v10 = (float)(svs_hash3((uint32_t)(iu + 1), (uint32_t)iv, seed) & 0xFFFF) * (1.f / 65535.f);
    float v01 = (float)(svs_hash3((uint32_t)iu, (uint32_t)(iv + 1), seed) & 0xFFFF) * (1.f / 65535.f);
    float v11 = (float)(svs_hash3((uint32_t)(iu + 1), (uint32_t)(iv + 1), seed) & 0xFFFF) * (1.f / 65535.f);
    float top = v00 + (v10 - v00) * ba;
    float bot = v01 + (v11 - v01) * ba;
    return top + (bot - top) * bb;
}

#define SVS_TUNNEL_XL (-6.0f)
#define SVS_TUNNEL_XR (6.5f)
#define SVS_TUNNEL_YT (-4.0f)
#define SVS_TUNNEL_YB (1.65f)

SVS_HD uint8_t svs_synth_pixel(const svs_synth_view *vw, int px, int py)
{
    float xc = ((float)px - vw->cx) / vw->fx;
    float yc = ((float)py - vw->cy) / vw->fy;
    const float *R = vw->R;
    float dx = R[0] * xc + R[1] * yc + R[2];
    float dy = R[3] * xc + R[4] * yc + R[5];
    float dz = R[6] * xc + R[7] * yc + R[8];
    float Cx = (float)vw->C[0], Cy = (float)vw->C[1];
    float tbest = 1e9f, dn = 1.f;
    int surf = 0;
    if (dx > 1e-6f) { float t = (SVS_TUNNEL_XR - Cx) / dx; if (t < tbest) { tbest = t; surf = 1; dn = dx; } }
    if (dx < -1e-6f) { float t = (SVS_TUNNEL_XL - Cx) / dx; if (t < tbest) { tbest = t; surf = 2; dn = -dx; } }
    if (dy > 1e-6f) { float t = (SVS_TUNNEL_YB - Cy) / dy; if (t < tbest) { tbest = t; surf = 3; dn = dy; } }
    if (dy < -1e-6f) { float t = (SVS_TUNNEL_YT - Cy) / dy; if (t < tbest) { tbest = t; surf = 4; dn = -dy; } }
    float val = 150.f;
    if (surf != 0 && tbest > 0.f) {
        double hx = vw->C[0] + (double)tbest * dx, hy = vw->C[1] + (double)tbest * dy;
        double hz = vw->C[2] + (double)tbest * dz;
        double s = hz, t = (surf <= 2) ? hy : hx;
        float d2 = dx * dx + dy * dy + dz * dz;
        float fp_perp = tbest * sqrtf(d2) / vw->fx;
        float fp_slant = tbest * d2 / (vw->fx * dn);
        float fp = sqrtf(fp_perp * fp_slant);
        const float cells[3] = { 2.4f, 0.8f, 0.27f };
        const float amps[3] = { 70.f, 48.f, 26.f };
        float acc = 0.f;
        for (int k = 0; k < 3; ++k) {
            float fade = svs_clampf((cells[k] / fp - 1.0f) * 0.5f, 0.f, 1.f);
            if (fade > 0.f) {
                float n = svs_tile(s, t, cells[k], fp, vw->seed * 16u + (uint32_t)(surf * 4 + k));
                acc += amps[k] * fade * (2.f * n - 1.f);
            }
        }
        float fog = tbest / (tbest + 160.f);
        val = (118.f + acc) * (1.f - fog) + 150.f * fog;
    }
    uint32_t h = svs_hash3((uint32_t)px, (uint32_t)py, vw->noise_seed);
    float nz = (float)((h & 0xFF) + ((h >> 8) & 0xFF) + ((h >> 16) & 0xFF) + (h >> 24)) - 510.f;
    val += nz * 0.01353f; /* sigma ~ 2 grey levels */
    val = svs_clampf(val, 0.f, 255.f);
    return (uint8_t)(int)(val + 0.5f);
}

#endif
