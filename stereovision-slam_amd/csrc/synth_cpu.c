/*
 * synth_cpu.c — CPU generator of the synthetic stereo stream (input data for
 * tests and for the CPU-baseline leg; see synth_scene.h).  Not the hot path.
 */
#include <stddef.h>
#include "synth_traj.h"

void svs_synth_render_view(const svs_synth_view *v, int w, int h, uint8_t *out, int stride)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) out[(size_t)y * stride + x] = svs_synth_pixel(v, x, y);
}

/* render the left/right pair of frame `frame` of stream `seed` */
void svs_synth_render_pair(uint32_t seed, int frame, int w, int h, const double cam[4],
                           double baseline, uint8_t *left, uint8_t *right, int stride)
{
    svs_synth_view vl, vr;
    svs_synth_views(seed, frame, cam, baseline, &vl, &vr);
    svs_synth_render_view(&vl, w, h, left, stride);
    svs_synth_render_view(&vr, w, h, right, stride);
}

void svs_synth_gt(uint32_t seed, int frame, double T[7]) { svs_synth_gt_pose(seed, frame, T); }

void svs_synth_make_views(uint32_t seed, int frame, const double cam[4], double baseline,
                          svs_synth_view *vl, svs_synth_view *vr)
{
    svs_synth_views(seed, frame, cam, baseline, vl, vr);
}

/* views of nframes consecutive frames of n streams, laid out [stream][frame] (bench input) */
void svs_synth_make_views_batch(int n, const uint32_t *seeds, int frame0, int nframes, const double cam[4],
                                double baseline, svs_synth_view *vl, svs_synth_view *vr)
{
    for (int s = 0; s < n; ++s)
        for (int f = 0; f < nframes; ++f)
            svs_synth_views(seeds[s], frame0 + f, cam, baseline, vl + (size_t)s * nframes + f, vr + (size_t)s * nframes + f);
}
