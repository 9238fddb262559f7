// k_rt.h — device-resident tracking state and the two per-feature bodies around a tracked frame.
// The features of every stream's last frame (position, map-point id, map-point position)
// stay in HBM in two alternating buffers per stream; a frame that is not a keyframe never
// costs the host a per-feature operation (Frontend::TrackLastFrame src/frontend.cpp:322-392 gather and
// :361-381 / EstimateCurrentPose :546-553 scatter become the two bodies below).
// They run as kernels of their own (k_geom.h: k_rt_gather, k_rt_finish) or inside the launches next to them:
// the gather as extra workgroups of the pyramid launch (k_pyramid.h: it depends on nothing the pyramid writes),
// the finish as the tail of the pose-only kernel (k_geom.h) — a lone camera's frame is three launches instead of six.
#pragma once
#include "dev_common.h"
#pragma clang fp contract(off)

struct RtJob {
    int stream, pt_ofs, npts, src_buf;
    double T_cam_w[7];      // cam_left.pose * predicted T_cw (src/camera.cpp:74-80)
    int n_tracked, n_edges, n_outlier, pad;
};
struct RtStore { float2 *xy[2]; int *mp[2]; double *xyz[2]; int max_pts; };
// the gather riding on another launch: chunks workgroups of `threads` threads per job (njobs = 0: nothing rides)
struct RtGatherArgs { const RtJob *jobs; RtStore rs; const double *cam; float2 *prev_xy, *next_xy; uint8_t *has_mp; double *xyz; int njobs, chunks; };

// feature i of the job's previous frame: LK inputs + pose-only inputs
__device__ __forceinline__ void rt_gather_point(const RtJob &jb, const RtStore &rs, const double *cam, int i, float2 *prev_xy,
                                                float2 *next_xy, uint8_t *has_mp, double *xyz)
{
    const size_t src = (size_t)jb.stream * rs.max_pts + i;
    const int pt = jb.pt_ofs + i;
    const float2 p = rs.xy[jb.src_buf][src];
    const int mp = rs.mp[jb.src_buf][src];
    prev_xy[pt] = p;
    if (mp >= 0) {
        const double *Xs = rs.xyz[jb.src_buf] + 3 * src;
        const double X[3] = { Xs[0], Xs[1], Xs[2] };
        double uv[2];
        d_project_exact(jb.T_cam_w, cam, X, uv);
        next_xy[pt] = make_float2((float)uv[0], (float)uv[1]);
        has_mp[pt] = 1;
        xyz[3 * (size_t)pt] = X[0]; xyz[3 * (size_t)pt + 1] = X[1]; xyz[3 * (size_t)pt + 2] = X[2];
    } else {
        next_xy[pt] = p;
        has_mp[pt] = 0;
        xyz[3 * (size_t)pt] = 0; xyz[3 * (size_t)pt + 1] = 0; xyz[3 * (size_t)pt + 2] = 1;
    }
}

// one wave per job: survivors, in order, become the features of the new frame (other buffer);
// an edge the pose optimisation classified as outlier loses its map point (:546-553).
// The compacted (xy, mp) list is also left in the staging arena for the host (keyframes).
__device__ __forceinline__ void rt_finish_wave(RtJob &jb, const RtStore &rs, const float2 *next_xy, const uint8_t *status,
                                               const uint8_t *outlier, const double *xyz, float2 *out_xy, int *out_mp, int lane)
{
    // (jb may be pinned host memory: fields read once)
    const int srcb = jb.src_buf, dstb = srcb ^ 1, npts = jb.npts, pt_ofs = jb.pt_ofs;
    const size_t sbase = (size_t)jb.stream * rs.max_pts;
    int base = 0, n_edges = 0, n_out = 0;
    for (int c0 = 0; c0 < npts; c0 += 64) {
        const int i = c0 + lane;
        const bool in = i < npts;
        const int pt = pt_ofs + (in ? i : 0);
        const bool ok = in && status[pt] != 0;
        const unsigned long long bal = __ballot(ok);
        int mp = -1;
        bool edge = false, outl = false;
        if (ok) {
            mp = rs.mp[srcb][sbase + i];
            edge = mp >= 0;
            outl = edge && outlier[pt] != 0;
            if (outl) mp = -1;
            const int r = base + __popcll(bal & ((1ull << lane) - 1ull));
            const float2 q = next_xy[pt];
            rs.xy[dstb][sbase + r] = q;
            rs.mp[dstb][sbase + r] = mp;
            double *Xd = rs.xyz[dstb] + 3 * (sbase + r);
            Xd[0] = xyz[3 * (size_t)pt]; Xd[1] = xyz[3 * (size_t)pt + 1]; Xd[2] = xyz[3 * (size_t)pt + 2];
            out_xy[pt_ofs + r] = q;
            out_mp[pt_ofs + r] = mp;
        }
        base += __popcll(bal);
        n_edges += __popcll(__ballot(edge));
        n_out += __popcll(__ballot(outl));
    }
    if (lane == 0) { jb.n_tracked = base; jb.n_edges = n_edges; jb.n_outlier = n_out; }
}
