// pipeline_capi_impl.h — implementation of pipeline_capi.h, generic in the kernel
// provider.  Included by exactly one .cpp per library with SVS_PIPE_KERNELS and
// SVS_PIPE_MAKE_KERNELS defined.
#pragma once
#include <cstring>
#include <malloc.h>
#include <memory>
#include <string>
#include "pipeline_capi.h"
#include "slam_host.h"

namespace {
thread_local std::string g_err;

struct PipeHandle {
    std::unique_ptr<SVS_PIPE_KERNELS> kernels;
    std::unique_ptr<svs::Pipeline<SVS_PIPE_KERNELS>> pipe;
    std::vector<const void *> lp, rp;
    std::vector<svs::FrameResult> res;
};

svs::Config to_config(const svs_pipe_config &c)
{
    svs::Config g;
    g.num_features = c.num_features; g.num_features_init = c.num_features_init;
    g.num_features_tracking = c.num_features_tracking; g.num_features_tracking_bad = c.num_features_tracking_bad;
    g.num_features_needed_for_keyframe = c.num_features_needed_for_keyframe;
    g.max_triangulation_depth = c.max_triangulation_depth;
    g.num_active_keyframes = c.num_active_keyframes; g.backend_on = c.backend_on; g.chi2_th = c.chi2_th;
    g.width = c.width; g.height = c.height;
    g.cam_l.fx = c.cam_l[0]; g.cam_l.fy = c.cam_l[1]; g.cam_l.cx = c.cam_l[2]; g.cam_l.cy = c.cam_l[3];
    g.cam_r.fx = c.cam_r[0]; g.cam_r.fy = c.cam_r[1]; g.cam_r.cx = c.cam_r[2]; g.cam_r.cy = c.cam_r[3];
    g.cam_l.pose = svs::SE3(c.ext_l); g.cam_r.pose = svs::SE3(c.ext_r);
    g.src_width = c.src_width; g.src_height = c.src_height;
    g.resident_track = c.resident_track;
    g.max_pts = c.max_pts > 0 ? c.max_pts : 512;
    g.max_kf = c.num_active_keyframes + 1; g.max_lm = c.max_lm; g.max_obs = c.max_obs;
    g.device_map = (c.device_map && c.resident_track) ? 1 : 0;
    g.backend_lag = c.backend_lag > 0 ? c.backend_lag : 1;
    return g;
}
} // namespace

extern "C" {

const char *svs_pipe_last_error(void) { return g_err.c_str(); }

void svs_pipe_tune_allocator(void)
{
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TOP_PAD, 64 << 20);
}

void *svs_pipe_create(const svs_pipe_config *cfg, int nstreams, int device)
{
    try {
        std::unique_ptr<PipeHandle> h(new PipeHandle());
        svslam_limits lim;
        std::memset(&lim, 0, sizeof(lim));
        lim.device = device; lim.width = cfg->width; lim.height = cfg->height;
        lim.max_slots = 3 * nstreams; lim.max_jobs = 2 * nstreams; // one call may build left+right pyramids
        lim.max_pts = cfg->max_pts > 0 ? cfg->max_pts : 512; lim.max_corners = cfg->num_features;
        lim.max_kf = cfg->num_active_keyframes + 1; lim.max_lm = cfg->max_lm; lim.max_obs = cfg->max_obs;
        // the map on the device (any backend mode: with backend_on 2 the local BA of a keyframe runs on a second stream of the
        // same context, svslam_dmap_params::ba_defer); otherwise backend_on 2 gives the BA its own context
        const bool dmap = cfg->device_map && cfg->resident_track;
        lim.max_streams = (cfg->resident_track && (cfg->backend_on <= 1 || dmap)) ? nstreams : 0;
        lim.device_map = dmap ? 1 : 0;
        svslam_limits lim_front = lim;
        if (cfg->backend_on >= 2 && !dmap) { lim_front.max_kf = 0; lim_front.max_lm = 0; lim_front.max_obs = 0; }   // BA lives in its own context
        h->kernels.reset(SVS_PIPE_MAKE_KERNELS(lim_front));
        if (cfg->backend_on >= 2 && !dmap) {
            svslam_limits lim_ba = lim;
            lim_ba.max_jobs = nstreams;
            h->kernels->enable_backend_context(lim_ba);
        }
        h->kernels->set_host_threads(cfg->host_threads > 0 ? cfg->host_threads : 1);
        h->kernels->set_low_latency(cfg->low_latency);
        if (cfg->src_width > 0 && h->kernels->set_source_size(cfg->src_width, cfg->src_height) != 0)
            throw std::runtime_error(std::string("source size: ") + h->kernels->last_error());
        h->pipe.reset(new svs::Pipeline<SVS_PIPE_KERNELS>(to_config(*cfg), *h->kernels, nstreams,
                                                             cfg->host_threads > 0 ? cfg->host_threads : 1));
        h->lp.resize(nstreams); h->rp.resize(nstreams); h->res.resize(nstreams);
        return h.release();
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}

void svs_pipe_destroy(void *p) { delete static_cast<PipeHandle *>(p); }

static void copy_results(const PipeHandle *h, svs_frame_result *out)
{
    for (size_t s = 0; s < h->res.size(); ++s) {
        const svs::FrameResult &r = h->res[s];
        std::memcpy(out[s].pose, r.pose, sizeof(r.pose));
        out[s].status = r.status; out[s].is_keyframe = r.is_keyframe;
        out[s].n_features = r.n_features; out[s].n_inliers = r.n_inliers;
        out[s].frame_id = r.frame_id; out[s].keyframe_id = r.keyframe_id;
    }
}

int svs_pipe_step(void *p, const void *const *left, const void *const *right, int is_device,
                  svs_frame_result *out)
{
    PipeHandle *h = static_cast<PipeHandle *>(p);
    try {
        h->pipe->step(left, right, nullptr, is_device, h->res.data());
        copy_results(h, out);
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

int svs_pipe_run_device(void *p, const void *left_base, const void *right_base, long long stream_stride,
                        long long frame_stride, int first_frame, int nframes, svs_frame_result *out)
{
    PipeHandle *h = static_cast<PipeHandle *>(p);
    const int S = h->pipe->nstreams();
    try {
        for (int f = 0; f < nframes; ++f) {
            for (int s = 0; s < S; ++s) {
                long long off = (long long)s * stream_stride + (long long)(first_frame + f) * frame_stride;
                h->lp[s] = static_cast<const unsigned char *>(left_base) + off;
                h->rp[s] = static_cast<const unsigned char *>(right_base) + off;
            }
            h->pipe->step(h->lp.data(), h->rp.data(), nullptr, SVS_PIPE_IMAGES_ARE_DEVICE, h->res.data());
            if (out) copy_results(h, out + (size_t)f * S);
        }
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

int svs_pipe_flush(void *p)
{
    try {
        static_cast<PipeHandle *>(p)->pipe->Flush();
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

int svs_pipe_save_outputs(void *p, int stream, const char *dir, const char *dataset_dir, int left_cam_index)
{
    PipeHandle *h = static_cast<PipeHandle *>(p);
    if (stream < 0 || stream >= h->pipe->nstreams()) return -1;
    if (svs_pipe_flush(p)) return -3;
    return h->pipe->SaveOutputs(stream, dir, dataset_dir, left_cam_index) ? 0 : -2;
}

long long svs_pipe_map_snapshot(void *p, int stream, long long *ints, long long cap_i, double *dbl, long long cap_d)
{
    PipeHandle *h = static_cast<PipeHandle *>(p);
    if (stream < 0 || stream >= h->pipe->nstreams()) return -1;
    if (!h->pipe->map_on_host()) return -2;
    if (svs_pipe_flush(p)) return -3;
    const svs::Map &m = h->pipe->stream(stream).map;
    long long ni = 2 + (long long)m.active_keyframes_.size(), nd = 7 * (long long)m.active_keyframes_.size() + 3 * (long long)m.active_landmarks_.size();
    for (const svs::MapPoint *mp : m.active_landmarks_) ni += 3 + 3 * (long long)mp->observations.size();
    if (ni > cap_i || nd > cap_d || !ints || !dbl) return ni;
    long long *o = ints;
    double *d = dbl;
    *o++ = (long long)m.active_keyframes_.size();
    for (const svs::Frame *kf : m.active_keyframes_) {
        *o++ = kf->keyframe_id;
        std::memcpy(d, kf->pose.v, sizeof(double) * 7); d += 7;
    }
    *o++ = (long long)m.active_landmarks_.size();
    for (const svs::MapPoint *mp : m.active_landmarks_) {
        *o++ = mp->id; *o++ = mp->observed_times; *o++ = (long long)mp->observations.size();
        for (size_t i = 0; i < mp->observations.size(); ++i) {
            const svs::ObsRef &r = mp->observations[i];
            *o++ = r.frame->keyframe_id; *o++ = r.is_left ? 0 : 1; *o++ = r.idx;
        }
        std::memcpy(d, mp->pos, sizeof(double) * 3); d += 3;
    }
    return ni;
}

int svs_pipe_counters_get(void *p, svs_pipe_counters *out)
{
    const svs::Counters &c = static_cast<PipeHandle *>(p)->pipe->counters();
    out->frames = c.frames; out->keyframes = c.keyframes; out->track_pts = c.track_pts; out->pose_edges = c.pose_edges;
    out->gftt_calls = c.gftt_calls; out->gftt_rects = c.gftt_rects; out->corners = c.corners;
    out->right_pts = c.right_pts; out->tri_pts = c.tri_pts; out->ba_calls = c.ba_calls; out->ba_edges = c.ba_edges;
    out->ba_kf = c.ba_kf; out->ba_lm = c.ba_lm; out->ba_iters = c.ba_iters; out->pyr_left = c.pyr_left;
    out->pyr_right = c.pyr_right;
    out->ns_step = c.ns_step; out->ns_kernel_calls = c.ns_kernel_calls;
    out->corners_dropped = c.corners_dropped; out->ba_skipped = c.ba_skipped;
    out->ba_pairs = c.ba_pairs; out->ba_trials = c.ba_trials;
    out->lm_full = c.lm_full;
    out->lm_total = c.lm_created_dev;       // (device map: the host Map allocates no ids; host map: lm_created_dev stays 0)
    out->lm_resident = 0;
    svs::Pipeline<SVS_PIPE_KERNELS> &pp = *static_cast<PipeHandle *>(p)->pipe;
    for (int s = 0; s < pp.nstreams(); ++s) {
        out->lm_total += (long long)pp.stream(s).map.num_landmarks();
        out->lm_resident += (long long)pp.stream(s).map.num_resident_landmarks();
    }
    return 0;
}

} // extern "C"
