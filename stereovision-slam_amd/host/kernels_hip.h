// kernels_hip.h — kernel provider of the product build: every call goes straight
// through the C ABI of libsvslam_hip.so (include/svslam.h).  No CPU path.
#pragma once
#include <stdexcept>
#include <string>
#include "../../include/svslam.h"

namespace svs {

class HipKernels {
public:
    static constexpr bool kHasDeviceMap = true;       // svslam_dmap_*: the map of every stream can live in device memory
    explicit HipKernels(const svslam_limits &lim)
    {
        int rc = svslam_create(&lim, &ctx_);
        if (rc != 0) {
            std::string msg = ctx_ ? svslam_last_error(ctx_) : "no usable HIP device (libsvslam_hip has no CPU path)";
            if (ctx_) svslam_destroy(ctx_);
            ctx_ = nullptr;
            throw std::runtime_error("svslam_create failed: " + msg);
        }
    }
    ~HipKernels() { if (ba_ctx_) svslam_destroy(ba_ctx_); if (ctx_) svslam_destroy(ctx_); }

    // A second context (own HIP stream + staging memory) for the backend, so a submitted
    // local-BA batch runs beside the frontend's calls the way the reference's Backend
    // thread runs beside Frontend (src/backend.cpp:345-367).
    void enable_backend_context(const svslam_limits &frontend_lim)
    {
        if (ba_ctx_) return;
        svslam_limits l = frontend_lim;
        l.max_slots = 1; l.max_pts = 8; l.max_corners = 8;
        if (svslam_create(&l, &ba_ctx_) != 0) {
            std::string msg = ba_ctx_ ? svslam_last_error(ba_ctx_) : "svslam_create";
            if (ba_ctx_) svslam_destroy(ba_ctx_);
            ba_ctx_ = nullptr;
            throw std::runtime_error("backend context: " + msg);
        }
        if (low_latency_) svslam_set_low_latency(ba_ctx_, 1);
    }
    HipKernels(const HipKernels &) = delete;
    HipKernels &operator=(const HipKernels &) = delete;

    svslam_ctx *ctx() { return ctx_; }
    void set_host_threads(int n) { svslam_set_host_threads(ctx_, n); if (ba_ctx_) svslam_set_host_threads(ba_ctx_, n); }
    const char *last_error()
    {
        if (ba_ctx_ && ba_failed_) return svslam_last_error(ba_ctx_);
        return svslam_last_error(ctx_);
    }
    svslam_ctx *backend_ctx() { return ba_ctx_ ? ba_ctx_ : ctx_; }
    int set_source_size(int src_w, int src_h) { return svslam_set_source_size(ctx_, src_w, src_h); }
    int set_low_latency(int on)
    {
        low_latency_ = on;
        ba_failed_ = false;
        if (ba_ctx_ && svslam_set_low_latency(ba_ctx_, on) != 0) { ba_failed_ = true; return -1; }   // last_error() then reports the backend context's reason
        return svslam_set_low_latency(ctx_, on);
    }

    // parameter tolerance of the pose-only LM (svslam_set_pose_only_xtol; pose-only runs on the frontend context only)
    int set_pose_only_xtol(double xtol) { ba_failed_ = false; return svslam_set_pose_only_xtol(ctx_, xtol); }

    int pyramid(int n, const int *slots, const void *const *imgs, const int *strides, int is_device)
    { return svslam_pyramid_batch(ctx_, n, slots, imgs, strides, is_device); }
    int track(int n, svslam_track_job *jobs, const void *const *imgs, const int *strides, int is_device,
              int total, const double *cam, const float *prev_xy, float *next_xy, const uint8_t *has_mp,
              const double *xyz, uint8_t *status, uint8_t *outlier, const svslam_lk_params *p, double chi2_th)
    { return svslam_track_batch(ctx_, n, jobs, imgs, strides, is_device, total, cam, prev_xy, next_xy, has_mp, xyz, status, outlier, p, chi2_th); }
    int rtrack(int n, svslam_rtrack_job *jobs, const void *const *imgs, const int *strides, int is_device, int total,
               const double *cam, float *out_xy, int *out_mp, const svslam_lk_params *p, double chi2_th)
    { return svslam_rtrack_batch(ctx_, n, jobs, imgs, strides, is_device, total, cam, out_xy, out_mp, p, chi2_th); }
    int rtrack_upload(int n, const int *streams, const int *ofs, const int *counts, const float *xy, const int *mp,
                      const double *xyz)
    { return svslam_rtrack_upload(ctx_, n, streams, ofs, counts, xy, mp, xyz); }
    int dmap_keyframe(int n, svslam_dmap_job *jobs, const void *const *left, const void *const *right, const int *strides,
                      int is_device, const double *cam_l, const double *ext_l, const double *cam_r, const double *ext_r,
                      const svslam_dmap_params *p)
    { return svslam_dmap_keyframe_batch(ctx_, n, jobs, left, right, strides, is_device, cam_l, ext_l, cam_r, ext_r, p); }
    int dmap_evicted(const svslam_dmap_evicted_rec **recs, int *n) { return svslam_dmap_evicted(ctx_, recs, n); }
    int dmap_ba_collect(int n, svslam_dmap_job *jobs, int *inflight) { return svslam_dmap_ba_collect(ctx_, n, jobs, inflight); }
    int dmap_read(int stream, long long *kf_frame, int *kf_id, double *kf_pose, int *kf_n, int *lm_id, double *lm_pos,
                  int *lm_obs, uint8_t *lm_state)
    { return svslam_dmap_read(ctx_, stream, kf_frame, kf_id, kf_pose, kf_n, lm_id, lm_pos, lm_obs, lm_state); }
    int lk(int n, const svslam_lk_job *jobs, int total, const float *prev_xy, float *next_xy, uint8_t *status,
           float *err, const svslam_lk_params *p)
    { return svslam_lk_batch(ctx_, n, jobs, total, prev_xy, next_xy, status, err, p); }
    int gftt(int n, const svslam_gftt_job *jobs, int total_rects, const float *rect_xy, int max_corners,
             double quality, double min_dist, float *out_xy, int *out_n)
    { return svslam_gftt_batch(ctx_, n, jobs, total_rects, rect_xy, max_corners, quality, min_dist, out_xy, out_n); }
    int triangulate(int n, const svslam_tri_job *jobs, int total, const double *cam_l, const double *ext_l,
                    const double *cam_r, const double *ext_r, const float *uv_l, const float *uv_r,
                    double *xyz, uint8_t *ok)
    { return svslam_triangulate_batch(ctx_, n, jobs, total, cam_l, ext_l, cam_r, ext_r, uv_l, uv_r, xyz, ok); }
    int local_ba(int n, svslam_ba_job *jobs, const double *cam_l, const double *ext_l, const double *cam_r,
                 const double *ext_r, int total_kf, double *poses, int total_lm, double *pts, int total_obs,
                 const int *okf, const int *olm, const uint8_t *oright, const float *ouv, double delta,
                 int iters, double *chi2)
    { return svslam_local_ba_batch(ctx_, n, jobs, cam_l, ext_l, cam_r, ext_r, total_kf, poses, total_lm, pts, total_obs, okf, olm, oright, ouv, delta, iters, chi2); }

    int local_ba_submit(int n, const svslam_ba_job *jobs, const double *cam_l, const double *ext_l, const double *cam_r,
                        const double *ext_r, int total_kf, const double *poses, int total_lm, const double *pts,
                        int total_obs, const int *okf, const int *olm, const uint8_t *oright, const float *ouv,
                        double delta, int iters)
    {
        int rc = svslam_local_ba_submit(backend_ctx(), n, jobs, cam_l, ext_l, cam_r, ext_r, total_kf, poses, total_lm, pts, total_obs, okf, olm, oright, ouv, delta, iters);
        ba_failed_ = rc != 0;
        return rc;
    }
    int local_ba_collect(int n, svslam_ba_job *jobs, int total_kf, double *poses, int total_lm, double *pts,
                         int total_obs, double *chi2)
    {
        int rc = svslam_local_ba_collect(backend_ctx(), n, jobs, total_kf, poses, total_lm, pts, total_obs, chi2);
        ba_failed_ = rc != 0;
        return rc;
    }

private:
    svslam_ctx *ctx_ = nullptr;
    svslam_ctx *ba_ctx_ = nullptr;
    bool ba_failed_ = false;
    int low_latency_ = 0;
};

} // namespace svs
