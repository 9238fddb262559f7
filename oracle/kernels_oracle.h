// kernels_oracle.h — CPU ORACLE (test infrastructure, NOT product code): the kernel provider of the
// CPU twin — every batched entry point of include/svslam.h served by the oracle's single-threaded
// restatement.  Included by oracle/pipeline_cpu.cpp (the twin library) and by the C++ tests of the
// host layer; never by the product.
//
// Faithful to the reference's cost model: cv::calcOpticalFlowPyrLK is handed plain images
// (src/frontend.cpp:105,353), so both pyramids and the Scharr derivatives are rebuilt on every LK
// call; BA uses numeric Jacobians like g2o does for EdgeProjection (g2o_types.h:176-229) unless
// SVS_ORACLE_BA_JAC overrides it.
#pragma once
#include <cstdlib>
#include <cstring>
#include <string>
#include <algorithm>
#include <vector>

#include "svs_oracle.h"
#include "../include/svslam.h"
#include "../stereovision-slam_amd/host/se3.h"

namespace svs {

class OracleKernels {
public:
    explicit OracleKernels(const svslam_limits &lim) : w_(lim.width), h_(lim.height), slots_((size_t)lim.max_slots)
    {
        const char *jm = std::getenv("SVS_ORACLE_BA_JAC");
        jac_mode_ = jm ? std::atoi(jm) : 1;
    }
    void *ctx() { return nullptr; }
    void set_host_threads(int) {}
    const char *last_error() { return err_.c_str(); }

    int set_source_size(int src_w, int src_h) { src_w_ = src_w; src_h_ = src_h; return 0; }
    int set_low_latency(int) { return 0; }      // a kernel shape of the HIP library; nothing to emulate
    int set_pose_only_xtol(double) { return 0; } // a stopping rule of the HIP library; the twin runs g2o's schedule to the last trial
    int pyramid(int n, const int *slots, const void *const *imgs, const int *strides, int)
    {
        for (int i = 0; i < n; ++i) {
            std::vector<uint8_t> &d = slots_[(size_t)slots[i]];
            d.resize((size_t)w_ * h_);
            const uint8_t *s = static_cast<const uint8_t *>(imgs[i]);
            if (src_w_ > 0) orc_decimate(s, src_w_, src_h_, strides[i], d.data(), w_, h_, w_);
            else for (int y = 0; y < h_; ++y) std::memcpy(&d[(size_t)y * w_], s + (size_t)y * strides[i], (size_t)w_);
        }
        return 0;
    }
    int lk(int n, const svslam_lk_job *jobs, int, const float *prev_xy, float *next_xy, uint8_t *status,
           float *err, const svslam_lk_params *p)
    {
        orc_lk_params prm = { p->max_level, p->max_iter, p->epsilon, p->min_eig_thr, p->use_initial_flow };
        for (int i = 0; i < n; ++i) {
            const svslam_lk_job &j = jobs[i];
            if (j.npts == 0) continue;
            orc_lk(slots_[(size_t)j.prev_slot].data(), w_, slots_[(size_t)j.next_slot].data(), w_, w_, h_, j.npts,
                   prev_xy + 2 * j.pt_ofs, next_xy + 2 * j.pt_ofs, status + j.pt_ofs, err ? err + j.pt_ofs : nullptr, &prm);
        }
        return 0;
    }
    int track(int n, svslam_track_job *jobs, const void *const *imgs, const int *strides, int is_device,
              int total, const double *cam, const float *prev_xy, float *next_xy, const uint8_t *has_mp,
              const double *xyz, uint8_t *status, uint8_t *outlier, const svslam_lk_params *p, double chi2_th)
    {
        std::vector<int> sl((size_t)n);
        for (int i = 0; i < n; ++i) sl[(size_t)i] = jobs[i].next_slot;
        pyramid(n, sl.data(), imgs, strides, is_device);
        std::vector<svslam_lk_job> lj((size_t)n);
        for (int i = 0; i < n; ++i) lj[(size_t)i] = { jobs[i].prev_slot, jobs[i].next_slot, jobs[i].pt_ofs, jobs[i].npts };
        std::vector<float> err((size_t)(total > 0 ? total : 1));
        lk(n, lj.data(), total, prev_xy, next_xy, status, err.data(), p);
        for (int i = 0; i < n; ++i) {
            svslam_track_job &j = jobs[i];
            std::vector<double> P; std::vector<float> uv; std::vector<int> idx;
            int ntr = 0;
            for (int q = 0; q < j.npts; ++q) {
                const int g = j.pt_ofs + q;
                const float x = next_xy[2 * g], y = next_xy[2 * g + 1];
                bool ok = status[g] != 0;
                if (y < 0.f || y >= (float)h_ || x < 0.f || x >= (float)w_) ok = false;
                status[g] = ok ? 1 : 0;
                outlier[g] = 0;
                if (!ok) continue;
                ++ntr;
                if (!has_mp[g]) continue;
                P.push_back(xyz[3 * g]); P.push_back(xyz[3 * g + 1]); P.push_back(xyz[3 * g + 2]);
                uv.push_back(x); uv.push_back(y); idx.push_back(g);
            }
            std::vector<uint8_t> outl(idx.size() + 1);
            j.n_inlier = orc_pose_only((int)idx.size(), cam, j.pose, P.data(), uv.data(), outl.data(), chi2_th, 4, 10);
            for (size_t e = 0; e < idx.size(); ++e) outlier[idx[e]] = outl[e];
            j.n_tracked = ntr;
        }
        return 0;
    }
    // resident feature lists, emulated in host memory (same semantics as svslam_rtrack_*)
    int rtrack(int n, svslam_rtrack_job *jobs, const void *const *imgs, const int *strides, int is_device, int total,
               const double *cam, float *out_xy, int *out_mp, const svslam_lk_params *p, double chi2_th)
    {
        const size_t T = (size_t)(total > 0 ? total : 1);
        std::vector<float> prev(2 * T), next(2 * T);
        std::vector<uint8_t> has(T), status(T), outl(T);
        std::vector<double> xyz(3 * T);
        std::vector<svslam_track_job> tj((size_t)n);
        for (int i = 0; i < n; ++i) {
            svslam_rtrack_job &j = jobs[i];
            if ((size_t)j.stream >= rt_.size()) rt_.resize((size_t)j.stream + 1);
            const RtList &L = rt_[(size_t)j.stream];
            if ((int)L.mp.size() != j.npts) { err_ = "rtrack: feature count mismatch"; return -1; }
            const SE3 Tc(j.T_cam_w);
            Camera K; K.fx = cam[0]; K.fy = cam[1]; K.cx = cam[2]; K.cy = cam[3];
            for (int q = 0; q < j.npts; ++q) {
                const size_t g = (size_t)j.pt_ofs + q;
                prev[2 * g] = L.xy[2 * q]; prev[2 * g + 1] = L.xy[2 * q + 1];
                if (L.mp[(size_t)q] >= 0) {
                    double uv[2];
                    K.project(Tc, &L.xyz[3 * (size_t)q], uv);
                    next[2 * g] = (float)uv[0]; next[2 * g + 1] = (float)uv[1];
                    has[g] = 1;
                    xyz[3 * g] = L.xyz[3 * (size_t)q]; xyz[3 * g + 1] = L.xyz[3 * (size_t)q + 1]; xyz[3 * g + 2] = L.xyz[3 * (size_t)q + 2];
                } else {
                    next[2 * g] = L.xy[2 * q]; next[2 * g + 1] = L.xy[2 * q + 1];
                    has[g] = 0;
                    xyz[3 * g] = 0; xyz[3 * g + 1] = 0; xyz[3 * g + 2] = 1;
                }
            }
            tj[(size_t)i].prev_slot = j.prev_slot; tj[(size_t)i].next_slot = j.next_slot;
            tj[(size_t)i].pt_ofs = j.pt_ofs; tj[(size_t)i].npts = j.npts;
            std::memcpy(tj[(size_t)i].pose, j.pose, 56);
        }
        if (track(n, tj.data(), imgs, strides, is_device, total, cam, prev.data(), next.data(), has.data(), xyz.data(),
                  status.data(), outl.data(), p, chi2_th)) return -1;
        for (int i = 0; i < n; ++i) {
            svslam_rtrack_job &j = jobs[i];
            RtList &L = rt_[(size_t)j.stream];
            RtList N;
            int ne = 0, no = 0;
            for (int q = 0; q < j.npts; ++q) {
                const size_t g = (size_t)j.pt_ofs + q;
                if (!status[g]) continue;
                int mp = L.mp[(size_t)q];
                if (mp >= 0) { ++ne; if (outl[g]) { mp = -1; ++no; } }
                const size_t r = N.mp.size();
                N.xy.push_back(next[2 * g]); N.xy.push_back(next[2 * g + 1]);
                N.mp.push_back(mp);
                N.xyz.push_back(xyz[3 * g]); N.xyz.push_back(xyz[3 * g + 1]); N.xyz.push_back(xyz[3 * g + 2]);
                if (out_xy) { out_xy[2 * ((size_t)j.pt_ofs + r)] = next[2 * g]; out_xy[2 * ((size_t)j.pt_ofs + r) + 1] = next[2 * g + 1]; }
                if (out_mp) out_mp[(size_t)j.pt_ofs + r] = mp;
            }
            std::memcpy(j.pose, tj[(size_t)i].pose, 56);
            j.n_tracked = (int)N.mp.size(); j.n_edges = ne; j.n_outlier = no;
            L = std::move(N);
        }
        return 0;
    }
    // the device-resident map is a property of the HIP provider (the twin IS the host map that checks it)
    int dmap_keyframe(int, svslam_dmap_job *, const void *const *, const void *const *, const int *, int, const double *,
                      const double *, const double *, const double *, const svslam_dmap_params *)
    { err_ = "the CPU twin keeps its map on the host (device_map = 0)"; return -1; }
    int dmap_evicted(const svslam_dmap_evicted_rec **, int *) { err_ = "no device map in the CPU twin"; return -1; }
    int dmap_ba_collect(int, svslam_dmap_job *, int *) { err_ = "no device map in the CPU twin"; return -1; }
    int dmap_read(int, long long *, int *, double *, int *, int *, double *, int *, uint8_t *) { err_ = "no device map in the CPU twin"; return -1; }
    int rtrack_upload(int n, const int *streams, const int *ofs, const int *counts, const float *xy, const int *mp,
                      const double *xyz)
    {
        for (int i = 0; i < n; ++i) {
            if ((size_t)streams[i] >= rt_.size()) rt_.resize((size_t)streams[i] + 1);
            RtList &L = rt_[(size_t)streams[i]];
            const size_t o = (size_t)ofs[i], c = (size_t)counts[i];
            L.xy.assign(xy + 2 * o, xy + 2 * (o + c));
            L.mp.assign(mp + o, mp + o + c);
            L.xyz.assign(xyz + 3 * o, xyz + 3 * (o + c));
        }
        return 0;
    }
    int gftt(int n, const svslam_gftt_job *jobs, int, const float *rect_xy, int max_corners, double quality,
             double min_dist, float *out_xy, int *out_n)
    {
        for (int i = 0; i < n; ++i)
            out_n[i] = orc_gftt(slots_[(size_t)jobs[i].slot].data(), w_, w_, h_, rect_xy + 2 * jobs[i].rect_ofs,
                                jobs[i].nrect, max_corners, quality, min_dist, out_xy + (size_t)i * max_corners * 2);
        return 0;
    }
    int triangulate(int n, const svslam_tri_job *jobs, int, const double *cam_l, const double *ext_l,
                    const double *cam_r, const double *ext_r, const float *uv_l, const float *uv_r,
                    double *xyz, uint8_t *ok)
    {
        for (int i = 0; i < n; ++i) {
            const svslam_tri_job &j = jobs[i];
            orc_triangulate(j.npts, cam_l, ext_l, cam_r, ext_r, uv_l + 2 * j.pt_ofs, uv_r + 2 * j.pt_ofs, j.T_wc,
                            j.zmax, xyz + 3 * j.pt_ofs, ok + j.pt_ofs);
        }
        return 0;
    }
    int local_ba(int n, svslam_ba_job *jobs, const double *cam_l, const double *ext_l, const double *cam_r,
                 const double *ext_r, int, double *poses, int, double *pts, int, const int *okf, const int *olm,
                 const uint8_t *oright, const float *ouv, double delta, int iters, double *chi2)
    {
        for (int i = 0; i < n; ++i) {
            svslam_ba_job &j = jobs[i];
            j.iters_done = orc_local_ba(cam_l, ext_l, cam_r, ext_r, j.nkf, poses + 7 * j.kf_ofs, j.nlm,
                                        pts + 3 * j.lm_ofs, j.nobs, okf + j.obs_ofs, olm + j.obs_ofs,
                                        oright + j.obs_ofs, ouv + 2 * j.obs_ofs, delta, iters, jac_mode_,
                                        chi2 + j.obs_ofs);
        }
        return 0;
    }

    // the CPU twin has no second device queue: submit solves at once and parks the result
    void enable_backend_context(const svslam_limits &) {}
    int local_ba_submit(int n, const svslam_ba_job *jobs, const double *cam_l, const double *ext_l, const double *cam_r,
                        const double *ext_r, int total_kf, const double *poses, int total_lm, const double *pts,
                        int total_obs, const int *okf, const int *olm, const uint8_t *oright, const float *ouv,
                        double delta, int iters)
    {
        ba_jobs_.assign(jobs, jobs + n);
        ba_poses_.assign(poses, poses + 7 * (size_t)total_kf);
        ba_pts_.assign(pts, pts + 3 * (size_t)total_lm);
        ba_chi2_.assign((size_t)std::max(total_obs, 1), 0.0);
        return local_ba(n, ba_jobs_.data(), cam_l, ext_l, cam_r, ext_r, total_kf, ba_poses_.data(), total_lm,
                        ba_pts_.data(), total_obs, okf, olm, oright, ouv, delta, iters, ba_chi2_.data());
    }
    int local_ba_collect(int n, svslam_ba_job *jobs, int total_kf, double *poses, int total_lm, double *pts,
                         int total_obs, double *chi2)
    {
        if ((int)ba_jobs_.size() != n) { err_ = "local_ba_collect: nothing submitted"; return -1; }
        for (int i = 0; i < n; ++i) jobs[i].iters_done = ba_jobs_[i].iters_done;
        std::copy(ba_poses_.begin(), ba_poses_.begin() + 7 * (size_t)total_kf, poses);
        std::copy(ba_pts_.begin(), ba_pts_.begin() + 3 * (size_t)total_lm, pts);
        std::copy(ba_chi2_.begin(), ba_chi2_.begin() + (size_t)total_obs, chi2);
        ba_jobs_.clear();
        return 0;
    }

private:
    struct RtList { std::vector<float> xy; std::vector<int> mp; std::vector<double> xyz; };
    std::vector<RtList> rt_;
    std::vector<svslam_ba_job> ba_jobs_;
    std::vector<double> ba_poses_, ba_pts_, ba_chi2_;
    int w_, h_;
    int src_w_ = 0, src_h_ = 0;
    std::vector<std::vector<uint8_t>> slots_;
    int jac_mode_ = 1;
    std::string err_;
};

} // namespace svs
