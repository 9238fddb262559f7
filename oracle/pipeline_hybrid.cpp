// pipeline_hybrid.cpp — TEST INFRASTRUCTURE (not product code): BASELINE config 2,
// "HIP GFTT+LK frontend only (backend still g2o CPU)".  The product's host pipeline over
// the HIP kernels for every frontend call site, with Backend::Optimize served by the
// oracle's g2o-shaped CPU restatement (numeric Jacobians unless SVS_ORACLE_BA_JAC=0).
// Built only by oracle/Makefile, loaded only by tests/ (tests/pipe_cpu.py:make_hybrid).
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "svs_oracle.h"
#include "../stereovision-slam_amd/host/kernels_hip.h"

namespace svs {

class HybridKernels : public HipKernels {
public:
    explicit HybridKernels(const svslam_limits &lim) : HipKernels(lim)
    {
        const char *jm = std::getenv("SVS_ORACLE_BA_JAC");
        jac_mode_ = jm ? std::atoi(jm) : 1;
    }
    void enable_backend_context(const svslam_limits &) {}
    int local_ba_submit(int n, const svslam_ba_job *jobs, const double *cam_l, const double *ext_l, const double *cam_r,
                        const double *ext_r, int total_kf, const double *poses, int total_lm, const double *pts,
                        int total_obs, const int *okf, const int *olm, const uint8_t *oright, const float *ouv,
                        double delta, int iters)
    {
        jobs_.assign(jobs, jobs + n);
        poses_.assign(poses, poses + 7 * (size_t)total_kf);
        pts_.assign(pts, pts + 3 * (size_t)total_lm);
        chi2_.assign((size_t)std::max(total_obs, 1), 0.0);
        for (int i = 0; i < n; ++i) {
            svslam_ba_job &j = jobs_[(size_t)i];
            j.iters_done = orc_local_ba(cam_l, ext_l, cam_r, ext_r, j.nkf, poses_.data() + 7 * j.kf_ofs, j.nlm,
                                        pts_.data() + 3 * j.lm_ofs, j.nobs, okf + j.obs_ofs, olm + j.obs_ofs,
                                        oright + j.obs_ofs, ouv + 2 * j.obs_ofs, delta, iters, jac_mode_,
                                        chi2_.data() + j.obs_ofs);
        }
        return 0;
    }
    int local_ba_collect(int n, svslam_ba_job *jobs, int total_kf, double *poses, int total_lm, double *pts,
                         int total_obs, double *chi2)
    {
        if ((int)jobs_.size() != n) return -1;
        for (int i = 0; i < n; ++i) jobs[i].iters_done = jobs_[(size_t)i].iters_done;
        std::copy(poses_.begin(), poses_.begin() + 7 * (size_t)total_kf, poses);
        std::copy(pts_.begin(), pts_.begin() + 3 * (size_t)total_lm, pts);
        std::copy(chi2_.begin(), chi2_.begin() + (size_t)total_obs, chi2);
        jobs_.clear();
        return 0;
    }

private:
    std::vector<svslam_ba_job> jobs_;
    std::vector<double> poses_, pts_, chi2_;
    int jac_mode_ = 1;
};

} // namespace svs

#define SVS_PIPE_KERNELS svs::HybridKernels
#define SVS_PIPE_MAKE_KERNELS(lim) new svs::HybridKernels(lim)
#define SVS_PIPE_IMAGES_ARE_DEVICE 1
#include "../stereovision-slam_amd/host/pipeline_capi_impl.h"

extern "C" void *svs_pipe_kernel_ctx(void *p) { return static_cast<PipeHandle *>(p)->kernels->ctx(); }
extern "C" void *svs_pipe_backend_ctx(void *p) { return static_cast<PipeHandle *>(p)->kernels->ctx(); }
