/*
 * svs_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Single-threaded plain-C restatement of the algorithms behind the hot path of
 * farhad-dalirani/StereoVision-SLAM.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product path
 * (libsvslam_hip.so) never links or calls it.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in un-vendored
 * third-party libraries (OpenCV 4.5.4, g2o, Eigen 3.4, Sophus; README.md:29-35
 * of the reference) that are absent from /root/reference and from this image,
 * and the reference ships no tests, golden vectors or fixtures.  What follows
 * restates the *published* algorithms of those libraries, anchored on the
 * reference's call sites (cited per function), to a declared operation order
 * (strict IEEE, no FMA contraction: build with -ffp-contract=off).  It is
 * cross-checked by independent numpy/scipy computations in tests/.
 *
 * Declared deviations from a literal OpenCV scalar build (documented in
 * DESIGN.md): the LK 2x2 normal-equation sums (A11,A12,A22,b1,b2) are
 * accumulated as exact integers and converted to float once (OpenCV's own
 * result depends on its SIMD lane layout); everything else follows the scalar
 * C++ code path.
 */
#ifndef SVS_ORACLE_H
#define SVS_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 4

/* ---- what-if knobs (sensitivity study of the recalled upstream conventions, tests/oracle_sensitivity.py) ----
 * The oracle restates OpenCV / g2o from published descriptions (PARITY UNPINNED above).  Each knob flips ONE recalled
 * convention to its plausible alternative so that the effect of a misrecollection on corners, poses and trajectory
 * error can be measured.  All knobs are 0 by default — the oracle as declared; nothing but that study sets them.
 *   0  LK normal-equation sums: 1 = f32 accumulators in pixel order (OpenCV's scalar path) instead of exact integers
 *   1  g2o LM tau (lambda_0 = tau max diag H): 1 = 1e-3 instead of 1e-5
 *   2  g2o rho denominator: 1 = without the + 1e-3
 *   3  Sobel scale of goodFeaturesToTrack: 1 = 1/12 (no 1/255 for 8-bit input) instead of 1/3060
 *   4  cv::rectangle corner rounding (Point2f -> Point): 1 = half-up instead of half-to-even
 *   5  box filter of cornerMinEigenVal: 1 = f32 accumulators instead of f64
 *   6  goodFeaturesToTrack tie-break of equal responses: 1 = address ascending instead of descending
 *   7  Huber kernel of the local BA: 1 = delta^2 = chi2_th (kernel bends at chi2 > 5.991) instead of delta = chi2_th  */
#define ORC_WHATIF_N 8
extern int orc_whatif[ORC_WHATIF_N];
void orc_set_whatif(int which, int value);

/* ---- pyramids (cv::buildOpticalFlowPyramid, pyrDown, calcSharrDeriv) ---- */
typedef struct orc_plane {
    int w, h;
    int stride;          /* in elements                                      */
    int border;
    uint8_t *base;       /* allocation incl. border                          */
    uint8_t *data;       /* pixel (0,0)                                      */
} orc_plane;

typedef struct orc_pyr {
    int nlevels;
    orc_plane lv[ORC_MAX_LEVELS];
} orc_pyr;

void orc_pyr_build(orc_pyr *p, const uint8_t *img, int stride, int w, int h,
                   int max_level, int win);
void orc_pyr_free(orc_pyr *p);
/* one pyrDown step on tight buffers (test hook) */
void orc_pyrdown(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst,
                 int dstride);
/* Scharr derivative of a tight image: out int16 interleaved (dx,dy) */
void orc_scharr(const uint8_t *src, int w, int h, int stride, int16_t *out);
/* 1/2 nearest decimation of Dataset::NextFrame (src/dataset.cpp:126-129) */
void orc_decimate(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst,
                  int dw, int dh, int dstride);

/* ---- pyramidal LK (cv::calcOpticalFlowPyrLK; src/frontend.cpp:105,353) -- */
typedef struct orc_lk_params {
    int max_level;
    int max_iter;
    double epsilon;
    double min_eig_thr;
    int use_initial_flow;
} orc_lk_params;

void orc_lk(const uint8_t *prev, int pstride, const uint8_t *next, int nstride,
            int w, int h, int n, const float *prev_xy, float *next_xy,
            uint8_t *status, float *err, const orc_lk_params *p);
/* same on prebuilt pyramids (no rebuild) */
void orc_lk_pyr(const orc_pyr *prev, const orc_pyr *next, int n,
                const float *prev_xy, float *next_xy, uint8_t *status,
                float *err, const orc_lk_params *p);

/* ---- GFTT (cv::GFTTDetector; src/frontend.cpp:24,42-51) ----------------- */
void orc_min_eig_map(const uint8_t *img, int stride, int w, int h, float *eig);
void orc_gftt_mask(uint8_t *mask, int w, int h, const float *rect_xy, int nrect);
int  orc_gftt(const uint8_t *img, int stride, int w, int h,
              const float *rect_xy, int nrect, int max_corners, double quality,
              double min_dist, float *out_xy);

/* ---- SE(3) helpers (Sophus::SE3d layout: qx qy qz qw tx ty tz) ---------- */
void orc_se3_identity(double T[7]);
void orc_se3_exp(const double xi[6], double T[7]);
void orc_se3_log(const double T[7], double xi[6]);
void orc_se3_mul(const double A[7], const double B[7], double C[7]);
void orc_se3_inv(const double T[7], double Ti[7]);
void orc_se3_act(const double T[7], const double p[3], double out[3]);

/* ---- triangulation (algorithm.h:10-87; src/frontend.cpp:165-174,277-295) - */
int orc_triangulate_dlt(const double ext_l[7], const double ext_r[7],
                        const double pl[3], const double pr[3], double out[3]);
void orc_triangulate(int n, const double cam_l[4], const double ext_l[7],
                     const double cam_r[4], const double ext_r[7],
                     const float *uv_l, const float *uv_r, const double T_wc[7],
                     double zmax, double *out_xyz, uint8_t *out_ok);

/* ---- pose-only LM (src/frontend.cpp:394-558) --------------------------- */
int orc_pose_only(int n, const double cam[4], double pose[7], const double *xyz,
                  const float *uv, uint8_t *outlier, double chi2_th, int rounds,
                  int iters);

/* ---- local BA (src/backend.cpp:22-164) ---------------------------------- */
/* jac_mode: 0 analytic, 1 numeric central differences delta=1e-9 (what g2o
 * does for EdgeProjection, which has no linearizeOplus: g2o_types.h:176-229) */
int orc_local_ba(const double cam_l[4], const double ext_l[7],
                 const double cam_r[4], const double ext_r[7], int nkf,
                 double *poses, int nlm, double *pts, int nobs,
                 const int *obs_kf, const int *obs_lm,
                 const uint8_t *obs_is_right, const float *obs_uv,
                 double huber_delta, int iters, int jac_mode,
                 double *edge_chi2);

/* LM trajectory hooks: ORC_TRACE_REC doubles per LM trial (iteration [pose-only: 16 round + iteration],
 * lambda of the trial, chi2 before, chi2 of the trial state, rho, accepted) */
#define ORC_TRACE_REC 6
int orc_local_ba_trace(const double cam_l[4], const double ext_l[7],
                 const double cam_r[4], const double ext_r[7], int nkf,
                 double *poses, int nlm, double *pts, int nobs,
                 const int *obs_kf, const int *obs_lm,
                 const uint8_t *obs_is_right, const float *obs_uv,
                 double huber_delta, int iters, int jac_mode, double *edge_chi2,
                 double *trace, int trace_cap, int *trace_n);
int orc_pose_only_trace(int n, const double cam[4], double pose[7], const double *xyz,
                        const float *uv, uint8_t *outlier, double chi2_th, int rounds,
                        int iters, double *trace, int trace_cap, int *trace_n);

/* test hooks: Jacobians of EdgeProjection (mode 0 analytic, 1 numeric) and of
 * EdgeProjectionPoseOnly::linearizeOplus as the oracle evaluates them */
void orc_ba_jacobian(const double cam[4], const double ext[7], const double T[7], const double P[3],
                     const float uv[2], int mode, double Jp[12], double Jl[6]);
void orc_po_jacobian(const double cam[4], const double T[7], const double P[3], double J[12]);

#ifdef __cplusplus
}
#endif
#endif
