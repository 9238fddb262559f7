/*
 * svslam.h — C ABI of libsvslam_hip.so: the MI355X (gfx950) implementation of the
 * per-frame hot path of farhad-dalirani/StereoVision-SLAM.
 *
 * The reference has no FFI layer; its hot path is five third-party library call
 * sites inside two private C++ members (Frontend::Track, Backend::Optimize).
 * Each entry point below replaces exactly one of those call sites (cited per
 * function, file:line into the reference tree).  The ABI is flat: plain
 * pointers and sizes, caller-owned host buffers, no C++/torch types, int
 * return codes (0 = ok, <0 = error, see svslam_last_error), never throws.
 *
 * Everything is *batched over jobs*: one job = one call site invocation of one
 * SLAM stream.  A single-stream caller passes njobs = 1; a multi-stream host
 * (one process per GPU, S streams in lockstep) passes S jobs and gets one
 * launch sequence for all of them.  Jobs are independent.
 *
 * Conventions
 *   - images: u8, row-major, (width,height) fixed at context creation
 *   - points: float32 (x,y) interleaved, exactly cv::Point2f
 *   - SE(3):  double[7] = unit quaternion (x,y,z,w) + translation (x,y,z),
 *             the memory layout of Sophus::SE3d; T_cw (world -> stereo rig)
 *   - camera: double[4] = fx, fy, cx, cy ; extrinsic = SE(3) rig -> camera
 */
#ifndef SVSLAM_H
#define SVSLAM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVSLAM_PYR_LEVELS 4      /* maxLevel 3  (src/frontend.cpp:107,355) */
#define SVSLAM_LK_WIN 11         /* cv::Size(11,11) (src/frontend.cpp:107,355) */

typedef struct svslam_ctx svslam_ctx;

typedef struct svslam_limits {
    int device;        /* HIP device ordinal                                   */
    int width, height; /* working resolution (620x188 for KITTI-00, F3)        */
    int max_slots;     /* resident image pyramids (3 per stream)               */
    int max_jobs;      /* max jobs per batched call (= streams per GPU)        */
    int max_pts;       /* max points per job (tracked + new corners)           */
    int max_corners;   /* max corners returned per GFTT job                    */
    int max_kf;        /* BA: max keyframes per problem                        */
    int max_lm;        /* BA: max landmarks per problem                        */
    int max_obs;       /* BA: max observations (edges) per problem             */
    int max_streams;   /* streams whose last-frame features stay resident in HBM
                          (svslam_rtrack_*); 0 = none                          */
    int device_map;    /* 1: every resident stream also keeps its MAP in HBM (keyframe window of
                          max_kf slots x max_pts features, max_lm landmark slots — a power of two):
                          svslam_dmap_*                                          */
} svslam_limits;

/* ---- lifetime -------------------------------------------------------- */
int  svslam_create(const svslam_limits *lim, svslam_ctx **out);
void svslam_destroy(svslam_ctx *ctx);
const char *svslam_last_error(const svslam_ctx *ctx);
/* version / build info string (arch, compiler) */
const char *svslam_build_info(void);

/* ---- image pyramids ---------------------------------------------------
 * Replaces cv::buildOpticalFlowPyramid as run inside cv::calcOpticalFlowPyrLK
 * (src/frontend.cpp:105-109, 353-357).  Builds the 4-level u8 pyramid of each
 * image into a resident slot; LK and GFTT refer to slots, so the previous
 * frame's pyramid is reused instead of rebuilt.
 * src_is_device: imgs[i] are device pointers already in HBM (bench / pipelined
 * use) instead of host pointers.                                            */
int svslam_pyramid_batch(svslam_ctx *ctx, int n, const int *slots,
                         const void *const *imgs, const int *strides,
                         int src_is_device);
/* Same, but the source is the full-resolution frame and the 1/2 nearest
 * decimation of Dataset::NextFrame (src/dataset.cpp:126-129) is fused into the
 * level-0 write: dst(x,y) = src(2x,2y).  src size = (src_w, src_h); the
 * context's (width,height) must equal (cvRound(src_w/2), cvRound(src_h/2)). */
int svslam_pyramid_decimate_batch(svslam_ctx *ctx, int n, const int *slots,
                                  const void *const *imgs, const int *strides,
                                  int src_w, int src_h, int src_is_device);
/* Declares that every image handed to svslam_pyramid_batch / svslam_track_batch from now on
 * is a full-resolution src_w x src_h frame to be decimated the same way (0,0 switches it
 * off): a caller that keeps the camera frames in HBM never stores the half-size copies the
 * reference makes in Dataset::NextFrame.                                          */
int svslam_set_source_size(svslam_ctx *ctx, int src_w, int src_h);
/* Shape of the latency-bound kernels.  0 (default): least total work per job, for batches of
 * hundreds of streams.  1: a few streams per launch (one camera per GPU, BASELINE config 4) —
 * the pose-only LM of EstimateCurrentPose (src/frontend.cpp:394-558) runs on four wavefronts per
 * job instead of one (about half the latency per frame, ~1.3x the arithmetic).  The two shapes
 * sum the normal equations in different orders: results agree to rounding, each is deterministic.
 * In this mode the blocking calls also wait inside the HIP runtime instead of sleep-polling.
 * A local BA of at most a few problems (Backend::Optimize, src/backend.cpp:22-164) is then dealt over 4 / 8 / 16 workgroups per
 * problem that meet at in-launch barriers.  Those need every workgroup resident at once, so the mode sizes itself from the
 * device: shards x problems <= co-resident solver workgroups per CU x CUs (SVSLAM_LL_CUS=<n> tells it a smaller CU count, for
 * a CU-masked process or a partition); calls with more problems — or a device without room for one — use one workgroup per
 * problem, and a problem whose shards still do not all arrive (the CUs were busy with other work) is solved again that way
 * inside the same call.  Restriction: the solver scratch is one per context — svslam_local_ba_submit / _batch are refused while
 * a deferred local BA of the device map (svslam_dmap_params::ba_defer) is in flight on the context.                          */
int svslam_set_low_latency(svslam_ctx *ctx, int on);
/* Parameter tolerance of the pose-only LM (svslam_pose_only_batch, svslam_track_batch, svslam_rtrack_batch; default 1e-12,
 * environment SVSLAM_PO_XTOL at svslam_create).  A round of EstimateCurrentPose (src/frontend.cpp:482-493: optimize(10)) ends
 * when the first trial of an LM iteration, taken at a damping not above g2o's initial one for the present system (1e-5 x the
 * largest diagonal entry of H), asks for a step whose six
 * components are all <= xtol (metres / radians): the round stands at a stationary point of its cost.  g2o has no such test;
 * it runs the ten iterations, and where a round has converged earlier it spends the rest on trials that move the pose by
 * rounding noise and are accepted or rejected by the sign of that noise.  The trials that are run are g2o's, bit for bit;
 * what the rule leaves out moves the pose by about xtol.  At the default that is what another summation order moves, and
 * every comparison with the oracle holds unchanged; 1e-9 ends the slowly (linearly) converging Huber rounds three
 * iterations earlier (-20 % of the kernel for a tracked frame) at the price that two runs which differ in it part ways as
 * early as runs of different kernel shapes do (LK rounds its initial guesses to f32): DESIGN 4.4 has the table.
 * xtol = 0: g2o's schedule to the last trial.  0 <= xtol <= 1e-6.  Local BA is not touched: its ten iterations all move
 * the window.                                                                                                           */
int svslam_set_pose_only_xtol(svslam_ctx *ctx, double xtol);
/* the tolerance the context actually uses (the default, svslam_set_pose_only_xtol or the SVSLAM_PO_XTOL environment variable
 * read by svslam_create — which fails on a value outside [0, 1e-6] instead of ignoring it); -1 for a null context */
double svslam_get_pose_only_xtol(const svslam_ctx *ctx);
/* test hook: read one level back (tight rows of *w bytes) */
int svslam_pyramid_read(svslam_ctx *ctx, int slot, int level, uint8_t *out,
                        int *w, int *h);

/* test hook: the level with its stored 16-pixel BORDER_REFLECT_101 continuation (what LK windows, pyrDown taps
 * and the GFTT stencils read without index arithmetic): (h + 32) tight rows of (w + 32) bytes */
int svslam_pyramid_read_padded(svslam_ctx *ctx, int slot, int level, uint8_t *out);

/* ---- pyramidal LK -----------------------------------------------------
 * Replaces cv::calcOpticalFlowPyrLK(prev, next, prevPts, nextPts, status, err,
 * Size(11,11), 3, TermCriteria(COUNT+EPS,30,0.01), OPTFLOW_USE_INITIAL_FLOW)
 * at src/frontend.cpp:353-357 (TrackLastFrame) and :105-109
 * (FindFeaturesInRight).  next_xy holds the initial guess on entry.          */
typedef struct svslam_lk_job {
    int prev_slot, next_slot; /* pyramids built by svslam_pyramid_batch      */
    int pt_ofs, npts;         /* range in the concatenated point arrays      */
} svslam_lk_job;

typedef struct svslam_lk_params {
    int    max_level;   /* 3                                                 */
    int    max_iter;    /* 30 (clamped to [0,100] like OpenCV)               */
    double epsilon;     /* 0.01 (squared internally like OpenCV)             */
    double min_eig_thr; /* 1e-4 (OpenCV default)                             */
    int    use_initial_flow; /* 1                                            */
} svslam_lk_params;

int svslam_lk_batch(svslam_ctx *ctx, int njobs, const svslam_lk_job *jobs,
                    int total_pts, const float *prev_xy, float *next_xy,
                    uint8_t *status, float *err, const svslam_lk_params *p);

/* ---- GFTT (Shi-Tomasi) --------------------------------------------------
 * Replaces the mask construction + cv::GFTTDetector::detect at
 * src/frontend.cpp:42-51 (detector created at :24 with
 * (num_features, 0.01, 20)).  rect_xy are the positions of the existing left
 * features; each masks the inclusive square [round(pt-10), round(pt+10)].
 * Output: up to max_corners integer-valued corners per job, quality-descending,
 * out_xy laid out [njobs][max_corners][2].                                  */
typedef struct svslam_gftt_job {
    int slot;              /* level 0 of this pyramid slot is the image      */
    int rect_ofs, nrect;   /* range in rect_xy                               */
} svslam_gftt_job;

int svslam_gftt_batch(svslam_ctx *ctx, int njobs, const svslam_gftt_job *jobs,
                      int total_rects, const float *rect_xy, int max_corners,
                      double quality, double min_dist, float *out_xy,
                      int *out_n);
/* test hook: min-eigenvalue map of a slot's level-0 image (w*h floats) */
int svslam_gftt_eigmap(svslam_ctx *ctx, int slot, float *out);

/* ---- stereo triangulation ----------------------------------------------
 * Replaces slam::triangulation() (include/StereoVisionSLAM/algorithm.h:10-87)
 * as called from BuildInitMap (src/frontend.cpp:165-174) and
 * TriangulateNewPoints (:277-295): pixel2camera on both pixels, 4x4 DLT + SVD,
 * gate sv[3]/sv[2] < 1e-2 && z > 0 && (zmax <= 0 || z <= zmax), then
 * p_world = T_wc * p.                                                        */
typedef struct svslam_tri_job {
    int    pt_ofs, npts;
    double T_wc[7];     /* current_frame->Pose().inverse(); identity at init */
    double zmax;        /* max_triangulation_depth, <=0: no upper gate       */
} svslam_tri_job;

int svslam_triangulate_batch(svslam_ctx *ctx, int njobs,
                             const svslam_tri_job *jobs, int total_pts,
                             const double cam_l[4], const double ext_l[7],
                             const double cam_r[4], const double ext_r[7],
                             const float *uv_l, const float *uv_r,
                             double *out_xyz, uint8_t *out_ok);

/* ---- pose-only refinement ------------------------------------------------
 * Replaces the g2o problem of Frontend::EstimateCurrentPose
 * (src/frontend.cpp:394-558; VertexPose + EdgeProjectionPoseOnly,
 * include/StereoVisionSLAM/g2o_types.h:25-65,94-174): 4 rounds x optimize(10)
 * of Levenberg-Marquardt, estimate reset to the prior each round, chi2 > 5.991
 * => outlier (excluded next round), Huber(1) in rounds 0-2.                  */
typedef struct svslam_pose_job {
    int    pt_ofs, npts;
    double pose[7];     /* in: prior T_cw; out: refined                      */
    int    n_inlier;    /* out                                                */
    int    reserved;
} svslam_pose_job;

int svslam_pose_only_batch(svslam_ctx *ctx, int njobs, svslam_pose_job *jobs,
                           int total_pts, const double cam[4],
                           const double *xyz, const float *uv,
                           uint8_t *outlier, double chi2_th, int rounds,
                           int iters);

/* ---- local bundle adjustment ----------------------------------------------
 * Replaces optimizer.initializeOptimization(); optimizer.optimize(10) of
 * Backend::Optimize (src/backend.cpp:22-164; VertexPose, VertexXYZ
 * marginalised, EdgeProjection with Huber(delta = chi2_th),
 * g2o_types.h:67-92,176-229): LM with Schur complement over the landmarks,
 * dense solve of the reduced camera system, no vertex fixed.  Returns the
 * optimised poses / points and the per-edge chi2 the caller thresholds
 * (src/backend.cpp:167-213).  obs_kf / obs_lm are indices local to the job.  */
typedef struct svslam_ba_job {
    int kf_ofs, nkf;
    int lm_ofs, nlm;
    int obs_ofs, nobs;
    int iters_done;     /* out: LM iterations executed                       */
    int reserved;       /* out: accounting — (LM trials << 24) | block pairs of the Schur complement */
} svslam_ba_job;

int svslam_local_ba_batch(svslam_ctx *ctx, int njobs, svslam_ba_job *jobs,
                          const double cam_l[4], const double ext_l[7],
                          const double cam_r[4], const double ext_r[7],
                          int total_kf, double *poses, int total_lm,
                          double *pts, int total_obs, const int *obs_kf,
                          const int *obs_lm, const uint8_t *obs_is_right,
                          const float *obs_uv, double huber_delta, int iters,
                          double *edge_chi2);

/* The same call split in two, for a backend that runs beside the frontend the way
 * the reference's Backend thread does (src/backend.cpp:345-367, BackendLoop woken by
 * UpdateMap): submit copies the inputs, enqueues the solve on the context's stream
 * and returns; collect waits for it and writes poses / pts / edge_chi2 /
 * jobs[i].iters_done.  Between the two the context's staging memory belongs to the
 * batch: every other batched call on the same context fails until collect (use a
 * second context for the frontend).  collect takes the same njobs / totals.    */
int svslam_local_ba_submit(svslam_ctx *ctx, int njobs, const svslam_ba_job *jobs,
                           const double cam_l[4], const double ext_l[7],
                           const double cam_r[4], const double ext_r[7],
                           int total_kf, const double *poses, int total_lm,
                           const double *pts, int total_obs, const int *obs_kf,
                           const int *obs_lm, const uint8_t *obs_is_right,
                           const float *obs_uv, double huber_delta, int iters);
int svslam_local_ba_collect(svslam_ctx *ctx, int njobs, svslam_ba_job *jobs,
                            int total_kf, double *poses, int total_lm, double *pts,
                            int total_obs, double *edge_chi2);

/* ---- shared-map bundle adjustment (BASELINE config 5; not in the reference) -----------
 * ONE local-BA problem whose landmarks are sharded over the GPUs of a node: every rank holds all
 * nkf poses and its landmarks with their edges.  The rank opens its shard, then runs the pieces of
 * an LM trial; the host sums what the pieces hand out over the ranks (one RCCL all-reduce of
 * (6 nkf)^2 + 2 (6 nkf) + 1 doubles per trial — 3 781 at nkf = 10) and feeds the reduced camera
 * system back.  The LM control flow (g2o's, exactly as in svslam_local_ba_batch) lives in the
 * caller: stereovision-slam_amd/shared_ba.py.  io layout (svslam_sba_io_doubles(nkf) doubles):
 *   S[np*np] | bs[np] | bp[np] | diag(Hpp)[np] | scalars[8],  np = 6 nkf
 *   scalars: 0 chi2, 1 largest landmark diagonal, 2 Cholesky ok, 3 rho denominator (landmarks),
 *            4 rho denominator (poses, identical on every rank), 5 chi2 of the trial state
 * phase 1: diag(Hpp) and scalar 1 out.  phase 2 (lambda): S (without lambda I), bs, bp, scalar 0 out.
 * phase 3 (lambda): reduced S (with lambda I), bs, bp in; scalars 2-5 out; state updated.
 * phase 4: reject the trial (restore).  phase 5: finalise (per-edge chi2).                     */
int svslam_sba_io_doubles(int nkf);
int svslam_sba_open(svslam_ctx *ctx, const double cam_l[4], const double ext_l[7],
                    const double cam_r[4], const double ext_r[7], int nkf, const double *poses,
                    int nlm, const double *pts, int nobs, const int *obs_kf, const int *obs_lm,
                    const uint8_t *obs_is_right, const float *obs_uv, double huber_delta);
int svslam_sba_phase(svslam_ctx *ctx, int phase, double lambda, double *io);
int svslam_sba_close(svslam_ctx *ctx, double *poses, double *pts, double *edge_chi2);
/* The same optimisation as ONE call: the LM control flow runs inside the library, the reduced camera system is
 * all-reduced in place on the device buffer with RCCL (ncclAllReduce on the context's stream; librccl.so is bound
 * with dlopen at the first svslam_sba_comm_* call, nothing links it otherwise), two launches and one 64-byte
 * read-back per LM trial.  Communicator: rank 0 creates the 128-byte id, every rank receives it over any channel
 * and calls svslam_sba_comm_init; without a communicator svslam_sba_solve runs the single-rank problem.
 * trace (optional): 6 doubles per LM trial as svslam_lm_trace.  stats (optional, 4 doubles): trials, total ms,
 * ms per trial, bytes all-reduced per trial.                                                                 */
int svslam_device_count(void);      /* HIP devices visible to this process */
int svslam_sba_comm_unique_id(char out128[128]);
int svslam_sba_comm_init(svslam_ctx *ctx, int nranks, int rank, const char id128[128]);
int svslam_sba_comm_destroy(svslam_ctx *ctx);
int svslam_sba_solve(svslam_ctx *ctx, int iters, int *iters_done, double *lambda_out, double *trace, int trace_cap,
                     int *n_trace, double *stats);

/* test hook: accumulated per-phase ticks (wall_clock64, 100 MHz) of BA job 0;
 * out12[11] = LM trials.  enable=1 allocates the counters, out12 != NULL reads
 * and clears them.                                                            */
int svslam_ba_profile(svslam_ctx *ctx, int enable, long long *out12);

/* test hook: the Levenberg-Marquardt trajectory of the last svslam_pose_only_batch / svslam_local_ba_batch call
 * on this context — one record of 6 doubles per LM trial, rejected trials included: iteration (pose-only:
 * 16 round + iteration), lambda of the trial, chi2 before, chi2 of the trial state, rho, accepted.  enable = 1
 * allocates the device buffer (kernels record while it exists), 0 frees it; out != NULL reads job `job`.    */
int svslam_lm_trace(svslam_ctx *ctx, int enable, int job, double *out, int cap_records, int *n_records);

/* ---- fused per-frame tracking (pyramid + LK + pose-only, one submission) --
 * The whole data-parallel part of Frontend::Track (src/frontend.cpp:645-663)
 * without a host round trip between TrackLastFrame and EstimateCurrentPose:
 * points that fail LK / leave the image are dropped on the device, the rest
 * that carry a map point (has_mp) become pose-only edges.                    */
typedef struct svslam_track_job {
    int    prev_slot, next_slot;
    int    pt_ofs, npts;
    double pose[7];     /* in: prior; out: refined                           */
    int    n_tracked;   /* out: status && in image                           */
    int    n_inlier;    /* out                                                */
} svslam_track_job;

int svslam_track_batch(svslam_ctx *ctx, int njobs, svslam_track_job *jobs,
                       const void *const *next_imgs, const int *strides,
                       int src_is_device, int total_pts, const double cam[4],
                       const float *prev_xy, float *next_xy,
                       const uint8_t *has_mp, const double *xyz,
                       uint8_t *status, uint8_t *outlier,
                       const svslam_lk_params *p, double chi2_th);

/* ---- the same with the features of the last frame resident in HBM ------------
 * svslam_track_batch makes the caller gather (position, map point, start guess) of
 * every feature of the last frame and scatter the survivors into the new frame
 * (src/frontend.cpp:331-347, :361-381, :546-553) — per-feature host work on every
 * frame.  Here the feature list of every stream's last frame lives in HBM: the call
 * reads it, tracks, optimises the pose and leaves the survivors (outlier edges
 * stripped of their map point) as the new list.  The caller only supplies the
 * predicted pose and gets counts back; the compacted survivors (xy, map point id) are
 * also returned for the frames the caller turns into keyframes.  After a keyframe has
 * added features / map points or the backend has moved them, the caller replaces the
 * stream's list with svslam_rtrack_upload.                                        */
typedef struct svslam_rtrack_job {
    int    stream;       /* resident slot, 0 .. max_streams-1                     */
    int    prev_slot, next_slot;
    int    pt_ofs, npts; /* npts = features of the last frame (the caller's count); pt_ofs:
                            where this job's survivors start in out_xy / out_mp   */
    double pose[7];      /* in: predicted T_cw; out: refined                      */
    double T_cam_w[7];   /* in: cam_left.pose * predicted T_cw (src/camera.cpp:74) */
    int    n_tracked;    /* out: survivors = features of the new frame            */
    int    n_edges;      /* out: survivors that carry a map point                 */
    int    n_outlier;    /* out: of those, classified outlier by the pose solve   */
    int    reserved;
} svslam_rtrack_job;

int svslam_rtrack_batch(svslam_ctx *ctx, int njobs, svslam_rtrack_job *jobs,
                        const void *const *next_imgs, const int *strides,
                        int src_is_device, int total_pts, const double cam[4],
                        float *out_xy, int *out_mp,
                        const svslam_lk_params *p, double chi2_th);
/* replaces the resident feature lists of n streams: stream i has counts[i] features at
 * xy/mp/xyz + ofs[i] (xyz is read only where mp >= 0)                               */
int svslam_rtrack_upload(svslam_ctx *ctx, int n, const int *streams, const int *ofs,
                         const int *counts, const float *xy, const int *mp,
                         const double *xyz);

/* ---- the map resident in HBM: the keyframe path without per-feature host work ------------------
 * With limits.device_map = 1 every resident stream (svslam_rtrack_*) also keeps its keyframe window, the features
 * of those keyframes, its landmarks and their observation counts on the device.  One call then runs everything
 * the reference does when a frame becomes a keyframe — Map::InsertKeyFrame / RemoveOldKeyframe / CleanMap
 * (src/map.cpp:53-181), SetObservationsForKeyFrame, DetectFeatures, FindFeaturesInRight, TriangulateNewPoints
 * (src/frontend.cpp:36-141, 251-320, 560-616; StereoInit / BuildInitMap :143-249 with is_init), Backend::Optimize
 * with its outlier handling and write-back (src/backend.cpp:22-246) — and leaves the keyframe's features as the
 * list the next frame tracks from.  The host supplies what is O(window): ids, the slot of the new keyframe and
 * the slot of the keyframe to retire (-1: none; the se3-log distance rule of RemoveOldKeyframe stays on the host),
 * and reads counts and the window's poses back.  mp values in the resident lists are landmark SLOTS here.       */
typedef struct svslam_dmap_job {
    int    stream, slot_cur, slot_right, is_init;  /* is_init: 0 keyframe of a tracked frame, 1 StereoInit, 2 no keyframe at all —
                                                      one local BA over the stream's window as it is (Backend::UpdateMap called from
                                                      outside the frontend, include/StereoVisionSLAM/backend.h:30): only stream, npts and
                                                      the BA outputs are used; such jobs do not share a call with keyframe jobs */
    int    kf_slot, remove_slot, kf_id, npts;   /* npts: features the frame has (tracked survivors); 0 at init   */
    long long frame_id;
    double pose[7];        /* in: T_cw (identity at init); out: after the local BA                              */
    double T_camr_w[7];    /* cam_right.pose * T_cw (src/camera.cpp:74)                                          */
    double T_wc[7];        /* inverse of pose (identity at init)                                                 */
    int    src_buf, dst_buf; /* filled by the library                                                            */
    int    stamp, corners_dropped; /* stamp: filled by the library; corners_dropped: out, surplus corners not appended (max_pts) */
    /* out */
    int    ok;             /* init: enough stereo matches (otherwise nothing was created); keyframe: 1           */
    int    n_features, n_corners, n_right_ok, n_tri_in, n_tri_ok;
    int    ba_nkf, ba_nlm, ba_nobs, ba_iters;
    int    flags;          /* 1 corners dropped (max_pts), 2 landmark slots exhausted, 4 BA skipped (max_obs)    */
    int    dead;
    int    ba_npair, ba_ntrial; /* accounting: (pose, landmark) block pairs of the Schur complement, LM trials (accepted + rejected) */
    int    ev_ofs, ev_n;   /* the landmarks this job freed: records [ev_ofs, ev_ofs + ev_n) of svslam_dmap_evicted()    */
    double win_pose[12][7];/* poses of the BA problem's keyframes after the solve ...                            */
    int    win_slot[12];   /* ... and their slots                                                                */
} svslam_dmap_job;

typedef struct svslam_dmap_params {
    int    num_features, num_features_init, num_active_keyframes, ba_iters;
    double max_triangulation_depth, chi2_th;
    int    ba_defer;       /* 1: the local BA of this call runs BESIDE what follows, on a second stream of the context (the
                              reference's backend thread, src/backend.cpp:250-287): the call returns once the keyframe is in
                              the map and its problem gathered; svslam_dmap_ba_collect applies the result — window poses,
                              landmark positions, outlier observations — later.  One deferred batch at a time. */
    int    reserved;
} svslam_dmap_params;
/* Applies the local BA a svslam_dmap_keyframe_batch call with ba_defer = 1 left running: waits for it, writes poses /
 * positions back into the device map, removes the outlier observations (src/backend.cpp:167-246) and refreshes the positions
 * in every job's resident feature list.  jobs_out (njobs as in that call, same order) receives the jobs with their BA
 * outputs (ba_*, win_pose / win_slot, pose).  Returns 0 and touches nothing when no batch is in flight (*njobs_inflight = 0). */
int svslam_dmap_ba_collect(svslam_ctx *ctx, int njobs, svslam_dmap_job *jobs_out, int *njobs_inflight);

int svslam_dmap_keyframe_batch(svslam_ctx *ctx, int njobs, svslam_dmap_job *jobs,
                               const void *const *left_imgs, const void *const *right_imgs, const int *strides,
                               int src_is_device, const double cam_l[4], const double ext_l[7],
                               const double cam_r[4], const double ext_r[7], const svslam_dmap_params *p);
/* The landmarks the LAST svslam_dmap_keyframe_batch freed from the device map (no observation left, outside the window,
 * not carried by tracking): id and last position.  Map::landmarks_ of the reference keeps every landmark for
 * saveSLAMOutputInFile (src/map.cpp:39-51, src/visual_odometry.cpp:226-304); a caller that writes landmarks.pcd archives
 * these.  *recs points into the context (valid until the next batch call); a job's records are unordered.             */
typedef struct svslam_dmap_evicted_rec { int id; float pos[3]; } svslam_dmap_evicted_rec;
int svslam_dmap_evicted(svslam_ctx *ctx, const svslam_dmap_evicted_rec **recs, int *n);
/* test / writer hook: one stream's window (max_kf entries; kf_frame < 0 = empty slot) and landmark arena
 * (max_lm entries; lm_id < 0 = free slot; lm_state 1 = active, 2 = outside the window)                           */
int svslam_dmap_read(svslam_ctx *ctx, int stream, long long *kf_frame, int *kf_id, double *kf_pose, int *kf_n,
                     int *lm_id, double *lm_pos, int *lm_obs, uint8_t *lm_state);

/* host threads the library may use to prepare a batched call (per-problem BA structure
 * building); default 1.                                                        */
int svslam_set_host_threads(svslam_ctx *ctx, int n);
/* test hooks */
int svslam_debug_host_ns(svslam_ctx *ctx, long long *out8);
/* test hook: shard descriptors of the last low-latency local-BA call, 8 ints per shard (landmarks, edges, blocks, tiles,
 * solver: 2 = resident kernel / 1 = streaming kernel, active poses, mask of shards with edges, iterations) for `nproblems` problems */
int svslam_debug_ll_shards(svslam_ctx *ctx, int nproblems, int *out8, int *shards_per_problem);
/* test hook: the low-latency solver's residency guard (svslam_set_low_latency): shards per problem (0 = batch solver only),
 * problems one call may hand to it, CUs counted, co-resident solver workgroups per CU                                      */
int svslam_debug_ll_limits(svslam_ctx *ctx, int *out4);
int svslam_debug_clock_mhz(svslam_ctx *ctx, int blocks, double ms, double *mhz);
/* test hook: `ncus` workgroups that each take one CU's whole LDS and spin for `ms` milliseconds, enqueued on the context's stream
 * (asynchronous; svslam_sync waits): those CUs cannot take a workgroup that needs LDS meanwhile.  What it shows
 * (tests/test_gpu_low_latency_pipeline.py): a launch of ANOTHER context whose workgroups do not all find a CU cannot
 * retire before the holders leave, whichever solver it uses — the give-up of the low-latency BA (SVSLAM_LL_TIMEOUT_US) bounds the
 * wait of partly resident problems on EACH OTHER, not the wait for foreign kernels                                             */
int svslam_debug_hold_cus(svslam_ctx *ctx, int ncus, double ms);

/* ---- device memory helpers for HBM-resident inputs (bench, pipelining) --- */
int svslam_dev_alloc(svslam_ctx *ctx, size_t bytes, void **out);
int svslam_dev_free(svslam_ctx *ctx, void *p);
int svslam_dev_upload(svslam_ctx *ctx, void *dst, const void *src, size_t bytes);
int svslam_dev_download(svslam_ctx *ctx, void *dst, const void *src, size_t bytes);
int svslam_sync(svslam_ctx *ctx);

/* ---- kernel timing (HIP events on the context's stream) -------------------
 * Accumulated per kernel family since the last reset; used by bench.py for
 * the roofline entry.  family: 0 pyramid, 1 lk, 2 gftt, 3 triangulate,
 * 4 pose_only, 5 local_ba.                                                   */
int svslam_timing_enable(svslam_ctx *ctx, int on);
int svslam_timing_reset(svslam_ctx *ctx);
int svslam_timing_get(svslam_ctx *ctx, int family, double *total_ms,
                      long long *launches, long long *units);

#ifdef __cplusplus
}
#endif
#endif /* SVSLAM_H */
