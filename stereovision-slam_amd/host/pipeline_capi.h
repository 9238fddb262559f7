// pipeline_capi.h — C entry points of the host pipeline, shared by the product
// build (pipeline_capi.cpp, bound to libsvslam_hip.so) and by the CPU twin that
// the oracle directory builds for tests / the cpu_baseline leg.
#pragma once
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct svs_pipe_config {
    int num_features, num_features_init, num_features_tracking, num_features_tracking_bad;
    int num_features_needed_for_keyframe;
    double max_triangulation_depth;
    int num_active_keyframes, backend_on;   /* backend_on: 0 off, 1 BA completes before the next frame,
                                               2 BA runs beside the next frame, result lands one frame late */
    double chi2_th;
    int width, height;
    double cam_l[4], ext_l[7], cam_r[4], ext_r[7];
    int max_lm, max_obs;      /* BA limits per problem */
    int host_threads;         /* threads for the per-stream host bookkeeping (>=1) */
    int src_width, src_height; /* > 0: the frames handed in are full-resolution src_width x src_height and
                                  are decimated 2:1 to width x height on the way into the pyramid (the
                                  resize of Dataset::NextFrame, src/dataset.cpp:126-129, fused)        */
    int resident_track;       /* 1: the features of every stream's last frame stay in device memory
                                  (svslam_rtrack_*); used when backend_on <= 1                      */
    int low_latency;          /* 1: latency shape of the serial kernels (svslam_set_low_latency), for a
                                  few streams per GPU                                                */
    int max_pts;              /* features per frame the kernel provider holds per stream (0 = 512)   */
    int device_map;           /* 1: the map of every stream lives in device memory (svslam_dmap_*): the
                                  keyframe path costs the host O(window) per keyframe, no per-feature work.
                                  Needs resident_track = 1, backend_on = 1.  The HIP provider only.          */
    int backend_lag;          /* backend_on == 2: frames a submitted local BA may stay in flight (0 / 1: one frame) */
} svs_pipe_config;

typedef struct svs_frame_result {
    double pose[7];
    int status, is_keyframe, n_features, n_inliers;
    long long frame_id, keyframe_id;
} svs_frame_result;

typedef struct svs_pipe_counters {
    long long frames, keyframes, track_pts, pose_edges, gftt_calls, gftt_rects, corners, right_pts, tri_pts;
    long long ba_calls, ba_edges, ba_kf, ba_lm, ba_iters, pyr_left, pyr_right;
    long long ns_step, ns_kernel_calls;
    long long corners_dropped, ba_skipped;   /* per-stream capacity events (max_pts / max_lm / max_obs) */
    long long ba_pairs, ba_trials;           /* block pairs of the Schur complements, LM trials: the flop accounting of bench.py */
    long long lm_total, lm_resident;         /* landmarks ever created / MapPoint objects the host still holds (the rest
                                                were evicted to the 16-byte archive, Map::ReleaseRetired)              */
    long long lm_full;                       /* device map: keyframes that ran out of landmark slots (max_lm = LIVE landmarks of a
                                                stream there); their surplus points were not created                     */
} svs_pipe_counters;

void *svs_pipe_create(const svs_pipe_config *cfg, int nstreams, int device);
void svs_pipe_destroy(void *p);
const char *svs_pipe_last_error(void);
/* Process-wide glibc malloc settings for a host that drives thousands of streams: the per-frame
 * Frame / feature-list churn otherwise makes every thread's heap top shrink and regrow (munmap /
 * mmap / page faults, all under the process' mm lock).  Never trims, serves large vectors from the
 * heaps, grows them in 64 MB steps.  Optional; call once before creating pipelines.            */
void svs_pipe_tune_allocator(void);
/* completes a backend optimisation still in flight (backend_on 2) */
int svs_pipe_flush(void *p);
/* one frame for every stream; left/right: nstreams image pointers (host or device) */
int svs_pipe_step(void *p, const void *const *left, const void *const *right, int is_device,
                  svs_frame_result *out);
/* run `nframes` steps over device-resident sequences laid out [stream][frame][h*w]
 * (left and right), writing results [frame][stream]; the whole loop stays in C++ */
int svs_pipe_run_device(void *p, const void *left_base, const void *right_base, long long stream_stride,
                        long long frame_stride, int first_frame, int nframes, svs_frame_result *out);
int svs_pipe_counters_get(void *p, svs_pipe_counters *out);
/* keyframes.txt + landmarks.pcd of one stream, in the reference's formats (src/visual_odometry.cpp:198-310) */
int svs_pipe_save_outputs(void *p, int stream, const char *dir, const char *dataset_dir, int left_cam_index);
/* Inspection hook (host map only): the map of one stream as it stands after the last step, flat.
 *   ints: n_active_keyframes, (keyframe id)*, n_active_landmarks, per landmark { id, observed_times, n_obs,
 *         (keyframe id, 0 left / 1 right, index of the feature in that keyframe's list)* }  — ids ascending, observations in list order
 *   dbl:  7 per active keyframe (pose, T_cw), then 3 per active landmark (position)
 * Returns the number of ints the snapshot has (nothing is written when cap_i or cap_d is too small: call again), -1 for a bad
 * stream, -2 when the map lives on the device (svslam_dmap_read is the reader there), -3 when a BA in flight cannot be completed. */
long long svs_pipe_map_snapshot(void *p, int stream, long long *ints, long long cap_i, double *dbl, long long cap_d);
/* underlying svslam_ctx (product build) or NULL (CPU twin) */
void *svs_pipe_kernel_ctx(void *p);
/* context the backend's local BA runs on: a second one when backend_on == 2, else the same */
void *svs_pipe_backend_ctx(void *p);

#ifdef __cplusplus
}
#endif
