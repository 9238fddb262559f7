"""ctypes binding of the C++ host pipeline (lib/libsvslam_pipeline.so): the
reference's Frontend/Backend/Map logic driving the HIP kernels, S streams in
lockstep.  `lib` may be overridden by tests with the CPU twin built under
oracle/ (same C API, oracle kernels) — the product never does that."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class PipeConfig(C.Structure):
    _fields_ = [("num_features", C.c_int), ("num_features_init", C.c_int), ("num_features_tracking", C.c_int),
                ("num_features_tracking_bad", C.c_int), ("num_features_needed_for_keyframe", C.c_int),
                ("max_triangulation_depth", C.c_double), ("num_active_keyframes", C.c_int), ("backend_on", C.c_int),
                ("chi2_th", C.c_double), ("width", C.c_int), ("height", C.c_int), ("cam_l", C.c_double * 4),
                ("ext_l", C.c_double * 7), ("cam_r", C.c_double * 4), ("ext_r", C.c_double * 7),
                ("max_lm", C.c_int), ("max_obs", C.c_int), ("host_threads", C.c_int),
                ("src_width", C.c_int), ("src_height", C.c_int), ("resident_track", C.c_int),
                ("low_latency", C.c_int), ("max_pts", C.c_int), ("device_map", C.c_int), ("backend_lag", C.c_int)]


class FrameResult(C.Structure):
    _fields_ = [("pose", C.c_double * 7), ("status", C.c_int), ("is_keyframe", C.c_int), ("n_features", C.c_int),
                ("n_inliers", C.c_int), ("frame_id", C.c_longlong), ("keyframe_id", C.c_longlong)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_longlong) for n in
                ("frames", "keyframes", "track_pts", "pose_edges", "gftt_calls", "gftt_rects", "corners", "right_pts",
                 "tri_pts", "ba_calls", "ba_edges", "ba_kf", "ba_lm", "ba_iters", "pyr_left", "pyr_right", "ns_step",
                 "ns_kernel_calls", "corners_dropped", "ba_skipped", "ba_pairs", "ba_trials", "lm_total", "lm_resident", "lm_full")]


RESULT_DTYPE = np.dtype([("pose", np.float64, 7), ("status", np.int32), ("is_keyframe", np.int32),
                         ("n_features", np.int32), ("n_inliers", np.int32), ("frame_id", np.int64),
                         ("keyframe_id", np.int64)])
assert RESULT_DTYPE.itemsize == C.sizeof(FrameResult)


def tune_allocator(lib=None):
    """process-wide malloc settings for many-stream hosts (svs_pipe_tune_allocator)"""
    (_bind(lib) if lib is not None else product_lib()).svs_pipe_tune_allocator()


def default_config(width=620, height=188, cam=(359.428, 359.428, 303.5964, 92.60785), baseline=0.537166, **kw):
    """config/stereo_slam_configs/config-00.yaml of the reference + KITTI-00 halved calibration"""
    c = PipeConfig()
    c.num_features = 150; c.num_features_init = 50; c.num_features_tracking = 50
    c.num_features_tracking_bad = 20; c.num_features_needed_for_keyframe = 80
    c.max_triangulation_depth = 300.0; c.num_active_keyframes = 10; c.backend_on = 1; c.chi2_th = 5.991
    c.width = width; c.height = height
    c.cam_l = (C.c_double * 4)(*cam); c.cam_r = (C.c_double * 4)(*cam)
    c.ext_l = (C.c_double * 7)(0, 0, 0, 1, 0, 0, 0)
    c.ext_r = (C.c_double * 7)(0, 0, 0, 1, -baseline, 0, 0)
    c.max_lm = 4096; c.max_obs = 16384; c.host_threads = 1
    c.src_width = 0; c.src_height = 0; c.resident_track = 1; c.low_latency = 0; c.max_pts = 512; c.device_map = 0; c.backend_lag = 1
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def load_yaml_config(path, **kw):
    """reads the scalar keys of a reference stereo_slam_configs/*.yaml (OpenCV FileStorage subset)"""
    vals = {}
    with open(path) as f:
        for line in f:
            line = line.split("#", 1)[0].strip()
            if not line or line.startswith("%") or ":" not in line:
                continue
            k, v = line.split(":", 1)
            vals[k.strip()] = v.strip()
    c = default_config(**kw)
    for k in ("num_features", "num_features_init", "num_features_tracking", "num_features_tracking_bad",
              "num_features_needed_for_keyframe", "num_active_keyframes", "backend_on"):
        if k in vals:
            setattr(c, k, int(float(vals[k])))
    for k in ("max_triangulation_depth", "chi2_th"):
        if k in vals:
            setattr(c, k, float(vals[k]))
    return c


def _bind(L):
    L.svs_pipe_create.restype = C.c_void_p
    L.svs_pipe_create.argtypes = [C.POINTER(PipeConfig), C.c_int, C.c_int]
    L.svs_pipe_destroy.argtypes = [C.c_void_p]
    L.svs_pipe_destroy.restype = None
    L.svs_pipe_last_error.restype = C.c_char_p
    L.svs_pipe_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.svs_pipe_run_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong, C.c_int,
                                      C.c_int, C.c_void_p]
    L.svs_pipe_counters_get.argtypes = [C.c_void_p, C.POINTER(Counters)]
    L.svs_pipe_save_outputs.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int]
    L.svs_pipe_flush.argtypes = [C.c_void_p]
    L.svs_pipe_map_snapshot.restype = C.c_longlong
    L.svs_pipe_map_snapshot.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong]
    L.svs_pipe_tune_allocator.restype = None
    L.svs_pipe_backend_ctx.restype = C.c_void_p
    L.svs_pipe_backend_ctx.argtypes = [C.c_void_p]
    L.svs_pipe_kernel_ctx.restype = C.c_void_p
    L.svs_pipe_kernel_ctx.argtypes = [C.c_void_p]
    return L


_product = None


def product_lib():
    global _product
    if _product is None:
        # always the product library next to this file (no environment override: the CPU twin under oracle/ has the same C
        # API and must never be loadable as the product; A/B scripts pass `lib=` to Pipeline explicitly)
        p = os.path.join(HERE, "lib", "libsvslam_pipeline.so")
        if not os.path.exists(p):
            raise RuntimeError("libsvslam_pipeline.so is missing: run stereovision-slam_amd/build.py")
        _product = _bind(C.CDLL(p))
    return _product


class Pipeline:
    def __init__(self, cfg=None, nstreams=1, device=0, lib=None):
        self.L = _bind(lib) if lib is not None else product_lib()
        self.cfg = cfg or default_config()
        self.n = nstreams
        self.h = self.L.svs_pipe_create(C.byref(self.cfg), nstreams, device)
        if not self.h:
            raise RuntimeError("svs_pipe_create failed: " + self.L.svs_pipe_last_error().decode())

    def close(self):
        if self.h:
            self.L.svs_pipe_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self, lefts, rights, device=False):
        """one frame per stream; lefts/rights: lists of np.uint8 images (or device pointers)"""
        n = self.n
        if device:
            lp = (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in lefts])
            rp = (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in rights])
        else:
            keep = [np.ascontiguousarray(a, np.uint8) for a in list(lefts) + list(rights)]
            lp = (C.c_void_p * n)(*[a.ctypes.data for a in keep[:n]])
            rp = (C.c_void_p * n)(*[a.ctypes.data for a in keep[n:]])
        out = np.zeros(n, RESULT_DTYPE)
        rc = self.L.svs_pipe_step(self.h, lp, rp, 1 if device else 0, out.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise RuntimeError("svs_pipe_step failed: " + self.L.svs_pipe_last_error().decode())
        return out

    def run_device(self, left_base, right_base, stream_stride, frame_stride, first_frame, nframes, want_results=True, out=None):
        """out: a caller-owned [nframes, n] RESULT_DTYPE array to fill (e.g. allocated and touched ahead of a measurement)"""
        if out is not None:
            assert out.shape == (nframes, self.n) and out.dtype == RESULT_DTYPE and out.flags.c_contiguous
            want_results = True
        elif want_results:
            out = np.zeros((nframes, self.n), RESULT_DTYPE)
        rc = self.L.svs_pipe_run_device(self.h, C.c_void_p(left_base), C.c_void_p(right_base), stream_stride,
                                        frame_stride, first_frame, nframes,
                                        out.ctypes.data_as(C.c_void_p) if want_results else None)
        if rc != 0:
            raise RuntimeError("svs_pipe_run_device failed: " + self.L.svs_pipe_last_error().decode())
        return out

    def flush(self):
        """complete a backend optimisation that is still in flight (backend_on == 2)"""
        if self.L.svs_pipe_flush(self.h) != 0:
            raise RuntimeError("svs_pipe_flush failed: " + self.L.svs_pipe_last_error().decode())

    def counters(self):
        c = Counters()
        self.L.svs_pipe_counters_get(self.h, C.byref(c))
        return {n: getattr(c, n) for n, _ in Counters._fields_}

    def save_outputs(self, stream, out_dir, dataset_dir="./data/dataset/sequences/00", left_cam_index=0):
        """keyframes.txt + landmarks.pcd in the reference's formats"""
        rc = self.L.svs_pipe_save_outputs(self.h, stream, out_dir.encode(), dataset_dir.encode(), left_cam_index)
        if rc != 0:
            raise RuntimeError("svs_pipe_save_outputs failed (%d)" % rc)

    def map_snapshot(self, stream=0):
        """the host map of one stream after the last step (svs_pipe_map_snapshot): active keyframe ids and poses, active landmarks
        with position, observation counter and observation list [(keyframe id, 0 left / 1 right, feature index)]"""
        ni = self.L.svs_pipe_map_snapshot(self.h, stream, None, 0, None, 0)
        if ni < 0:
            raise RuntimeError("svs_pipe_map_snapshot: %d (-2: the map of this pipeline lives on the device)" % ni)
        ints = np.zeros(ni, np.int64)
        dbl = np.zeros(7 * ni, np.float64)
        if self.L.svs_pipe_map_snapshot(self.h, stream, ints.ctypes.data_as(C.c_void_p), ni, dbl.ctypes.data_as(C.c_void_p), dbl.size) != ni:
            raise RuntimeError("svs_pipe_map_snapshot changed size between two calls")
        o = d = 0
        nk = int(ints[o]); o += 1
        kf = [int(v) for v in ints[o:o + nk]]; o += nk
        poses = {k: dbl[d + 7 * i:d + 7 * i + 7].copy() for i, k in enumerate(kf)}; d += 7 * nk
        nl = int(ints[o]); o += 1
        lms = []
        for _ in range(nl):
            mid, times, no = (int(v) for v in ints[o:o + 3]); o += 3
            obs = tuple((int(ints[o + 3 * i]), int(ints[o + 3 * i + 1]), int(ints[o + 3 * i + 2])) for i in range(no)); o += 3 * no
            lms.append((mid, times, obs, tuple(float(v) for v in dbl[d:d + 3]))); d += 3
        return {"active_keyframes": kf, "landmarks": lms, "keyframe_poses": poses}

    def kernel_ctx(self):
        return self.L.svs_pipe_kernel_ctx(self.h)

    def backend_ctx(self):
        return self.L.svs_pipe_backend_ctx(self.h)


# ---- trajectory error (the reference has no evaluator) -----------------
def quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def camera_centres(poses_cw):
    """T_cw poses [n,7] -> camera centres in world [n,3]"""
    out = np.zeros((len(poses_cw), 3))
    for i, T in enumerate(poses_cw):
        out[i] = -quat_to_R(T[:4]).T @ T[4:]
    return out


def ate_rmse(est_cw, gt_cw):
    """RMSE of camera-centre error after rigid (SE3, no scale) alignment (Horn / Kabsch)"""
    a = camera_centres(est_cw); b = camera_centres(gt_cw)
    ma, mb = a.mean(0), b.mean(0)
    Hm = (a - ma).T @ (b - mb)
    U, _, Vt = np.linalg.svd(Hm)
    D = np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))])
    R = Vt.T @ D @ U.T
    al = (a - ma) @ R.T + mb
    return float(np.sqrt(((al - b) ** 2).sum(1).mean()))
