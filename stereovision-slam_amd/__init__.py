"""stereovision-slam_amd — MI355X-native hot path of StereoVision-SLAM.

Thin ctypes binding over the C ABI of lib/libsvslam_hip.so (include/svslam.h).
There is no CPU fallback: importing works anywhere (so the build can be checked
without a GPU), but creating a Context without a usable HIP device raises.
The package directory name contains a hyphen; load it with
``importlib.import_module("stereovision-slam_amd")``.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(HERE, "lib")

ABI_SYMBOLS = [
    "svslam_create", "svslam_destroy", "svslam_last_error", "svslam_build_info",
    "svslam_pyramid_batch", "svslam_pyramid_decimate_batch", "svslam_set_source_size", "svslam_set_low_latency", "svslam_set_pose_only_xtol", "svslam_get_pose_only_xtol", "svslam_pyramid_read",
    "svslam_pyramid_read_padded", "svslam_lk_batch", "svslam_gftt_batch", "svslam_gftt_eigmap", "svslam_triangulate_batch",
    "svslam_pose_only_batch", "svslam_local_ba_batch", "svslam_local_ba_submit", "svslam_local_ba_collect",
    "svslam_track_batch", "svslam_rtrack_batch", "svslam_rtrack_upload",
    "svslam_sba_io_doubles", "svslam_sba_open", "svslam_sba_phase", "svslam_sba_close",
    "svslam_device_count", "svslam_dmap_keyframe_batch", "svslam_dmap_ba_collect", "svslam_dmap_read", "svslam_dmap_evicted", "svslam_sba_comm_unique_id", "svslam_sba_comm_init", "svslam_sba_comm_destroy", "svslam_sba_solve",
    "svslam_dev_alloc", "svslam_dev_free", "svslam_dev_upload", "svslam_dev_download", "svslam_sync",
    "svslam_timing_enable", "svslam_timing_reset", "svslam_timing_get", "svslam_ba_profile",
    "svslam_set_host_threads", "svslam_debug_host_ns", "svslam_debug_ll_shards", "svslam_debug_ll_limits", "svslam_debug_clock_mhz", "svslam_debug_hold_cus", "svslam_lm_trace",
]

FAMILIES = {"pyramid": 0, "lk": 1, "gftt": 2, "triangulate": 3, "pose_only": 4, "local_ba": 5}
DEBUG_FAMILIES = {"dbg0": 6, "dbg1": 7, "dbg2": 8, "dbg3": 9}     # per-kernel split (SVSLAM_TIMING_SPLIT=1), development
# intervals nested inside a family: the local-BA solver kernel alone (k_local_ba_t<0, 1, EID>, or the low-latency solver's kernels)
# inside "local_ba" (= map gather + structure build + solver + scatter)
KERNEL_FAMILIES = {"ba_solve": 10}
# the HIP kernel(s) behind each timing family at the batch operating point (what a rocprofv3 --kernel-trace --stats row is named)
FAMILY_KERNELS = {"pyramid": ["k_pyr_fused"], "lk": ["k_lk"], "gftt": ["k_gftt_eig3", "k_gftt_select2"], "triangulate": ["k_triangulate"],
                  "pose_only": ["k_pose_only"], "local_ba": ["k_dmap_ba_gather", "k_ba_build", "k_local_ba_t", "k_dmap_ba_scatter"],
                  "ba_solve": ["k_local_ba_t"]}


class Limits(C.Structure):
    _fields_ = [("device", C.c_int), ("width", C.c_int), ("height", C.c_int), ("max_slots", C.c_int),
                ("max_jobs", C.c_int), ("max_pts", C.c_int), ("max_corners", C.c_int), ("max_kf", C.c_int),
                ("max_lm", C.c_int), ("max_obs", C.c_int), ("max_streams", C.c_int), ("device_map", C.c_int)]


class RtrackJob(C.Structure):
    _fields_ = [("stream", C.c_int), ("prev_slot", C.c_int), ("next_slot", C.c_int), ("pt_ofs", C.c_int),
                ("npts", C.c_int), ("pose", C.c_double * 7), ("T_cam_w", C.c_double * 7), ("n_tracked", C.c_int),
                ("n_edges", C.c_int), ("n_outlier", C.c_int), ("reserved", C.c_int)]


class LkJob(C.Structure):
    _fields_ = [("prev_slot", C.c_int), ("next_slot", C.c_int), ("pt_ofs", C.c_int), ("npts", C.c_int)]


class LkParams(C.Structure):
    _fields_ = [("max_level", C.c_int), ("max_iter", C.c_int), ("epsilon", C.c_double),
                ("min_eig_thr", C.c_double), ("use_initial_flow", C.c_int)]


class GfttJob(C.Structure):
    _fields_ = [("slot", C.c_int), ("rect_ofs", C.c_int), ("nrect", C.c_int)]


class TriJob(C.Structure):
    _fields_ = [("pt_ofs", C.c_int), ("npts", C.c_int), ("T_wc", C.c_double * 7), ("zmax", C.c_double)]


class PoseJob(C.Structure):
    _fields_ = [("pt_ofs", C.c_int), ("npts", C.c_int), ("pose", C.c_double * 7), ("n_inlier", C.c_int),
                ("reserved", C.c_int)]


class BaJob(C.Structure):
    _fields_ = [("kf_ofs", C.c_int), ("nkf", C.c_int), ("lm_ofs", C.c_int), ("nlm", C.c_int),
                ("obs_ofs", C.c_int), ("nobs", C.c_int), ("iters_done", C.c_int), ("reserved", C.c_int)]


class TrackJob(C.Structure):
    _fields_ = [("prev_slot", C.c_int), ("next_slot", C.c_int), ("pt_ofs", C.c_int), ("npts", C.c_int),
                ("pose", C.c_double * 7), ("n_tracked", C.c_int), ("n_inlier", C.c_int)]


_lib = None


def build(force=False, verbose=False):
    """Compile every native library of the package (hipcc, gfx950)."""
    return _build.build_all(force=force, verbose=verbose)


def lib_path():
    return os.path.join(LIB_DIR, "libsvslam_hip.so")


def load():
    """dlopen libsvslam_hip.so; raises if it has not been built."""
    global _lib
    if _lib is None:
        p = lib_path()
        if not os.path.exists(p):
            raise RuntimeError("libsvslam_hip.so is missing: run stereovision-slam_amd/build.py "
                               "(there is no CPU fallback)")
        L = C.CDLL(p)
        L.svslam_last_error.restype = C.c_char_p
        L.svslam_build_info.restype = C.c_char_p
        L.svslam_last_error.argtypes = [C.c_void_p]
        L.svslam_destroy.argtypes = [C.c_void_p]
        L.svslam_destroy.restype = None
        L.svslam_set_pose_only_xtol.argtypes = [C.c_void_p, C.c_double]
        L.svslam_get_pose_only_xtol.argtypes = [C.c_void_p]
        L.svslam_get_pose_only_xtol.restype = C.c_double
        L.svslam_dev_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.svslam_dev_free.argtypes = [C.c_void_p, C.c_void_p]
        L.svslam_dev_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.svslam_dev_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a, cols):
    return np.ascontiguousarray(a, np.float32).reshape(-1, cols)


def _d(a):
    return np.ascontiguousarray(a, np.float64)


IDENT = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)


class Context:
    """One svslam_ctx: own HIP stream, resident pyramid slots, staging arena."""

    def __init__(self, width, height, max_slots=4, max_jobs=1, max_pts=512, max_corners=150, max_kf=10,
                 max_lm=2048, max_obs=8192, device=0, max_streams=0, device_map=0):
        self.L = load()
        self.lim = Limits(device, width, height, max_slots, max_jobs, max_pts, max_corners, max_kf, max_lm,
                          max_obs, max_streams, device_map)
        self.h = C.c_void_p()
        rc = self.L.svslam_create(C.byref(self.lim), C.byref(self.h))
        if rc != 0:
            msg = self.L.svslam_last_error(self.h).decode() if self.h else "no HIP device / bad limits"
            if self.h:
                self.L.svslam_destroy(self.h)
                self.h = C.c_void_p()
            raise RuntimeError("svslam_create failed (%d): %s" % (rc, msg))
        self.width, self.height = width, height
        self._owned = True

    @classmethod
    def borrow(cls, handle, width, height):
        """wrap an svslam_ctx owned by someone else (e.g. the host pipeline)"""
        self = cls.__new__(cls)
        self.L = load()
        self.h = C.c_void_p(handle)
        self.width, self.height = width, height
        self._owned = False
        return self

    def close(self):
        if self.h and getattr(self, "_owned", False):
            self.L.svslam_destroy(self.h)
        self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.L.svslam_last_error(self.h).decode()))

    # ---- device memory -------------------------------------------------
    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self.L.svslam_dev_alloc(self.h, nbytes, C.byref(p)), "dev_alloc")
        return p.value

    def dev_free(self, ptr):
        self._chk(self.L.svslam_dev_free(self.h, C.c_void_p(ptr)), "dev_free")

    def dev_upload(self, ptr, arr):
        arr = np.ascontiguousarray(arr)
        self._chk(self.L.svslam_dev_upload(self.h, C.c_void_p(ptr), _p(arr), arr.nbytes), "dev_upload")

    def dev_download(self, ptr, arr):
        self._chk(self.L.svslam_dev_download(self.h, _p(arr), C.c_void_p(ptr), arr.nbytes), "dev_download")

    def sync(self):
        self._chk(self.L.svslam_sync(self.h), "sync")

    # ---- timing ----------------------------------------------------------
    def ll_limits(self):
        """test hook (svslam_debug_ll_limits): [shards per problem, problems per call, CUs counted, solver workgroups per CU]"""
        o = (C.c_int * 4)()
        self._chk(self.L.svslam_debug_ll_limits(self.h, o), "debug_ll_limits")
        return [int(v) for v in o]

    def hold_cus(self, ncus, ms):
        """test hook (svslam_debug_hold_cus): ncus workgroups take a CU's LDS each and spin for ms; asynchronous"""
        self._chk(self.L.svslam_debug_hold_cus(self.h, int(ncus), C.c_double(ms)), "debug_hold_cus")

    def clock_mhz(self, blocks=1, ms=1.0):
        """effective shader clock seen by a wave that spins for `ms` of the constant 100 MHz counter (svslam_debug_clock_mhz)"""
        mhz = C.c_double(0)
        self._chk(self.L.svslam_debug_clock_mhz(self.h, blocks, C.c_double(ms), C.byref(mhz)), "debug_clock_mhz")
        return mhz.value

    def host_counters(self):
        """test hook (svslam_debug_host_ns): 8 host-side counters since the last call; [6] = local-BA problems the
        low-latency solver took"""
        o = (C.c_longlong * 8)()
        self._chk(self.L.svslam_debug_host_ns(self.h, o), "debug_host_ns")
        return [int(v) for v in o]

    def ll_shards(self, nproblems=1):
        """test hook (svslam_debug_ll_shards): [nproblems, W, 8] ints of the last low-latency local-BA call — landmarks,
        edges, blocks, tiles, solver (2 resident / 1 streaming), active poses, shard mask, iterations"""
        w = C.c_int(0)
        out = np.zeros((nproblems * 16, 8), np.int32)
        self._chk(self.L.svslam_debug_ll_shards(self.h, nproblems, _p(out), C.byref(w)), "debug_ll_shards")
        return out[:nproblems * w.value].reshape(nproblems, w.value, 8).copy()

    def low_latency(self, on=True):
        """latency shape of the serial kernels (svslam_set_low_latency): a few jobs per launch"""
        self._chk(self.L.svslam_set_low_latency(self.h, 1 if on else 0), "set_low_latency")

    def pose_only_xtol(self, xtol):
        """parameter tolerance of the pose-only LM (svslam_set_pose_only_xtol); 0 = g2o's schedule to the last trial"""
        self._chk(self.L.svslam_set_pose_only_xtol(self.h, C.c_double(xtol)), "set_pose_only_xtol")

    def pose_only_xtol_effective(self):
        """the tolerance the context uses (svslam_get_pose_only_xtol)"""
        return float(self.L.svslam_get_pose_only_xtol(self.h))

    def timing(self, on=True):
        self.L.svslam_timing_enable(self.h, 1 if on else 0)
        self.L.svslam_timing_reset(self.h)

    def timing_get(self, family):
        ms, n, u = C.c_double(), C.c_longlong(), C.c_longlong()
        fam = FAMILIES[family] if family in FAMILIES else (KERNEL_FAMILIES[family] if family in KERNEL_FAMILIES else DEBUG_FAMILIES[family])
        self._chk(self.L.svslam_timing_get(self.h, fam, C.byref(ms), C.byref(n), C.byref(u)), "timing")
        return ms.value, n.value, u.value

    def ba_profile(self, enable=True, read=False):
        out = (C.c_longlong * 12)()
        self._chk(self.L.svslam_ba_profile(self.h, 1 if enable else 0, out if read else None), "ba_profile")
        return list(out)

    def lm_trace(self, enable=True, job=None, cap=408):
        """test hook: record / read the LM trajectory [ntrials, 6] of the last pose-only / local-BA call (svslam_lm_trace)"""
        if job is None:
            self._chk(self.L.svslam_lm_trace(self.h, 1 if enable else 0, 0, None, 0, None), "lm_trace")
            return None
        out = np.zeros((cap, 6)); n = C.c_int(0)
        self._chk(self.L.svslam_lm_trace(self.h, 1, job, _p(out), cap, C.byref(n)), "lm_trace")
        return out[:n.value].copy()

    # ---- pyramids --------------------------------------------------------
    def pyramid(self, slots, imgs, device=False, strides=None, decimate_from=None):
        n = len(slots)
        sl = (C.c_int * n)(*slots)
        keep = []
        if device:
            ptrs = (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in imgs])
            st = (C.c_int * n)(*(strides or [self.width if decimate_from is None else decimate_from[0]] * n))
        else:
            keep = [np.ascontiguousarray(im, np.uint8) for im in imgs]
            ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in keep])
            st = (C.c_int * n)(*[im.shape[1] for im in keep])
        if decimate_from is None:
            rc = self.L.svslam_pyramid_batch(self.h, n, sl, ptrs, st, 1 if device else 0)
        else:
            rc = self.L.svslam_pyramid_decimate_batch(self.h, n, sl, ptrs, st, decimate_from[0], decimate_from[1],
                                                      1 if device else 0)
        self._chk(rc, "pyramid")

    def pyramid_read(self, slot, level):
        w, h = C.c_int(), C.c_int()
        self._chk(self.L.svslam_pyramid_read(self.h, slot, level, None, C.byref(w), C.byref(h)), "pyramid_read")
        out = np.zeros((h.value, w.value), np.uint8)
        self._chk(self.L.svslam_pyramid_read(self.h, slot, level, _p(out), C.byref(w), C.byref(h)), "pyramid_read")
        return out

    def pyramid_read_padded(self, slot, level):
        """the level with its stored 16-px REFLECT_101 border"""
        w, h = C.c_int(), C.c_int()
        self._chk(self.L.svslam_pyramid_read(self.h, slot, level, None, C.byref(w), C.byref(h)), "pyramid_read")
        out = np.zeros((h.value + 32, w.value + 32), np.uint8)
        self._chk(self.L.svslam_pyramid_read_padded(self.h, slot, level, _p(out)), "pyramid_read_padded")
        return out

    # ---- LK --------------------------------------------------------------
    def lk(self, jobs, params=None):
        """jobs: list of (prev_slot, next_slot, prev_xy[n,2], next_xy_guess[n,2]).
        returns list of (next_xy, status, err)."""
        n = len(jobs)
        arr = (LkJob * n)()
        prevs, nexts, ofs = [], [], 0
        for i, (ps, ns, p, q) in enumerate(jobs):
            p = _f32(p, 2); q = _f32(q, 2)
            arr[i] = LkJob(ps, ns, ofs, p.shape[0])
            ofs += p.shape[0]
            prevs.append(p); nexts.append(q)
        prev = np.concatenate(prevs) if prevs else np.zeros((0, 2), np.float32)
        nxt = np.concatenate(nexts).copy() if nexts else np.zeros((0, 2), np.float32)
        prev = np.ascontiguousarray(prev); nxt = np.ascontiguousarray(nxt)
        status = np.zeros(max(ofs, 1), np.uint8)
        err = np.zeros(max(ofs, 1), np.float32)
        prm = params or LkParams(3, 30, 0.01, 1e-4, 1)
        self._chk(self.L.svslam_lk_batch(self.h, n, arr, ofs, _p(prev), _p(nxt), _p(status), _p(err), C.byref(prm)),
                  "lk_batch")
        out = []
        for j in arr:
            s = slice(j.pt_ofs, j.pt_ofs + j.npts)
            out.append((nxt[s].copy(), status[s].copy(), err[s].copy()))
        return out

    # ---- GFTT ------------------------------------------------------------
    def gftt(self, jobs, max_corners=150, quality=0.01, min_dist=20.0):
        """jobs: list of (slot, rect_xy[n,2] or None). returns list of corners[k,2]."""
        n = len(jobs)
        arr = (GfttJob * n)()
        rects, ofs = [], 0
        for i, (slot, r) in enumerate(jobs):
            r = np.zeros((0, 2), np.float32) if r is None else _f32(r, 2)
            arr[i] = GfttJob(slot, ofs, r.shape[0])
            ofs += r.shape[0]
            rects.append(r)
        rect = np.ascontiguousarray(np.concatenate(rects)) if ofs else np.zeros((1, 2), np.float32)
        out = np.zeros((n, max_corners, 2), np.float32)
        cnt = np.zeros(n, np.int32)
        self._chk(self.L.svslam_gftt_batch(self.h, n, arr, ofs, _p(rect), max_corners, C.c_double(quality),
                                           C.c_double(min_dist), _p(out), _p(cnt)), "gftt_batch")
        return [out[i, :cnt[i]].copy() for i in range(n)]

    def gftt_eigmap(self, slot):
        out = np.zeros((self.height, self.width), np.float32)
        self._chk(self.L.svslam_gftt_eigmap(self.h, slot, _p(out)), "gftt_eigmap")
        return out

    # ---- triangulation -----------------------------------------------------
    def triangulate(self, jobs, cam_l, ext_l, cam_r, ext_r):
        """jobs: list of (uv_l[n,2], uv_r[n,2], T_wc[7] or None, zmax). returns list of (xyz, ok)."""
        n = len(jobs)
        arr = (TriJob * n)()
        ls, rs, ofs = [], [], 0
        for i, (ul, ur, T, zmax) in enumerate(jobs):
            ul = _f32(ul, 2); ur = _f32(ur, 2)
            T = IDENT if T is None else _d(T)
            arr[i] = TriJob(ofs, ul.shape[0], (C.c_double * 7)(*T), float(zmax))
            ofs += ul.shape[0]
            ls.append(ul); rs.append(ur)
        if ofs == 0:
            return [(np.zeros((0, 3)), np.zeros(0, np.uint8)) for _ in jobs]
        L_ = np.ascontiguousarray(np.concatenate(ls)); R_ = np.ascontiguousarray(np.concatenate(rs))
        xyz = np.zeros((ofs, 3)); ok = np.zeros(ofs, np.uint8)
        self._chk(self.L.svslam_triangulate_batch(self.h, n, arr, ofs, _p(_d(cam_l)), _p(_d(ext_l)), _p(_d(cam_r)),
                                                  _p(_d(ext_r)), _p(L_), _p(R_), _p(xyz), _p(ok)), "triangulate")
        return [(xyz[j.pt_ofs:j.pt_ofs + j.npts].copy(), ok[j.pt_ofs:j.pt_ofs + j.npts].copy()) for j in arr]

    # ---- pose-only ---------------------------------------------------------
    def pose_only(self, jobs, cam, chi2_th=5.991, rounds=4, iters=10):
        """jobs: list of (pose[7], xyz[n,3], uv[n,2]). returns list of (pose, outlier, n_inlier)."""
        n = len(jobs)
        arr = (PoseJob * n)()
        xs, us, ofs = [], [], 0
        for i, (T, xyz, uv) in enumerate(jobs):
            xyz = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3); uv = _f32(uv, 2)
            arr[i] = PoseJob(ofs, xyz.shape[0], (C.c_double * 7)(*_d(T)), 0, 0)
            ofs += xyz.shape[0]
            xs.append(xyz); us.append(uv)
        X = np.ascontiguousarray(np.concatenate(xs)) if ofs else np.zeros((1, 3))
        U = np.ascontiguousarray(np.concatenate(us)) if ofs else np.zeros((1, 2), np.float32)
        outl = np.zeros(max(ofs, 1), np.uint8)
        self._chk(self.L.svslam_pose_only_batch(self.h, n, arr, ofs, _p(_d(cam)), _p(X), _p(U), _p(outl),
                                                C.c_double(chi2_th), rounds, iters), "pose_only")
        return [(np.array(j.pose[:]), outl[j.pt_ofs:j.pt_ofs + j.npts].copy(), j.n_inlier) for j in arr]

    # ---- local BA ------------------------------------------------------------
    def local_ba(self, jobs, cam_l, ext_l, cam_r, ext_r, huber_delta=5.991, iters=10, split=False, between=None):
        """jobs: list of (poses[k,7], pts[m,3], obs_kf, obs_lm, obs_is_right, obs_uv).
        returns list of (poses, pts, edge_chi2, iters_done).  split=True goes through
        svslam_local_ba_submit / _collect (between() runs while the batch is in flight)."""
        n = len(jobs)
        arr = (BaJob * n)()
        P, X, K_, L_, R_, U = [], [], [], [], [], []
        ko = lo = oo = 0
        for i, (poses, pts, okf, olm, ori, ouv) in enumerate(jobs):
            poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7)
            pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
            okf = np.ascontiguousarray(okf, np.int32); olm = np.ascontiguousarray(olm, np.int32)
            ori = np.ascontiguousarray(ori, np.uint8); ouv = _f32(ouv, 2)
            arr[i] = BaJob(ko, poses.shape[0], lo, pts.shape[0], oo, okf.shape[0], 0, 0)
            ko += poses.shape[0]; lo += pts.shape[0]; oo += okf.shape[0]
            P.append(poses); X.append(pts); K_.append(okf); L_.append(olm); R_.append(ori); U.append(ouv)
        P = np.ascontiguousarray(np.concatenate(P)); X = np.ascontiguousarray(np.concatenate(X))
        K_ = np.ascontiguousarray(np.concatenate(K_)); L_ = np.ascontiguousarray(np.concatenate(L_))
        R_ = np.ascontiguousarray(np.concatenate(R_)); U = np.ascontiguousarray(np.concatenate(U))
        chi2 = np.zeros(max(oo, 1))
        if split:
            self._chk(self.L.svslam_local_ba_submit(self.h, n, arr, _p(_d(cam_l)), _p(_d(ext_l)), _p(_d(cam_r)),
                                                    _p(_d(ext_r)), ko, _p(P), lo, _p(X), oo, _p(K_), _p(L_), _p(R_),
                                                    _p(U), C.c_double(huber_delta), iters), "local_ba_submit")
            if between is not None:
                between()
            self._chk(self.L.svslam_local_ba_collect(self.h, n, arr, ko, _p(P), lo, _p(X), oo, _p(chi2)), "local_ba_collect")
        else:
            self._chk(self.L.svslam_local_ba_batch(self.h, n, arr, _p(_d(cam_l)), _p(_d(ext_l)), _p(_d(cam_r)),
                                                   _p(_d(ext_r)), ko, _p(P), lo, _p(X), oo, _p(K_), _p(L_), _p(R_),
                                                   _p(U), C.c_double(huber_delta), iters, _p(chi2)), "local_ba")
        out = []
        for j in arr:
            out.append((P[j.kf_ofs:j.kf_ofs + j.nkf].copy(), X[j.lm_ofs:j.lm_ofs + j.nlm].copy(),
                        chi2[j.obs_ofs:j.obs_ofs + j.nobs].copy(), j.iters_done))
        return out

    # ---- shared-map BA (one rank's shard; the LM driver is shared_ba.py) ---------
    def sba_open(self, cam_l, ext_l, cam_r, ext_r, poses, pts, okf, olm, ori, ouv, huber_delta=5.991):
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7); pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
        okf = np.ascontiguousarray(okf, np.int32); olm = np.ascontiguousarray(olm, np.int32)
        ori = np.ascontiguousarray(ori, np.uint8); ouv = _f32(ouv, 2)
        self._sba = (poses.shape[0], pts.shape[0], okf.shape[0])
        self._chk(self.L.svslam_sba_open(self.h, _p(_d(cam_l)), _p(_d(ext_l)), _p(_d(cam_r)), _p(_d(ext_r)), poses.shape[0], _p(poses),
                                         pts.shape[0], _p(pts), okf.shape[0], _p(okf), _p(olm), _p(ori), _p(ouv),
                                         C.c_double(huber_delta)), "sba_open")
        return self.L.svslam_sba_io_doubles(poses.shape[0])

    def sba_phase(self, phase, lam, io):
        self._chk(self.L.svslam_sba_phase(self.h, phase, C.c_double(lam), _p(io)), "sba_phase")
        return io

    def sba_comm_init(self, nranks, rank, id128):
        """RCCL communicator of this context's rank for svslam_sba_solve (id128: sba_comm_unique_id() of rank 0)"""
        buf = (C.c_char * 128).from_buffer_copy(bytes(id128))
        self._chk(self.L.svslam_sba_comm_init(self.h, nranks, rank, buf), "sba_comm_init")

    def sba_comm_destroy(self):
        self._chk(self.L.svslam_sba_comm_destroy(self.h), "sba_comm_destroy")

    def sba_solve(self, iters=10):
        """the LM loop of the open shard inside the library (svslam_sba_solve): returns (iterations, lambda,
        trace[ntrials, 6], stats dict)"""
        it = C.c_int(0); lam = C.c_double(0); nt = C.c_int(0)
        cap = 10 * max(iters, 1) + 2
        tr = np.zeros((cap, 6)); st = np.zeros(4)
        self._chk(self.L.svslam_sba_solve(self.h, iters, C.byref(it), C.byref(lam), _p(tr), cap, C.byref(nt), _p(st)), "sba_solve")
        return it.value, lam.value, tr[:nt.value].copy(), dict(trials=int(st[0]), ms=float(st[1]), ms_per_trial=float(st[2]),
                                                                 allreduce_bytes_per_trial=int(st[3]))

    def sba_close(self):
        nkf, nlm, nobs = self._sba
        poses = np.zeros((nkf, 7)); pts = np.zeros((nlm, 3)); chi2 = np.zeros(max(nobs, 1))
        self._chk(self.L.svslam_sba_close(self.h, _p(poses), _p(pts), _p(chi2)), "sba_close")
        return poses, pts, chi2[:nobs]

    def dmap_read(self, stream, max_kf, max_lm):
        """test hook: one stream's device-resident window and landmark arena (svslam_dmap_read)"""
        d = dict(kf_frame=np.zeros(max_kf, np.int64), kf_id=np.zeros(max_kf, np.int32), kf_pose=np.zeros((max_kf, 7)),
                 kf_n=np.zeros(max_kf, np.int32), lm_id=np.zeros(max_lm, np.int32), lm_pos=np.zeros((max_lm, 3)),
                 lm_obs=np.zeros(max_lm, np.int32), lm_state=np.zeros(max_lm, np.uint8))
        self._chk(self.L.svslam_dmap_read(self.h, stream, _p(d["kf_frame"]), _p(d["kf_id"]), _p(d["kf_pose"]), _p(d["kf_n"]),
                                          _p(d["lm_id"]), _p(d["lm_pos"]), _p(d["lm_obs"]), _p(d["lm_state"])), "dmap_read")
        return d

    # ---- resident tracking -----------------------------------------------------
    def rtrack_upload(self, lists):
        """lists: [(stream, xy[n,2], mp[n], xyz[n,3])] -> replaces the resident feature lists"""
        n = len(lists)
        streams = (C.c_int * n)(*[l[0] for l in lists])
        cnt = [len(l[2]) for l in lists]
        ofs = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.int32) if n else np.zeros(0, np.int32)
        xy = np.ascontiguousarray(np.concatenate([_f32(l[1], 2) for l in lists]))
        mp = np.ascontiguousarray(np.concatenate([np.asarray(l[2], np.int32) for l in lists]))
        xyz = np.ascontiguousarray(np.concatenate([np.asarray(l[3], np.float64).reshape(-1, 3) for l in lists]))
        self._chk(self.L.svslam_rtrack_upload(self.h, n, streams, _p(ofs), (C.c_int * n)(*cnt), _p(xy), _p(mp), _p(xyz)),
                  "rtrack_upload")

    def rtrack(self, jobs, cam, params=None, chi2_th=5.991):
        """jobs: [(stream, prev_slot, next_slot, next_img, pose[7], T_cam_w[7], npts)].
        returns list of dict(xy, mp, pose, n_tracked, n_edges, n_outlier)."""
        n = len(jobs)
        arr = (RtrackJob * n)()
        ofs, keep = 0, []
        for i, (sid, ps, ns, img, T, Tc, npts) in enumerate(jobs):
            arr[i] = RtrackJob(sid, ps, ns, ofs, npts, (C.c_double * 7)(*_d(T)), (C.c_double * 7)(*_d(Tc)), 0, 0, 0, 0)
            ofs += npts
            keep.append(np.ascontiguousarray(img, np.uint8))
        ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in keep]); st = (C.c_int * n)(*[im.shape[1] for im in keep])
        xy = np.zeros((max(ofs, 1), 2), np.float32); mp = np.zeros(max(ofs, 1), np.int32)
        prm = params or LkParams(3, 30, 0.01, 1e-4, 1)
        self._chk(self.L.svslam_rtrack_batch(self.h, n, arr, ptrs, st, 0, ofs, _p(_d(cam)), _p(xy), _p(mp), C.byref(prm),
                                             C.c_double(chi2_th)), "rtrack")
        out = []
        for j in arr:
            s = slice(j.pt_ofs, j.pt_ofs + j.n_tracked)
            out.append(dict(xy=xy[s].copy(), mp=mp[s].copy(), pose=np.array(list(j.pose)), n_tracked=j.n_tracked,
                            n_edges=j.n_edges, n_outlier=j.n_outlier))
        return out

    # ---- fused tracking --------------------------------------------------------
    def track(self, jobs, cam, params=None, chi2_th=5.991, device=False):
        """jobs: list of (prev_slot, next_slot, next_img, pose[7], prev_xy, guess_xy, has_mp, xyz).
        returns list of dict(next_xy, status, outlier, pose, n_tracked, n_inlier)."""
        n = len(jobs)
        arr = (TrackJob * n)()
        keep, prevs, nexts, mps, xs, ofs = [], [], [], [], [], 0
        for i, (ps, ns, img, T, p, q, mp, xyz) in enumerate(jobs):
            p = _f32(p, 2); q = _f32(q, 2)
            arr[i] = TrackJob(ps, ns, ofs, p.shape[0], (C.c_double * 7)(*_d(T)), 0, 0)
            ofs += p.shape[0]
            prevs.append(p); nexts.append(q)
            mps.append(np.ascontiguousarray(mp, np.uint8)); xs.append(np.ascontiguousarray(xyz, np.float64).reshape(-1, 3))
            keep.append(img if device else np.ascontiguousarray(img, np.uint8))
        if device:
            ptrs = (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in keep]); st = (C.c_int * n)(*[self.width] * n)
        else:
            ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in keep]); st = (C.c_int * n)(*[im.shape[1] for im in keep])
        prev = np.ascontiguousarray(np.concatenate(prevs)); nxt = np.ascontiguousarray(np.concatenate(nexts)).copy()
        mp = np.ascontiguousarray(np.concatenate(mps)); X = np.ascontiguousarray(np.concatenate(xs))
        status = np.zeros(max(ofs, 1), np.uint8); outl = np.zeros(max(ofs, 1), np.uint8)
        prm = params or LkParams(3, 30, 0.01, 1e-4, 1)
        self._chk(self.L.svslam_track_batch(self.h, n, arr, ptrs, st, 1 if device else 0, ofs, _p(_d(cam)), _p(prev),
                                            _p(nxt), _p(mp), _p(X), _p(status), _p(outl), C.byref(prm),
                                            C.c_double(chi2_th)), "track_batch")
        out = []
        for j in arr:
            s = slice(j.pt_ofs, j.pt_ofs + j.npts)
            out.append(dict(next_xy=nxt[s].copy(), status=status[s].copy(), outlier=outl[s].copy(),
                            pose=np.array(j.pose[:]), n_tracked=j.n_tracked, n_inlier=j.n_inlier))
        return out


# ---- synthetic stream (CPU generator; HIP generator is in libsvslam_hip.so) ----
class SynthView(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("R", C.c_float * 9), ("C", C.c_double * 3), ("seed", C.c_uint32), ("noise_seed", C.c_uint32),
                ("scale", C.c_float)]


KITTI00_HALF_CAM = (718.856 * 0.5, 718.856 * 0.5, 607.1928 * 0.5, 185.2157 * 0.5)
KITTI00_BASELINE = 0.537166
_synth = None


def sba_comm_unique_id():
    """the 128-byte RCCL id rank 0 hands to every rank (svslam_sba_comm_unique_id)"""
    buf = (C.c_char * 128)()
    if load().svslam_sba_comm_unique_id(buf) != 0:
        raise RuntimeError("svslam_sba_comm_unique_id failed (librccl.so missing?)")
    return bytes(buf.raw)


def synth_lib():
    global _synth
    if _synth is None:
        p = os.path.join(LIB_DIR, "libsvslam_synth.so")
        if not os.path.exists(p):
            _build.build_synth()
        _synth = C.CDLL(p)
    return _synth


def synth_pair(seed, frame, w=620, h=188, cam=KITTI00_HALF_CAM, baseline=KITTI00_BASELINE):
    left = np.zeros((h, w), np.uint8); right = np.zeros((h, w), np.uint8)
    camv = (C.c_double * 4)(*cam)
    synth_lib().svs_synth_render_pair(C.c_uint32(seed), frame, w, h, camv, C.c_double(baseline), _p(left), _p(right), w)
    return left, right


def synth_gt(seed, frame):
    T = np.zeros(7)
    synth_lib().svs_synth_gt(C.c_uint32(seed), frame, _p(T))
    return T


def synth_views(seed, frame, cam=KITTI00_HALF_CAM, baseline=KITTI00_BASELINE):
    vl, vr = SynthView(), SynthView()
    camv = (C.c_double * 4)(*cam)
    synth_lib().svs_synth_make_views(C.c_uint32(seed), frame, camv, C.c_double(baseline), C.byref(vl), C.byref(vr))
    return vl, vr


def synth_render_streams_device(seeds, frame0, nframes, w, h, d_left, d_right, device=0, cam=KITTI00_HALF_CAM,
                                baseline=KITTI00_BASELINE, block=4096):
    """render frames [frame0, frame0 + nframes) of every stream in `seeds` into the device buffers
    d_left / d_right, laid out [stream][nframes][h*w]; a few launches per `block` images."""
    n = len(seeds)
    per = max(1, block // max(nframes, 1))            # streams per launch
    camv = (C.c_double * 4)(*cam)
    img = w * h
    for s0 in range(0, n, per):
        m = min(per, n - s0)
        sd = (C.c_uint32 * m)(*[int(x) & 0xFFFFFFFF for x in seeds[s0:s0 + m]])
        vl = (SynthView * (m * nframes))(); vr = (SynthView * (m * nframes))()
        synth_lib().svs_synth_make_views_batch(m, sd, frame0, nframes, camv, C.c_double(baseline), vl, vr)
        for arr, base in ((vl, d_left), (vr, d_right)):
            rc = load().svslam_synth_render_batch(device, m * nframes, arr, w, h, C.c_void_p(base + s0 * nframes * img))
            if rc != 0:
                raise RuntimeError("svslam_synth_render_batch failed (%d)" % rc)


def synth_render_device(views, w, h, d_out, device=0):
    """render len(views) images into the device buffer d_out (tight w*h each)."""
    n = len(views)
    arr = (SynthView * n)(*views)
    rc = load().svslam_synth_render_batch(device, n, arr, w, h, C.c_void_p(d_out))
    if rc != 0:
        raise RuntimeError("svslam_synth_render_batch failed (%d)" % rc)
