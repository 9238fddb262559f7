"""ctypes binding of the CPU ORACLE (oracle/_build/libsvs_oracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Never imported by the product package.
"""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(ROOT, "oracle")
ORC_SO = os.path.join(ORC_DIR, "_build", "libsvs_oracle.so")


def build(force=False):
    """make decides what is stale (the twin of the host pipeline also depends on the product's host headers)"""
    if force:
        subprocess.check_call(["make", "-C", ORC_DIR, "clean"], stdout=subprocess.DEVNULL)
    if shutil.which("make") and shutil.which("gcc"):
        subprocess.check_call(["make", "-C", ORC_DIR], stdout=subprocess.DEVNULL)
    elif not os.path.exists(ORC_SO):
        raise RuntimeError("oracle library missing and no toolchain to build it")
    return ORC_SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_gftt.restype = C.c_int
        _lib.orc_pose_only.restype = C.c_int
        _lib.orc_local_ba.restype = C.c_int
        _lib.orc_triangulate_dlt.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class LKParams(C.Structure):
    _fields_ = [("max_level", C.c_int), ("max_iter", C.c_int), ("epsilon", C.c_double),
                ("min_eig_thr", C.c_double), ("use_initial_flow", C.c_int)]


def lk_params(max_level=3, max_iter=30, epsilon=0.01, min_eig_thr=1e-4, use_initial_flow=1):
    return LKParams(max_level, max_iter, epsilon, min_eig_thr, use_initial_flow)


def pyrdown(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().orc_pyrdown(_p(img), w, h, w, _p(out), out.shape[1])
    return out


def pyramid(img, max_level=3, win=11):
    """levels as the LK pyramid builder produces them (stops like OpenCV)."""
    levels = [np.ascontiguousarray(img, np.uint8)]
    for _ in range(max_level):
        h, w = levels[-1].shape
        nw, nh = (w + 1) // 2, (h + 1) // 2
        if nw <= win or nh <= win:   # buildOpticalFlowPyramid stops before this level
            break
        levels.append(pyrdown(levels[-1]))
    return levels


def scharr(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((h, w, 2), np.int16)
    lib().orc_scharr(_p(img), w, h, w, _p(out))
    return out


def decimate(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    dw, dh = int(np.rint(w * 0.5)), int(np.rint(h * 0.5))
    out = np.zeros((dh, dw), np.uint8)
    lib().orc_decimate(_p(img), w, h, w, _p(out), dw, dh, dw)
    return out


def lk(prev, nxt, prev_xy, next_xy, params=None):
    prev = np.ascontiguousarray(prev, np.uint8)
    nxt = np.ascontiguousarray(nxt, np.uint8)
    h, w = prev.shape
    p = np.ascontiguousarray(prev_xy, np.float32).reshape(-1, 2)
    q = np.ascontiguousarray(next_xy, np.float32).reshape(-1, 2).copy()
    n = p.shape[0]
    status = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float32)
    prm = params or lk_params()
    lib().orc_lk(_p(prev), w, _p(nxt), w, w, h, n, _p(p), _p(q), _p(status), _p(err), C.byref(prm))
    return q, status, err


def min_eig_map(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((h, w), np.float32)
    lib().orc_min_eig_map(_p(img), w, w, h, _p(out))
    return out


def gftt_mask(w, h, rect_xy):
    r = np.ascontiguousarray(rect_xy, np.float32).reshape(-1, 2)
    m = np.zeros((h, w), np.uint8)
    lib().orc_gftt_mask(_p(m), w, h, _p(r), r.shape[0])
    return m


def gftt(img, rect_xy=None, max_corners=150, quality=0.01, min_dist=20.0):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    r = np.zeros((0, 2), np.float32) if rect_xy is None else np.ascontiguousarray(rect_xy, np.float32).reshape(-1, 2)
    cap = max_corners if max_corners > 0 else w * h
    out = np.zeros((cap, 2), np.float32)
    n = lib().orc_gftt(_p(img), w, w, h, _p(r), r.shape[0], max_corners, C.c_double(quality),
                       C.c_double(min_dist), _p(out))
    return out[:n].copy()


def _d(a, n=None):
    a = np.ascontiguousarray(a, np.float64)
    return a


def se3_exp(xi):
    T = np.zeros(7)
    lib().orc_se3_exp(_p(_d(xi)), _p(T))
    return T


def se3_log(T):
    xi = np.zeros(6)
    lib().orc_se3_log(_p(_d(T)), _p(xi))
    return xi


def se3_mul(A, B):
    Cc = np.zeros(7)
    lib().orc_se3_mul(_p(_d(A)), _p(_d(B)), _p(Cc))
    return Cc


def se3_inv(T):
    Ti = np.zeros(7)
    lib().orc_se3_inv(_p(_d(T)), _p(Ti))
    return Ti


def se3_act(T, p):
    o = np.zeros(3)
    lib().orc_se3_act(_p(_d(T)), _p(_d(p)), _p(o))
    return o


IDENT = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)


def triangulate(cam_l, ext_l, cam_r, ext_r, uv_l, uv_r, T_wc=None, zmax=0.0):
    uv_l = np.ascontiguousarray(uv_l, np.float32).reshape(-1, 2)
    uv_r = np.ascontiguousarray(uv_r, np.float32).reshape(-1, 2)
    n = uv_l.shape[0]
    T = IDENT if T_wc is None else _d(T_wc)
    xyz = np.zeros((n, 3))
    ok = np.zeros(n, np.uint8)
    lib().orc_triangulate(n, _p(_d(cam_l)), _p(_d(ext_l)), _p(_d(cam_r)), _p(_d(ext_r)), _p(uv_l), _p(uv_r),
                          _p(_d(T)), C.c_double(zmax), _p(xyz), _p(ok))
    return xyz, ok


def pose_only(cam, pose, xyz, uv, chi2_th=5.991, rounds=4, iters=10):
    xyz = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
    uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
    n = xyz.shape[0]
    T = _d(pose).copy()
    outl = np.zeros(max(n, 1), np.uint8)
    ninl = lib().orc_pose_only(n, _p(_d(cam)), _p(T), _p(xyz), _p(uv), _p(outl), C.c_double(chi2_th), rounds, iters)
    return T, outl[:n], ninl


def local_ba(cam_l, ext_l, cam_r, ext_r, poses, pts, obs_kf, obs_lm, obs_is_right, obs_uv,
             huber_delta=5.991, iters=10, jac_mode=1):
    poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7).copy()
    pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3).copy()
    okf = np.ascontiguousarray(obs_kf, np.int32)
    olm = np.ascontiguousarray(obs_lm, np.int32)
    ori = np.ascontiguousarray(obs_is_right, np.uint8)
    ouv = np.ascontiguousarray(obs_uv, np.float32).reshape(-1, 2)
    ne = okf.shape[0]
    chi2 = np.zeros(max(ne, 1))
    it = lib().orc_local_ba(_p(_d(cam_l)), _p(_d(ext_l)), _p(_d(cam_r)), _p(_d(ext_r)), poses.shape[0], _p(poses),
                            pts.shape[0], _p(pts), ne, _p(okf), _p(olm), _p(ori), _p(ouv),
                            C.c_double(huber_delta), iters, jac_mode, _p(chi2))
    return poses, pts, chi2[:ne], it


TRACE_REC = 6     # ORC_TRACE_REC: iteration, lambda, chi2 before, chi2 of the trial, rho, accepted


def local_ba_trace(cam_l, ext_l, cam_r, ext_r, poses, pts, obs_kf, obs_lm, obs_is_right, obs_uv,
                   huber_delta=5.991, iters=10, jac_mode=0):
    """orc_local_ba + the per-trial LM trajectory [ntrials, TRACE_REC] (rejected trials included)"""
    poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7).copy()
    pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3).copy()
    okf = np.ascontiguousarray(obs_kf, np.int32)
    olm = np.ascontiguousarray(obs_lm, np.int32)
    ori = np.ascontiguousarray(obs_is_right, np.uint8)
    ouv = np.ascontiguousarray(obs_uv, np.float32).reshape(-1, 2)
    ne = okf.shape[0]
    chi2 = np.zeros(max(ne, 1))
    cap = 10 * max(iters, 1) + 1
    trace = np.zeros((cap, TRACE_REC)); nt = C.c_int(0)
    lib().orc_local_ba_trace.restype = C.c_int
    it = lib().orc_local_ba_trace(_p(_d(cam_l)), _p(_d(ext_l)), _p(_d(cam_r)), _p(_d(ext_r)), poses.shape[0], _p(poses),
                                  pts.shape[0], _p(pts), ne, _p(okf), _p(olm), _p(ori), _p(ouv),
                                  C.c_double(huber_delta), iters, jac_mode, _p(chi2), _p(trace), cap, C.byref(nt))
    return poses, pts, chi2[:ne], it, trace[:nt.value].copy()


def pose_only_trace(cam, pose, xyz, uv, chi2_th=5.991, rounds=4, iters=10):
    xyz = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
    uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
    n = xyz.shape[0]
    T = _d(pose).copy()
    outl = np.zeros(max(n, 1), np.uint8)
    cap = 10 * rounds * max(iters, 1) + 1
    trace = np.zeros((cap, TRACE_REC)); nt = C.c_int(0)
    lib().orc_pose_only_trace.restype = C.c_int
    ninl = lib().orc_pose_only_trace(n, _p(_d(cam)), _p(T), _p(xyz), _p(uv), _p(outl), C.c_double(chi2_th), rounds,
                                     iters, _p(trace), cap, C.byref(nt))
    return T, outl[:n], ninl, trace[:nt.value].copy()
