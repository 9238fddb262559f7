"""CPU tests pinning the oracle's geometry (SE3, triangulation, pose-only LM, local BA)
against numpy / scipy and against synthetic scenes with known ground truth."""
import numpy as np
import pytest
from scipy.linalg import expm, logm
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation

import common as cm


def _T4(T):
    M = np.eye(4); M[:3, :3] = cm.quat_R(T[:4]); M[:3, 3] = T[4:]
    return M


def _hat6(xi):
    u, w = xi[:3], xi[3:]
    M = np.zeros((4, 4))
    M[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    M[:3, 3] = u
    return M


def test_se3_exp_log_mul_inv_act(orc):
    rng = np.random.default_rng(0)
    for scale in (1e-12, 1e-6, 0.1, 1.5):
        xi = rng.normal(0, 1, 6) * scale
        T = orc.se3_exp(xi)
        assert np.allclose(_T4(T), expm(_hat6(xi)), atol=1e-12)
        assert abs(np.linalg.norm(T[:4]) - 1) < 1e-14
        assert np.allclose(orc.se3_log(T), xi, atol=1e-10 * max(1.0, 1 / max(scale, 1e-3)))
    A, B = cm.random_pose(rng, 2, 0.7), cm.random_pose(rng, 2, 0.7)
    assert np.allclose(_T4(orc.se3_mul(A, B)), _T4(A) @ _T4(B), atol=1e-13)
    assert np.allclose(_T4(orc.se3_inv(A)), np.linalg.inv(_T4(A)), atol=1e-13)
    p = rng.normal(0, 3, 3)
    assert np.allclose(orc.se3_act(A, p), (_T4(A) @ np.append(p, 1))[:3], atol=1e-13)


def test_triangulation_vs_numpy_svd_and_closed_form(orc):
    rng = np.random.default_rng(1)
    n = 200
    P = np.stack([rng.uniform(-8, 8, n), rng.uniform(-3, 1.5, n), rng.uniform(3, 120, n)], 1)
    uvl, _ = cm.project(cm.CAM, cm.EXT_L, cm.EXT_L, P)
    uvr, _ = cm.project(cm.CAM, cm.EXT_L, cm.EXT_R, P)
    uvl = uvl.astype(np.float32); uvr = uvr.astype(np.float32)
    xyz, ok = orc.triangulate(cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R, uvl, uvr)
    # numpy SVD of the same 4x4 system (algorithm.h:62-86)
    for i in range(n):
        A = []
        for uv, ext in ((uvl[i], cm.EXT_L), (uvr[i], cm.EXT_R)):
            x = (float(uv[0]) - cm.CAM[2]) / cm.CAM[0]; y = (float(uv[1]) - cm.CAM[3]) / cm.CAM[1]
            m = np.hstack([cm.quat_R(ext[:4]), ext[4:, None]])
            A += [x * m[2] - m[0], y * m[2] - m[1]]
        U, S, Vt = np.linalg.svd(np.array(A))
        ref = Vt[3, :3] / Vt[3, 3]
        assert np.allclose(xyz[i], ref, rtol=1e-8, atol=1e-9)
        assert bool(ok[i]) == bool(S[3] / S[2] < 1e-2 and ref[2] > 0)
    # rectified stereo: Z = fx * b / disparity (float32 pixel rounding limits the agreement)
    Z = cm.CAM[0] * cm.BASELINE / (uvl[:, 0].astype(np.float64) - uvr[:, 0])
    assert np.allclose(xyz[:, 2], Z, rtol=2e-3)
    assert ok.mean() > 0.95
    # gates: behind the camera / beyond zmax / world transform
    xyz2, ok2 = orc.triangulate(cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R, uvl, uvr, zmax=30.0)
    assert np.array_equal(ok2.astype(bool), ok.astype(bool) & (xyz[:, 2] <= 30.0))
    _, ok3 = orc.triangulate(cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R, uvr, uvl)   # swapped -> negative depth
    assert ok3.sum() == 0
    T = cm.random_pose(rng)
    xyz4, _ = orc.triangulate(cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R, uvl, uvr, T_wc=T)
    assert np.allclose(xyz4, xyz @ cm.quat_R(T[:4]).T + T[4:], atol=1e-10)


def _reproj(x, P, uv):
    T = np.concatenate([Rotation.from_rotvec(x[3:]).as_quat(), x[:3]])
    pr, _ = cm.project(cm.CAM, T, cm.EXT_L, P)
    return (uv - pr).ravel()


def test_pose_only_optimum_matches_scipy(orc):
    rng = np.random.default_rng(2)
    n = 150
    P = np.stack([rng.uniform(-8, 8, n), rng.uniform(-3, 1.5, n), rng.uniform(5, 50, n)], 1)
    T_true = cm.random_pose(rng, 0.5, 0.03)
    uv, _ = cm.project(cm.CAM, T_true, cm.EXT_L, P)
    uv = (uv + rng.normal(0, 0.3, uv.shape)).astype(np.float32)
    T, outl, ninl = orc.pose_only(cm.CAM, cm.EXT_L, P, uv)
    assert ninl == n and outl.sum() == 0          # 0.3 px noise: chi2 << 5.991 everywhere
    sol = least_squares(_reproj, np.zeros(6), args=(P, uv.astype(np.float64)), xtol=1e-14, ftol=1e-14, gtol=1e-14)
    T_ref = np.concatenate([Rotation.from_rotvec(sol.x[3:]).as_quat(), sol.x[:3]])
    # last round has no robust kernel and all edges are inliers -> plain least squares optimum
    assert np.allclose(T[4:], T_ref[4:], atol=1e-5) and np.allclose(T[:4], T_ref[:4], atol=1e-6)
    assert np.linalg.norm(T[4:] - T_true[4:]) < 0.03


def test_pose_only_outliers_and_protocol(orc):
    rng = np.random.default_rng(3)
    n = 230
    P = np.stack([rng.uniform(-8, 8, n), rng.uniform(-3, 1.5, n), rng.uniform(5, 50, n)], 1)
    T_true = cm.random_pose(rng, 0.6, 0.03)
    uv, _ = cm.project(cm.CAM, T_true, cm.EXT_L, P)
    uv += rng.normal(0, 0.4, uv.shape)
    bad = rng.choice(n, 35, replace=False)
    uv[bad] += rng.normal(0, 40, (35, 2)) + 15
    T, outl, ninl = orc.pose_only(cm.CAM, cm.EXT_L, P, uv.astype(np.float32))
    assert ninl == n - outl.sum()
    assert outl[bad].mean() > 0.95 and outl.sum() <= 45
    assert np.linalg.norm(T[4:] - T_true[4:]) < 0.05
    # n = 0 and all-outlier problems leave the prior untouched (optimize() returns -1)
    T0 = cm.random_pose(rng)
    T2, o2, n2 = orc.pose_only(cm.CAM, T0, np.zeros((0, 3)), np.zeros((0, 2), np.float32))
    assert np.array_equal(T2, T0) and n2 == 0
    # rounds=1: a single optimize(10) from the prior
    T3, o3, n3 = orc.pose_only(cm.CAM, cm.EXT_L, P, uv.astype(np.float32), rounds=1)
    assert np.linalg.norm(T3[4:] - T_true[4:]) < 0.5


def test_local_ba_noise_free_recovers_structure(orc):
    rng = np.random.default_rng(4)
    p = cm.make_ba_problem(rng, 6, 250, noise=0.0, outlier_frac=0.0, pose_noise=0.03, pt_noise=0.08)
    poses, pts, chi2, it = orc.local_ba(cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R, p["poses0"], p["pts0"], p["okf"],
                                        p["olm"], p["ori"], p["ouv"], jac_mode=0)
    assert it >= 3
    assert chi2.max() < 1e-3          # 10 LM iterations from a 3 cm / 8 cm perturbation: residual << 0.05 px
    # gauge-free: compare relative poses with the truth
    rel = lambda P: np.array([orc.se3_mul(P[k], orc.se3_inv(P[0])) for k in range(len(P))])
    assert np.allclose(rel(poses), rel(p["poses"]), atol=2e-4)


def test_local_ba_numeric_vs_analytic_and_robustness(orc):
    rng = np.random.default_rng(5)
    p = cm.make_ba_problem(rng, 7, 300)
    args = (cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R, p["poses0"], p["pts0"], p["okf"], p["olm"], p["ori"], p["ouv"])
    pa, xa, ca, ia = orc.local_ba(*args, jac_mode=0)
    pn, xn, cn, inn = orc.local_ba(*args, jac_mode=1)
    rel = lambda P: np.array([orc.se3_mul(P[k], orc.se3_inv(P[0])) for k in range(len(P))])
    assert np.allclose(rel(pa), rel(pn), atol=1e-4)
    assert abs(ca.sum() - cn.sum()) < 1e-4 * ca.sum()
    # cost went down and the gross outliers stand out
    _, _, c0, _ = orc.local_ba(*args, iters=0, jac_mode=0)
    assert c0.shape == ca.shape
    assert np.median(ca) < 1.0 and (ca > 5.991).mean() < 0.2
    # landmarks / keyframes without edges are left untouched
    okf = p["okf"].copy(); olm = p["olm"].copy()
    keep = (olm != 7) & (okf != 3)
    p2, x2, _, _ = orc.local_ba(cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R, p["poses0"], p["pts0"], okf[keep], olm[keep],
                                p["ori"][keep], p["ouv"][keep], jac_mode=0)
    assert np.array_equal(x2[7], p["pts0"][7]) and np.array_equal(p2[3], p["poses0"][3])
