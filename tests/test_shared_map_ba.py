"""Shared-map bundle adjustment (BASELINE config 5): landmarks sharded over ranks, the reduced camera
system all-reduced per LM trial (stereovision-slam_amd/shared_ba.py drives g2o's LM control flow).

CPU (gloo, world 2): the driver + the reduction structure with a numpy engine that restates the phases
(test infrastructure) — two ranks on half the landmarks each reproduce one rank on all of them to
rounding, and both reproduce the oracle's local BA.
GPU (-m gpu): the product engine (phases of k_local_ba_t<1>): one rank reproduces svslam_local_ba_batch,
two ranks (gloo, same device) reproduce one rank to 1e-9."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

import common as cm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class NumpyEngine:
    """the phases of include/svslam.h:svslam_sba_phase restated in numpy (dense, analytic Jacobians)"""

    def __init__(self, poses, pts, okf, olm, ori, ouv, delta=5.991):
        from scipy.spatial.transform import Rotation
        self.R = Rotation.from_quat(poses[:, :4]).as_matrix(); self.t = poses[:, 4:].copy()
        self.X = pts.copy(); self.okf, self.olm, self.ori = okf, olm, ori.astype(float)
        self.uv = ouv.astype(np.float64); self.delta = delta; self.K = len(poses)
        self.chi2 = None

    def _lin(self):
        fx, fy, cx, cy = cm.CAM
        q = np.einsum("eij,ej->ei", self.R[self.okf], self.X[self.olm]) + self.t[self.okf]
        p = q.copy(); p[:, 0] -= cm.BASELINE * self.ori
        zi = 1.0 / p[:, 2]
        e = self.uv - np.stack([fx * p[:, 0] * zi + cx, fy * p[:, 1] * zi + cy], 1)
        s = (e * e).sum(1)
        big = s > self.delta ** 2
        sq = np.sqrt(np.maximum(s, 1e-300))
        rho = np.where(big, 2 * sq * self.delta - self.delta ** 2, s); w = np.where(big, self.delta / sq, 1.0)
        M = np.zeros((len(s), 2, 3))
        M[:, 0, 0] = -fx * zi; M[:, 0, 2] = fx * p[:, 0] * zi * zi; M[:, 1, 1] = -fy * zi; M[:, 1, 2] = fy * p[:, 1] * zi * zi
        qh = np.zeros((len(s), 3, 3))
        qh[:, 0, 1] = q[:, 2]; qh[:, 0, 2] = -q[:, 1]; qh[:, 1, 0] = -q[:, 2]; qh[:, 1, 2] = q[:, 0]; qh[:, 2, 0] = q[:, 1]; qh[:, 2, 1] = -q[:, 0]
        Jp = np.concatenate([M, M @ qh], 2)              # [I | -q^]
        Jl = M @ self.R[self.okf]
        return e, s, rho, w, Jp, Jl

    def phase(self, p, lam, io):
        n = 6 * self.K
        oS, obs, obp, ohd, osc = 0, n * n, n * n + n, n * n + 2 * n, n * n + 3 * n
        out = np.zeros_like(io)
        if p in (1, 2):
            e, s, rho, w, Jp, Jl = self._lin()
            Hpp = np.zeros((n, n)); bp = np.zeros(n); M = len(self.X)
            Hll = np.zeros((M, 3, 3)); bl = np.zeros((M, 3)); W = {}
            for i in range(len(s)):
                k, l = self.okf[i], self.olm[i]
                Hpp[6 * k:6 * k + 6, 6 * k:6 * k + 6] += w[i] * Jp[i].T @ Jp[i]; bp[6 * k:6 * k + 6] -= w[i] * Jp[i].T @ e[i]
                Hll[l] += w[i] * Jl[i].T @ Jl[i]; bl[l] -= w[i] * Jl[i].T @ e[i]
                W[(k, l)] = W.get((k, l), 0) + w[i] * Jp[i].T @ Jl[i]
            if p == 1:
                out[ohd:ohd + n] = np.diag(Hpp)
                seen = np.bincount(self.olm, minlength=M) > 0
                out[osc + 1] = np.abs(Hll[seen][:, [0, 1, 2], [0, 1, 2]]).max() if seen.any() else 0.0
                return out
            S = Hpp.copy(); bs = bp.copy()
            bylm = {}
            for (k, l), Wkl in W.items():
                bylm.setdefault(l, []).append((k, Wkl))
            self.Dinv = np.zeros((M, 3, 3))
            for l, blocks in bylm.items():
                Di = np.linalg.inv(Hll[l] + lam * np.eye(3)); self.Dinv[l] = Di
                for (a, Wa) in blocks:
                    Y = Wa @ Di
                    bs[6 * a:6 * a + 6] -= Y @ bl[l]
                    for (b, Wb) in blocks:
                        S[6 * a:6 * a + 6, 6 * b:6 * b + 6] -= Y @ Wb.T
            self.W, self.bl, self.bylm, self.e = W, bl, bylm, e
            self.backup = (self.R.copy(), self.t.copy(), self.X.copy())
            out[oS:obs] = S.ravel(); out[obs:obp] = bs; out[obp:ohd] = bp; out[osc] = rho.sum()
            self.chi2 = s
            return out
        if p == 3:
            from scipy.spatial.transform import Rotation
            S = io[oS:obs].reshape(n, n); bs = io[obs:obp]; bp = io[obp:ohd]
            try:
                np.linalg.cholesky(S); ok = 1.0
            except np.linalg.LinAlgError:
                ok = 0.0
            sl = sp = 0.0
            if ok:
                dx = np.linalg.solve(S, bs)
                for l, blocks in self.bylm.items():
                    g = self.bl[l] - sum(Wkl.T @ dx[6 * k:6 * k + 6] for (k, Wkl) in blocks)
                    d = self.Dinv[l] @ g
                    self.X[l] += d
                    sl += float(d @ (lam * d + self.bl[l]))
                sp = float(dx @ (lam * dx + bp))
                for k in range(self.K):
                    u, om = dx[6 * k:6 * k + 3], dx[6 * k + 3:6 * k + 6]
                    th = np.linalg.norm(om); Om = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]])
                    V = np.eye(3) + (0.5 * Om if th < 1e-10 else (1 - np.cos(th)) / th ** 2 * Om + (th - np.sin(th)) / th ** 3 * Om @ Om)
                    dR = Rotation.from_rotvec(om).as_matrix()
                    self.R[k] = dR @ self.R[k]; self.t[k] = dR @ self.t[k] + V @ u
            e, s, rho, w, Jp, Jl = self._lin()
            self.chi2 = s
            out[osc + 2] = ok; out[osc + 3] = sl; out[osc + 4] = sp; out[osc + 5] = rho.sum()
            return out
        if p == 4:
            self.R, self.t, self.X = self.backup[0].copy(), self.backup[1].copy(), self.backup[2].copy()
        return out

    def poses(self):
        from scipy.spatial.transform import Rotation
        return np.concatenate([Rotation.from_matrix(self.R).as_quat(), self.t], 1)


def _problem(seed=5, nkf=6, nlm=160):
    rng = np.random.default_rng(seed)
    p = cm.make_ba_problem(rng, nkf, nlm)
    m = rng.random(len(p["okf"])) < 0.5
    cnt = np.bincount(p["olm"][m], minlength=nlm)
    m &= cnt[p["olm"]] >= 2
    return p["poses0"], p["pts0"], p["okf"][m], p["olm"][m], p["ori"][m], p["ouv"][m]


def _cpu_rank_main():
    """entry of the world-2 gloo worker (re-executes this file)"""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    sdist = importlib.import_module("stereovision-slam_amd.dist")
    sba = importlib.import_module("stereovision-slam_amd.shared_ba")
    rk = sdist.init("gloo")
    poses, pts, okf, olm, ori, ouv = _problem()
    mine, k2, l2, r2, u2, _ = sba.shard_by_landmark(len(pts), okf, olm, ori, ouv, rk.rank, rk.world)
    eng = NumpyEngine(poses, pts[mine], k2, l2, r2, u2)
    it, lam = sba.shared_map_ba(eng, rk, len(poses), iters=10)
    np.savez(os.environ["SBA_OUT"] + ".%d.npz" % rk.rank, poses=eng.poses(), pts=eng.X, mine=mine, it=it, lam=lam)
    rk.close()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_shared_map_ba_over_gloo_equals_one_rank_and_the_oracle(tmp_path, orc, world):
    """BASELINE config 5 at the world sizes of one node (2, 4, 8 ranks; gloo on the CPU): landmarks dealt over the ranks, the
    reduced camera system all-reduced per LM trial — every rank must end with the one-rank answer (= the oracle's)"""
    sdist = importlib.import_module("stereovision-slam_amd.dist")
    sba = importlib.import_module("stereovision-slam_amd.shared_ba")
    poses, pts, okf, olm, ori, ouv = _problem()
    one = NumpyEngine(poses, pts, okf, olm, ori, ouv)
    it1, lam1 = sba.shared_map_ba(one, sdist.Rank(0, 0, 1), len(poses), iters=10)
    # the driver is g2o's LM: the oracle's local BA (analytic Jacobians) from the same start gives the same
    # trajectory (gauge-invariant comparison: no vertex is fixed)
    po, xo, co, ito = orc.local_ba(cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R, poses, pts, okf, olm, ori, ouv, jac_mode=0)
    assert it1 == ito == 10
    rel = lambda P: np.array([orc.se3_mul(P[k], orc.se3_inv(P[0])) for k in range(len(P))])
    assert np.allclose(rel(one.poses())[:, 4:], rel(po)[:, 4:], atol=1e-6)
    assert abs(np.where(one.chi2 <= 5.991 ** 2, one.chi2, 2 * 5.991 * np.sqrt(one.chi2) - 5.991 ** 2).sum() -
               np.where(co <= 5.991 ** 2, co, 2 * 5.991 * np.sqrt(co) - 5.991 ** 2).sum()) < 1e-6 * co.sum()
    # `world` ranks, landmarks dealt round-robin, reduced system all-reduced over gloo
    out = str(tmp_path / "sba")
    port = str(29533 + world)
    env = dict(os.environ, SBA_OUT=out, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                        "--master-port", port, os.path.abspath(__file__), "--rank-main"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    res = [np.load(out + ".%d.npz" % k) for k in range(world)]
    a = res[0]
    X = np.zeros_like(pts); seen = np.zeros(len(pts), int)
    for b in res:
        assert int(b["it"]) == it1 and float(b["lam"]) == float(a["lam"])
        assert np.array_equal(a["poses"], b["poses"])                  # every rank solves the same reduced system
        X[b["mine"]] = b["pts"]; seen[b["mine"]] += 1
    assert (seen == 1).all()                                           # the shards partition the landmarks
    assert np.allclose(a["poses"], one.poses(), atol=1e-9) and abs(float(a["lam"]) - lam1) <= 1e-9 * lam1
    assert np.allclose(X, one.X, atol=1e-9)


def _gpu_rank_main():
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    svs = importlib.import_module("stereovision-slam_amd")
    sdist = importlib.import_module("stereovision-slam_amd.dist")
    sba = importlib.import_module("stereovision-slam_amd.shared_ba")
    rk = sdist.init("gloo")                       # two ranks on the one GPU of the test box
    poses, pts, okf, olm, ori, ouv = _problem(7, 10, 900)
    mine, k2, l2, r2, u2, _ = sba.shard_by_landmark(len(pts), okf, olm, ori, ouv, rk.rank, rk.world)
    ctx = svs.Context(cm.W, cm.H, max_slots=1, max_jobs=1, max_kf=11, max_lm=2048, max_obs=20000)
    eng = sba.HipEngine(ctx, cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R, poses, pts[mine], k2, l2, r2, u2)
    it, lam = sba.shared_map_ba(eng, rk, len(poses), iters=10)
    P, X, chi2 = eng.close()
    np.savez(os.environ["SBA_OUT"] + ".%d.npz" % rk.rank, poses=P, pts=X, mine=mine, it=it, lam=lam)
    ctx.close(); rk.close()


@pytest.mark.gpu
def test_shared_map_ba_on_the_gpu(svs, tmp_path):
    sdist = importlib.import_module("stereovision-slam_amd.dist")
    sba = importlib.import_module("stereovision-slam_amd.shared_ba")
    poses, pts, okf, olm, ori, ouv = _problem(7, 10, 900)
    ctx = svs.Context(cm.W, cm.H, max_slots=1, max_jobs=1, max_kf=11, max_lm=2048, max_obs=20000)
    # one rank through the phases == the single-launch local BA (same kernel code, host-driven LM)
    (pr, xr, cr, itr), = ctx.local_ba([(poses, pts, okf, olm, ori, ouv)], cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)
    eng = sba.HipEngine(ctx, cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R, poses, pts, okf, olm, ori, ouv)
    # an open shard owns the staging arena and job 0 of the BA scratch: every other batched call is refused
    # (used to overwrite the shard silently)
    for call in (lambda: ctx.local_ba([(poses, pts, okf, olm, ori, ouv)], cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R),
                 lambda: ctx.pose_only([(cm.EXT_L, pts[:50], np.zeros((50, 2), np.float32))], cm.CAM),
                 lambda: ctx.pyramid([0], [np.zeros((cm.H, cm.W), np.uint8)]),
                 lambda: ctx.triangulate([(np.ones((4, 2)), np.ones((4, 2)), None, 0.0)], cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)):
        with pytest.raises(RuntimeError, match="svslam_sba_close"):
            call()
    it1, lam1 = sba.shared_map_ba(eng, sdist.Rank(0, 0, 1), len(poses), iters=10)
    P1, X1, C1 = eng.close()
    assert it1 == itr == 10
    assert np.allclose(P1, pr, atol=1e-9) and np.allclose(X1, xr, atol=1e-8) and np.allclose(C1, cr, rtol=1e-7, atol=1e-9)
    ctx.close()
    out = str(tmp_path / "sba")
    env = dict(os.environ, SBA_OUT=out, MASTER_ADDR="127.0.0.1", MASTER_PORT="29534")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29534", os.path.abspath(__file__), "--gpu-rank-main"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    a, b = np.load(out + ".0.npz"), np.load(out + ".1.npz")
    assert int(a["it"]) == int(b["it"]) == it1
    assert np.array_equal(a["poses"], b["poses"])
    assert np.allclose(a["poses"], P1, atol=1e-9), np.abs(a["poses"] - P1).max()
    X = np.zeros_like(pts); X[a["mine"]] = a["pts"]; X[b["mine"]] = b["pts"]
    assert np.allclose(X, X1, atol=1e-8)


def _native_single_main():
    """svslam_sba_solve on one rank with a one-rank RCCL communicator; runs in its own process so that the test
    process never loads RCCL (the GPU box's pytest also imports torch, which brings its own RCCL + HIP runtime)"""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import lm_cases as lc
    import oracle_lib as orc
    svs = importlib.import_module("stereovision-slam_amd")
    sdist = importlib.import_module("stereovision-slam_amd.dist")
    sba = importlib.import_module("stereovision-slam_amd.shared_ba")
    rank = sdist.Rank(0, 0, 1)
    for name, job, iters in (("default", _problem(7, 10, 900), 10), ("rejecting 1024", lc.ba_synth_case(1024), 10),
                             ("rejecting 1013", lc.ba_synth_case(1013), 10)):
        poses, pts, okf, olm, ori, ouv = job
        ctx = svs.Context(cm.W, cm.H, max_slots=1, max_jobs=1, max_kf=11, max_lm=2048, max_obs=20000)
        (pr, xr, cr, itr), = ctx.local_ba([job], cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R, iters=iters)
        eng = sba.HipEngine(ctx, cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R, poses, pts, okf, olm, ori, ouv)
        tr_py = []
        it_py, lam_py = sba.shared_map_ba(eng, rank, len(poses), iters=iters, trace=tr_py)
        P_py, X_py, C_py = eng.close()
        eng = sba.HipEngine(ctx, cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R, poses, pts, okf, olm, ori, ouv)
        it_n, lam_n, tr_n, stats = sba.shared_map_ba_native(ctx, rank, iters=iters)          # creates the RCCL communicator
        P_n, X_n, C_n = eng.close()
        assert it_n == it_py == itr, name
        assert np.array_equal(tr_n, np.array(tr_py)), (name, np.abs(tr_n - np.array(tr_py)).max())
        assert lam_n == lam_py
        assert np.array_equal(P_n, P_py) and np.array_equal(X_n, X_py) and np.array_equal(C_n, C_py)
        assert np.allclose(P_n, pr, atol=1e-9) and np.allclose(X_n, xr, atol=1e-8), name
        ref = orc.local_ba_trace(cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R, *job, iters=iters, jac_mode=0)[4]
        lc.assert_traces_agree(tr_n, ref, need_rejected=0 if name == "default" else 1, what="native sba " + name)
        K = len(poses)       # (6K)^2 + 3 (6K) + 1 doubles of the reduced system + chi2, and two scalars
        assert stats["trials"] == len(tr_n) and stats["allreduce_bytes_per_trial"] == 8 * (36 * K * K + 18 * K + 3)
        print("native shared-map BA (%s): %d trials, %.3f ms per trial (1 rank, RCCL world 1, %d B all-reduced per trial)"
              % (name, stats["trials"], stats["ms_per_trial"], stats["allreduce_bytes_per_trial"]))
        ctx.sba_comm_destroy()
        ctx.close()
    print("native-single ok")


@pytest.mark.gpu
def test_native_shared_map_ba_with_rccl_on_the_device_buffer():
    """svslam_sba_solve: the LM loop inside the library, ncclAllReduce on the device buffer (a one-rank RCCL
    communicator on the 1-GPU test box: RCCL is initialised and every collective of the path runs).  Must equal
    the Python-driven phases trial for trial (same kernels, same control flow), the single-launch local BA at
    1e-9, and follow the oracle through rejected trials."""
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--native-single-main"], capture_output=True, text=True, timeout=600)
    print(r.stdout[-1500:])
    assert r.returncode == 0 and "native-single ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def _native_rank_main():
    """one rank per GPU, nccl backend: the product path of BASELINE config 5"""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    svs = importlib.import_module("stereovision-slam_amd")
    sdist = importlib.import_module("stereovision-slam_amd.dist")
    sba = importlib.import_module("stereovision-slam_amd.shared_ba")
    rk = sdist.init("nccl")
    poses, pts, okf, olm, ori, ouv = _problem(7, 10, 900)
    mine, k2, l2, r2, u2, _ = sba.shard_by_landmark(len(pts), okf, olm, ori, ouv, rk.rank, rk.world)
    ctx = svs.Context(cm.W, cm.H, max_slots=1, max_jobs=1, max_kf=11, max_lm=2048, max_obs=20000, device=rk.local_rank)
    eng = sba.HipEngine(ctx, cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R, poses, pts[mine], k2, l2, r2, u2)
    it, lam, tr, stats = sba.shared_map_ba_native(ctx, rk, iters=10)
    P, X, chi2 = eng.close()
    np.savez(os.environ["SBA_OUT"] + ".%d.npz" % rk.rank, poses=P, pts=X, mine=mine, it=it, lam=lam, ms_per_trial=stats["ms_per_trial"])
    ctx.sba_comm_destroy(); ctx.close(); rk.close()


@pytest.mark.gpu
def test_native_shared_map_ba_two_gpus_over_rccl(svs, tmp_path):
    """two ranks on two GPUs, nccl backend (= RCCL over xGMI), landmarks sharded, ncclAllReduce of the reduced camera
    system on the device buffers — runs wherever two GPUs are visible, skipped on the 1-GPU test box"""
    if svs.load().svslam_device_count() < 2:
        pytest.skip("needs two GPUs")
    poses, pts, okf, olm, ori, ouv = _problem(7, 10, 900)
    ctx = svs.Context(cm.W, cm.H, max_slots=1, max_jobs=1, max_kf=11, max_lm=2048, max_obs=20000)
    (pr, xr, cr, itr), = ctx.local_ba([(poses, pts, okf, olm, ori, ouv)], cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)
    ctx.close()
    out = str(tmp_path / "sba")
    env = dict(os.environ, SBA_OUT=out, MASTER_ADDR="127.0.0.1", MASTER_PORT="29535", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29535", os.path.abspath(__file__), "--native-rank-main"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    a, b = np.load(out + ".0.npz"), np.load(out + ".1.npz")
    assert int(a["it"]) == int(b["it"]) == itr and np.array_equal(a["poses"], b["poses"])
    assert np.allclose(a["poses"], pr, atol=1e-9)
    X = np.zeros_like(pts); X[a["mine"]] = a["pts"]; X[b["mine"]] = b["pts"]
    assert np.allclose(X, xr, atol=1e-8)


if __name__ == "__main__":
    if "--rank-main" in sys.argv:
        _cpu_rank_main()
    elif "--gpu-rank-main" in sys.argv:
        _gpu_rank_main()
    elif "--native-rank-main" in sys.argv:
        _native_rank_main()
    elif "--native-single-main" in sys.argv:
        _native_single_main()
