"""Seeded LM test cases that PROVABLY exercise rejected Levenberg-Marquardt trials, and the trajectory
comparison used by the CPU and GPU tests (a defect in the path after a rejected trial
passed every per-call test because none of them rejected a trial).

A trial is *significant* when the decision it takes is not a matter of rounding: the chi2 it compares
differ by more than 1e-6 relative and |rho| > 1e-3.  Past the first insignificant trial two correct
implementations may legitimately take different branches (the sign of a difference at the rounding floor),
so trajectories are compared on the significant prefix; the cases are chosen (and asserted) to hold
rejected trials inside that prefix."""
import os

import numpy as np

import common as cm

HERE = os.path.dirname(os.path.abspath(__file__))

# (seed, iterations): common.make_ba_problem far from the optimum; multi-view landmarks
BA_SYNTH_REJECT = [(1000, 10), (1013, 10), (1021, 10), (1024, 10), (1036, 10), (1017, 16), (1033, 16)]
# (captured pipeline problem, pose noise m, rotation noise rad, outlier share, landmark noise m, iterations):
# 75-79 % single-view landmarks (the shape Backend::Optimize really produces), pushed off the optimum
BA_PIPE_REJECT = [(0, 0.5, 0.09, 0.3, 0.5, 10), (3, 0.5, 0.09, 0.3, 0.5, 10), (5, 0.3, 0.05, 0.3, 1.0, 16),
                  (9, 0.5, 0.09, 0.3, 0.5, 16), (9, 0.3, 0.05, 0.3, 1.0, 10), (11, 1.0, 0.09, 0.1, 2.0, 16)]
# seeds of hard pose-only problems (close points, prior 1-2 m / 0.2-0.4 rad off, 10-50 % gross outliers)
PO_REJECT = [0, 14, 15, 16, 24, 32, 40, 45, 55]


def ba_synth_case(seed):
    rng = np.random.default_rng(seed)
    nkf = int(rng.integers(4, 11)); nlm = int(rng.integers(150, 900))
    pn = float(rng.choice([0.3, 0.5, 1.0])); ptn = float(rng.choice([0.5, 1.0, 3.0])); of = float(rng.choice([0.1, 0.3, 0.4]))
    p = cm.make_ba_problem(rng, nkf, nlm, noise=0.5, outlier_frac=of, pose_noise=pn, pt_noise=ptn)
    return p["poses0"], p["pts0"], p["okf"], p["olm"], p["ori"], p["ouv"]


_golden = None


def pipeline_problems():
    """tests/golden/ba_pipeline.npz (make_ba_golden.py): list of dicts"""
    global _golden
    if _golden is None:
        G = np.load(os.path.join(HERE, "golden", "ba_pipeline.npz"))
        out = []
        for i in range(int(G["n"][0])):
            p = "p%02d_" % i
            d = {k: G[p + k] for k in ("cam", "ext_r", "poses0", "pts0", "okf", "olm", "ori", "ouv", "poses", "pts", "chi2",
                                       "iters", "trace")}
            d["okf"] = d["okf"].astype(np.int32); d["olm"] = d["olm"].astype(np.int32)
            d["tag"] = str(G["tags"][i])
            out.append(d)
        _golden = out
    return _golden


def ba_pipe_case(idx, pn, rot, of, ptn):
    """captured problem `idx`, perturbed: returns (cam, ext_r, job tuple)"""
    from scipy.spatial.transform import Rotation
    P = pipeline_problems()[idx]
    rng = np.random.default_rng(500 + idx)
    poses = P["poses0"].copy()
    for k in range(len(poses)):
        dq = Rotation.from_rotvec(rng.normal(0, rot, 3))
        poses[k, :4] = (dq * Rotation.from_quat(poses[k, :4])).as_quat()
        poses[k, 4:] = dq.apply(poses[k, 4:]) + rng.normal(0, pn, 3)
    pts = P["pts0"] + rng.normal(0, ptn, P["pts0"].shape)
    uv = P["ouv"].copy()
    m = rng.random(len(uv)) < of
    uv[m] += rng.normal(0, 25, (int(m.sum()), 2)).astype(np.float32)
    return P["cam"], P["ext_r"], (poses, pts, P["okf"], P["olm"], P["ori"], uv)


def po_case(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(40, 400)); bad = float(rng.choice([0.1, 0.3, 0.5])); tn = float(rng.choice([1.0, 2.0]))
    rn = float(rng.choice([0.2, 0.4])); zlo = float(rng.choice([2.5, 4.0]))
    P = np.stack([rng.uniform(-4, 4, n), rng.uniform(-2, 1.5, n), rng.uniform(zlo, zlo + 20, n)], 1)
    T_true = cm.random_pose(rng, tn, rn)
    uv, _ = cm.project(cm.CAM, T_true, cm.EXT_L, P)
    uv += rng.normal(0, 0.5, uv.shape)
    b = rng.random(n) < bad
    uv[b] += rng.normal(0, 30, (int(b.sum()), 2))
    return cm.EXT_L.copy(), P, uv.astype(np.float32)


def po_tracking_case(seed, bad=0.03):
    """a pose-only job shaped like steady tracking: the prior a frame's motion away, half-pixel noise, a few gross outliers"""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(90, 200))
    P = np.stack([rng.uniform(-8, 8, n), rng.uniform(-2, 1.5, n), rng.uniform(4.0, 40.0, n)], 1)
    T_true = cm.random_pose(rng, 0.9, 0.02)
    uv, _ = cm.project(cm.CAM, T_true, cm.EXT_L, P)
    uv += rng.normal(0, 0.5, uv.shape)
    b = rng.random(n) < bad
    uv[b] += rng.normal(0, 30, (int(b.sum()), 2))
    return cm.EXT_L.copy(), P, uv.astype(np.float32)


def po_rounds(tr):
    """pose-only trace split by round, the iteration made round-relative"""
    out = []
    for r in range(4):
        t = tr[(tr[:, 0] // 16) == r].copy()
        t[:, 0] -= 16 * r
        out.append(t)
    return out


def significant(tr):
    return (np.abs(tr[:, 2] - tr[:, 3]) > 1e-6 * np.abs(tr[:, 2])) & (np.abs(tr[:, 4]) > 1e-3)


def sig_prefix(tr):
    """number of leading trials that are all significant"""
    s = significant(tr)
    bad = np.nonzero(~s)[0]
    return int(bad[0]) if len(bad) else len(tr)


def assert_traces_agree(dev, ref, need_rejected=1, rtol_chi=2e-5, rtol_lam=1e-4, what=""):
    """`dev` must reproduce `ref` trial by trial on ref's significant prefix: same iteration, same accept /
    reject decision (exactly), lambda and both chi2 within tolerance.  The tolerance is that of the worst trial:
    a step at small lambda solves a reduced system of condition ~1e7-1e9, so two f64 implementations that agree
    to 1e-13 at the start are 1e-9 apart after an iteration, and the chi2 of a *rejected* step (far outside the
    trust region, chi2 several times the current one) differs by up to ~5e-6 relative — measured between the
    oracle and the independent numpy LM.  Returns (prefix length, rejected trials in it)."""
    n = sig_prefix(ref)
    nrej = int((ref[:n, 5] == 0).sum())
    assert nrej >= need_rejected, "%s: the case no longer rejects a trial inside its significant prefix (%d of %d trials)" % (what, n, len(ref))
    assert len(dev) >= n, "%s: %d trials recorded, oracle has %d significant" % (what, len(dev), n)
    d, r = dev[:n], ref[:n]
    assert np.array_equal(d[:, 0], r[:, 0]), "%s: iteration numbers differ\n%s\n%s" % (what, d[:, 0], r[:, 0])
    assert np.array_equal(d[:, 5], r[:, 5]), "%s: accept/reject decisions differ\n%s\n%s" % (what, d[:, 5], r[:, 5])
    assert np.allclose(d[:, 1], r[:, 1], rtol=rtol_lam, atol=0), "%s: lambda %s" % (what, np.abs(d[:, 1] / r[:, 1] - 1).max())
    assert np.allclose(d[:, 2], r[:, 2], rtol=rtol_chi, atol=0), "%s: chi2 before %s" % (what, np.abs(d[:, 2] / r[:, 2] - 1).max())
    fin = r[:, 3] < 1e300
    assert np.array_equal(d[:, 3] < 1e300, fin), "%s: failed solves differ" % what
    assert np.allclose(d[fin, 3], r[fin, 3], rtol=rtol_chi, atol=0), "%s: chi2 of the trial %s" % (what, np.abs(d[fin, 3] / r[fin, 3] - 1).max())
    return n, nrej


def assert_po_traces_agree(dev, ref, need_rejected=1, what=""):
    """pose-only: the four rounds restart from the prior, so every round has its own significant prefix"""
    tot = 0
    for r in range(4):
        dr = dev[(dev[:, 0] // 16) == r]; rr = ref[(ref[:, 0] // 16) == r]
        if len(rr) == 0:
            continue
        n, nrej = assert_traces_agree(dr, rr, need_rejected=0, what="%s round %d" % (what, r))
        tot += nrej
    assert tot >= need_rejected, "%s: no rejected trial inside the significant prefixes" % what
    return tot
