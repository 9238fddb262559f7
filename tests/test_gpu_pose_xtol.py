"""The parameter tolerance of the pose-only LM (svslam_set_pose_only_xtol, include/svslam.h; default 1e-12).

g2o runs the ten iterations of every optimize(10) of EstimateCurrentPose (src/frontend.cpp:482-493) whether or not the
estimate still moves; converged after four or five, it spends the rest on trials whose step is rounding noise.  The HIP
kernel ends a round at the first LM iteration whose (nearly undamped) first step has no component above xtol.  What this
file pins, through the C ABI:

  * the trials the kernel DOES run are g2o's, bit for bit: the trace of a run with the rule is a per-round prefix of the
    trace of the run without it (xtol = 0), and every trial that moves chi2 is in that prefix;
  * the result at xtol = 1e-9: pose within 1e-8 (m, quaternion component) of the full schedule's — two orders inside the LM
    tolerances against the oracle —, outlier flags and inlier counts identical, on tracking-shaped jobs and on the hard
    cases that reject trials; at the default 1e-12: within 1e-10, i.e. what another summation order moves (the other parity
    files keep checking against the oracle with the default switched on);
  * the saving: a third of all trials of tracking-shaped jobs at 1e-9 (the Huber rounds converge linearly: the rule's
    price list is in DESIGN 4.4);
  * xtol = 0 is the full schedule again after the rule was on, and the setter's range check;
  * a pipeline run with the rule and one without agree like two runs of different kernel shapes do (first frames to
    rounding, the whole run at the level of the trajectory error).
"""
import importlib

import numpy as np
import pytest

import common as cm
import lm_cases as lc

pytestmark = pytest.mark.gpu


def _run(c, jobs, xtol):
    c.pose_only_xtol(xtol)
    res = c.pose_only(jobs, cm.CAM)
    return res, [c.lm_trace(job=i) for i in range(len(jobs))]


@pytest.mark.parametrize("low_latency", [0, 1])
def test_rule_cuts_only_the_tail_of_every_round(svs, orc, low_latency):
    c = svs.Context(cm.W, cm.H, max_slots=1, max_jobs=64)
    c.lm_trace(True)
    c.low_latency(bool(low_latency))
    seeds = list(range(40)) + [100 + s for s in lc.PO_REJECT]
    jobs = [lc.po_tracking_case(s) if s < 100 else lc.po_case(s - 100) for s in seeds]
    full, tr_full = _run(c, jobs, 0.0)
    cut, tr_cut = _run(c, jobs, 1e-9)
    dflt, tr_dflt = _run(c, jobs, 1e-12)
    again, tr_again = _run(c, jobs, 0.0)
    n_full = n_cut = n_track_full = n_track_cut = 0
    worst = 0.0
    for i, seed in enumerate(seeds):
        (Tf, of, nf), (Tc, oc, nc) = full[i], cut[i]
        assert np.array_equal(full[i][0], again[i][0]) and np.array_equal(tr_full[i], tr_again[i])      # the switch switches back
        assert np.array_equal(of, oc) and nf == nc, seed
        worst = max(worst, np.abs(Tf - Tc).max())
        df, dc = lc.po_rounds(tr_full[i]), lc.po_rounds(tr_cut[i])
        for r in range(4):
            assert len(dc[r]) <= len(df[r]), (seed, r)
            assert np.array_equal(dc[r], df[r][:len(dc[r])]), (seed, r)                 # the trials that ran: the same, bit for bit
            assert len(dc[r]) >= lc.sig_prefix(df[r]), (seed, r, len(dc[r]), lc.sig_prefix(df[r]))   # nothing that moved chi2 is cut
        n_full += len(tr_full[i]); n_cut += len(tr_cut[i])
        if seed < 100:
            n_track_full += len(tr_full[i]); n_track_cut += len(tr_cut[i])
            if seed < 3:
                print("seed", seed, "trials per round, full schedule", [len(x) for x in df], "with the rule", [len(x) for x in dc])
    assert worst < 1e-8, worst
    assert n_track_cut <= 0.67 * n_track_full, (n_track_cut, n_track_full)
    worst12 = max(np.abs(full[i][0] - dflt[i][0]).max() for i in range(len(jobs)))
    n12 = sum(len(t) for t, sd in zip(tr_dflt, seeds) if sd < 100)
    for i in range(len(jobs)):
        assert np.array_equal(full[i][1], dflt[i][1]) and full[i][2] == dflt[i][2]
    assert worst12 < 1e-10 and n12 <= n_track_full, (worst12, n12)
    print("pose-only trials of the tracking-shaped jobs: full schedule %d, xtol 1e-12 %d, xtol 1e-9 %d; largest pose difference %.2e / %.2e"
          % (n_track_full, n12, n_track_cut, worst12, worst))
    with pytest.raises(RuntimeError):
        c.pose_only_xtol(1e-3)
    with pytest.raises(RuntimeError):
        c.pose_only_xtol(-1.0)
    c.close()


def test_rule_against_the_oracle_on_tracking_jobs(svs, orc):
    """with the rule (the default of a new context) the results stay inside the LM tolerances against the oracle's full schedule"""
    c = svs.Context(cm.W, cm.H, max_slots=1, max_jobs=64)
    jobs = [lc.po_tracking_case(s) for s in range(48)]
    res = c.pose_only(jobs, cm.CAM)
    for i, ((T, outl, ninl), (T0, P, uv)) in enumerate(zip(res, jobs)):
        T_ref, outl_ref, ninl_ref, _ = orc.pose_only_trace(cm.CAM, T0, P, uv)
        assert np.array_equal(outl, outl_ref) and ninl == ninl_ref, i
        assert np.allclose(T[4:], T_ref[4:], atol=1e-6) and np.allclose(T[:4], T_ref[:4], atol=1e-7), (i, np.abs(T - T_ref).max())
    c.close()


def test_pipeline_with_and_without_the_rule(svs):
    pl = importlib.import_module("stereovision-slam_amd.pipeline")
    W, H, S, N = 620, 188, 4, 64
    frames = [[svs.synth_pair(700 + s, f) for s in range(S)] for f in range(N)]

    def run(xtol):
        p = pl.Pipeline(pl.default_config(W, H, device_map=1, backend_on=1), nstreams=S)
        svs.Context.borrow(p.kernel_ctx(), W, H).pose_only_xtol(xtol)
        est = np.zeros((N, S, 7)); kf = []
        for f, pairs in enumerate(frames):
            r = p.step([x[0] for x in pairs], [x[1] for x in pairs])
            est[f] = r["pose"]; kf.append(r["is_keyframe"].copy())
        p.flush(); p.close()
        return est, np.array(kf)

    e0, k0 = run(0.0)
    e1, k1 = run(1e-9)                 # (the default 1e-12 is what every other pipeline test runs with)
    assert np.array_equal(k0[:8], k1[:8])
    # the first frames: the bounds two kernel shapes of the same sums keep (tests/test_gpu_low_latency_pipeline.py) — a pose
    # that differs in its last digits rounds a few of LK's f32 initial guesses the other way, and the local BA's free gauge
    # carries that on
    dt, dq = np.abs(e0[:8, :, 4:] - e1[:8, :, 4:]).max(), np.abs(e0[:8, :, :4] - e1[:8, :, :4]).max()
    print("with / without the rule, first 8 frames: max |dt| %.2e m, max |dq| %.2e" % (dt, dq))
    assert dt < 2e-3 and dq < 2e-4, (dt, dq)
    # the whole run: two trajectories of the same accuracy (differences feed back through LK's rounding of its initial guesses)
    for s in range(S):
        gt = np.array([svs.synth_gt(700 + s, f) for f in range(N)])
        a0, a1 = pl.ate_rmse(e0[:, s], gt), pl.ate_rmse(e1[:, s], gt)
        assert a0 < 0.1 and a1 < 0.1 and abs(a0 - a1) < 3e-2, (s, a0, a1)
        assert pl.ate_rmse(e0[:, s], e1[:, s]) < 8e-2
