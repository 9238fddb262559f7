"""GPU: Levenberg-Marquardt parity that would have caught round 2's escaped defect.

  * svslam_local_ba_batch on 12 problems captured from the host pipeline (78 % single-view landmarks: the
    single-view row pass, the largest code path of k_local_ba, is now compared with the oracle at full size);
  * per-trial LM trajectories (svslam_lm_trace vs orc_*_trace) on cases that PROVABLY reject trials — for
    k_local_ba_t<0>, the svslam_sba_* phase path (k_local_ba_t<1>) and k_pose_only<1|4>.
All through the C ABI (Context = ctypes over libsvslam_hip.so)."""
import importlib

import numpy as np
import pytest

import common as cm
import lm_cases as lc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(svs):
    c = svs.Context(cm.W, cm.H, max_slots=1, max_jobs=16, max_kf=11, max_lm=4096, max_obs=16384)
    c.lm_trace(True)
    yield c
    c.close()


def _rel(orc, P):
    return np.array([orc.se3_mul(P[k], orc.se3_inv(P[0])) for k in range(len(P))])


def test_local_ba_on_captured_pipeline_problems(ctx, orc):
    """Backend::Optimize's real problems (src/backend.cpp:39-164 gathered by the host pipeline): K = 10 with
    ~1700 landmarks / ~4000 edges and K = 7 at 613x185; tolerances of SURVEY 8d against the analytic-Jacobian
    oracle (committed answers), gauge-invariant 1e-4 against the numeric-Jacobian oracle (g2o's behaviour)."""
    probs = lc.pipeline_problems()
    for tag in ("k10", "k7"):
        sel = [p for p in probs if p["tag"] == tag]
        cam, ext_r = sel[0]["cam"], sel[0]["ext_r"]
        jobs = [(p["poses0"], p["pts0"], p["okf"], p["olm"], p["ori"], p["ouv"]) for p in sel]
        res = ctx.local_ba(jobs, cam, cm.EXT_L, cam, ext_r)
        for i, ((poses, pts, chi2, it), p, job) in enumerate(zip(res, sel, jobs)):
            assert it == int(p["iters"][0])
            assert np.allclose(poses[:, 4:], p["poses"][:, 4:], atol=1e-6), (tag, i, np.abs(poses - p["poses"]).max())
            assert np.allclose(poses[:, :4], p["poses"][:, :4], atol=1e-7)
            assert np.allclose(pts, p["pts"], rtol=1e-6, atol=1e-6), (tag, i, np.abs(pts - p["pts"]).max())
            assert np.allclose(chi2, p["chi2"], rtol=1e-5, atol=1e-6)
            tr = ctx.lm_trace(job=i)
            assert len(tr) == len(p["trace"])
            assert np.array_equal(tr[:, [0, 5]], p["trace"][:, [0, 5]])
            assert np.allclose(tr[:, 1:4], p["trace"][:, 1:4], rtol=1e-7)
            if i % 2 == 0:
                pn, xn, cn, _ = orc.local_ba(cam, cm.EXT_L, cam, ext_r, *job, jac_mode=1)
                assert np.allclose(_rel(orc, poses), _rel(orc, pn), atol=1e-4)
                assert abs(chi2.sum() - cn.sum()) <= 1e-4 * cn.sum()


def _ba_reject_cases():
    out = []
    for seed, iters in lc.BA_SYNTH_REJECT:
        out.append(("synth %d" % seed, cm.CAM, cm.EXT_R, lc.ba_synth_case(seed), iters))
    for (idx, pn, rot, of, ptn, iters) in lc.BA_PIPE_REJECT:
        cam, ext_r, job = lc.ba_pipe_case(idx, pn, rot, of, ptn)
        out.append(("pipeline %d (%.1f m, %.2f rad, %.0f %% outliers)" % (idx, pn, rot, 100 * of), cam, ext_r, job, iters))
    return out


def test_local_ba_follows_the_oracle_through_rejected_trials(ctx, orc):
    """k_local_ba_t<0>: every LM trial — lambda, chi2 before / after, accept or reject — equals the oracle's on
    the significant prefix of cases that reject trials (and accept again afterwards: the successor of a rejected
    trial must linearise at the restored state — the defect of round 2)."""
    total_rej = 0
    for name, cam, ext_r, job, iters in _ba_reject_cases():
        (poses, pts, chi2, it), = ctx.local_ba([job], cam, cm.EXT_L, cam, ext_r, iters=iters)
        tr = ctx.lm_trace(job=0)
        pr, xr, cr, itr, ref = orc.local_ba_trace(cam, cm.EXT_L, cam, ext_r, *job, iters=iters, jac_mode=0)
        n, nrej = lc.assert_traces_agree(tr, ref, need_rejected=1, what=name)
        total_rej += nrej
        if n == len(ref):            # the whole run is decided away from the rounding floor: end states must agree
            assert it == itr and len(tr) == len(ref), name
            assert np.allclose(poses[:, 4:], pr[:, 4:], atol=2e-5) and np.allclose(poses[:, :4], pr[:, :4], atol=2e-6), (name, np.abs(poses - pr).max())
            assert np.allclose(pts, xr, rtol=1e-4, atol=1e-4), (name, np.abs(pts - xr).max())
    assert total_rej >= 20


def test_local_ba_rejections_inside_a_batch(ctx, orc):
    """the same cases as ONE batch (different iteration counts are separate calls): a job's trajectory does not
    depend on its neighbours, and jobs whose trials are rejected do not disturb jobs that accept"""
    cases = [c for c in _ba_reject_cases() if c[4] == 10 and c[1] is cm.CAM]
    good = cm.make_ba_problem(np.random.default_rng(33), 7, 300)
    goodjob = (good["poses0"], good["pts0"], good["okf"], good["olm"], good["ori"], good["ouv"])
    jobs = [goodjob] + [c[3] for c in cases] + [goodjob]
    res = ctx.local_ba(jobs, cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)
    traces = [ctx.lm_trace(job=i) for i in range(len(jobs))]
    assert np.array_equal(res[0][0], res[-1][0]) and np.array_equal(res[0][1], res[-1][1]) and np.array_equal(traces[0], traces[-1])
    for i, c in enumerate(cases):
        (poses, pts, chi2, it), = ctx.local_ba([c[3]], cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)
        assert np.array_equal(poses, res[1 + i][0]) and np.array_equal(pts, res[1 + i][1]) and it == res[1 + i][3]
        assert np.array_equal(ctx.lm_trace(job=0), traces[1 + i])


def test_shared_map_phases_follow_the_oracle_through_rejected_trials(svs, orc):
    """svslam_sba_* (k_local_ba_t<1>, one LM-trial phase per launch, LM control in shared_ba.py): same
    trajectories, on one rank and with the landmarks sharded over two engines on the one device"""
    sdist = importlib.import_module("stereovision-slam_amd.dist")
    sba = importlib.import_module("stereovision-slam_amd.shared_ba")
    for seed, iters in [(1024, 10), (1013, 10), (1017, 16)]:
        poses, pts, okf, olm, ori, ouv = lc.ba_synth_case(seed)
        # the shared-map engine keeps every keyframe active and needs landmarks with edges only
        ref = orc.local_ba_trace(cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R, poses, pts, okf, olm, ori, ouv, iters=iters, jac_mode=0)
        c1 = svs.Context(cm.W, cm.H, max_slots=1, max_jobs=1, max_kf=11, max_lm=2048, max_obs=20000)
        eng = sba.HipEngine(c1, cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R, poses, pts, okf, olm, ori, ouv)
        tr = []
        it, lam = sba.shared_map_ba(eng, sdist.Rank(0, 0, 1), len(poses), iters=iters, trace=tr)
        P1, X1, C1 = eng.close()
        c1.close()
        n, nrej = lc.assert_traces_agree(np.array(tr), ref[4], need_rejected=1, what="sba seed %d" % seed)
        if n == len(ref[4]):
            assert it == ref[3]
            assert np.allclose(P1[:, 4:], ref[0][:, 4:], atol=2e-5), np.abs(P1 - ref[0]).max()


@pytest.mark.parametrize("low_latency", [0, 1])
def test_pose_only_follows_the_oracle_through_rejected_trials(svs, orc, low_latency):
    """k_pose_only<1> / <4>: the 4 x optimize(10) protocol of src/frontend.cpp:482-527 trial by trial on hard
    problems (prior 1-2 m / 0.2-0.4 rad off, points from 2.5 m, up to 50 % gross outliers)"""
    c = svs.Context(cm.W, cm.H, max_slots=1, max_jobs=16)
    c.lm_trace(True)
    c.low_latency(bool(low_latency))
    jobs = [lc.po_case(s) for s in lc.PO_REJECT]
    res = c.pose_only(jobs, cm.CAM)
    total = 0
    for i, ((T, outl, ninl), (T0, P, uv), seed) in enumerate(zip(res, jobs, lc.PO_REJECT)):
        tr = c.lm_trace(job=i)
        T_ref, outl_ref, ninl_ref, ref = orc.pose_only_trace(cm.CAM, T0, P, uv)
        total += lc.assert_po_traces_agree(tr, ref, need_rejected=1, what="pose-only seed %d" % seed)
        assert np.array_equal(outl, outl_ref) and ninl == ninl_ref, seed
        assert np.allclose(T[4:], T_ref[4:], atol=1e-6) and np.allclose(T[:4], T_ref[:4], atol=1e-7), (seed, np.abs(T - T_ref).max())
    assert total >= 30
    c.close()


@pytest.mark.parametrize("low_latency", [0, 1])
def test_pose_only_rounds_that_repeat_are_skipped_without_a_trace_of_it(svs, orc, low_latency):
    """k_pose_only does not execute a round whose outlier flags and robust flag equal the previous round's (it would
    repeat it operation for operation; tests/test_lm_rejection_cases.py checks that premise on the oracle, which runs
    every round).  Tracking-shaped jobs: where the oracle's round 2 is bitwise its round 1 the device's replayed
    trace must be too, where the flags changed the device runs the round — and poses, outlier sets and inlier counts
    equal the oracle's either way."""
    c = svs.Context(cm.W, cm.H, max_slots=1, max_jobs=32)
    c.lm_trace(True)
    c.low_latency(bool(low_latency))
    seeds = list(range(12)) + [100 + s for s in lc.PO_REJECT[:4]]
    jobs = [lc.po_tracking_case(s) if s < 100 else lc.po_case(s - 100) for s in seeds]
    res = c.pose_only(jobs, cm.CAM)
    repeated = ran = 0
    for i, ((T, outl, ninl), (T0, P, uv), seed) in enumerate(zip(res, jobs, seeds)):
        T_ref, outl_ref, ninl_ref, ref = orc.pose_only_trace(cm.CAM, T0, P, uv)
        assert np.array_equal(outl, outl_ref) and ninl == ninl_ref, seed
        assert np.allclose(T[4:], T_ref[4:], atol=1e-6) and np.allclose(T[:4], T_ref[:4], atol=1e-7), (seed, np.abs(T - T_ref).max())
        d = lc.po_rounds(c.lm_trace(job=i)); r = lc.po_rounds(ref)
        for k in (0, 1):
            ref_same = r[k].shape == r[k + 1].shape and np.array_equal(r[k], r[k + 1])
            dev_same = d[k].shape == d[k + 1].shape and np.array_equal(d[k], d[k + 1])
            if ref_same:
                assert dev_same, (seed, k)
                repeated += 1
            elif len(r[k + 1]) and lc.sig_prefix(r[k + 1]) > 0 and not dev_same:
                ran += 1
        assert len(d[3]) > 0                                   # the round without the robust kernel always runs
    assert repeated >= 6 and ran >= 6, (repeated, ran)
    c.close()
