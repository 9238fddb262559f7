"""GPU: the low-latency local BA — ONE problem dealt over several workgroups (k_ba_split + k_local_ba_t<2, W>,
svslam_set_low_latency) — against the same oracle answers, at the same tolerances, as the batch kernel
(tests/test_gpu_lm_parity.py), for 4 / 8 / 16 shards; plus what only this path has: shards without edges, a
problem smaller than its shard count, keyframes no shard sees, several problems per call.
Backend::Optimize, reference src/backend.cpp:22-164.  All through the C ABI."""
import os

import numpy as np
import pytest

import common as cm
import lm_cases as lc

pytestmark = pytest.mark.gpu


def _sorted_job(job):
    """(landmark, keyframe) order — the order the backend gathers edges in (src/backend.cpp:83-160); the
    low-latency path only takes problems in that order (anything else goes to the batch kernel)"""
    poses, pts, okf, olm, ori, ouv = job
    o = np.lexsort((okf, olm))
    return poses, pts, okf[o], olm[o], ori[o], ouv[o]


def _make_ctx(svs, shards, resident):
    """resident = 1: problems whose shards all fit LDS go to k_ba_ll, the others to k_local_ba_t<2>; 0: all to the latter"""
    old = {k: os.environ.get(k) for k in ("SVSLAM_LL_SHARDS", "SVSLAM_LL_RESIDENT")}
    os.environ["SVSLAM_LL_SHARDS"] = str(shards)
    os.environ["SVSLAM_LL_RESIDENT"] = str(resident)
    try:
        c = svs.Context(cm.W, cm.H, max_slots=1, max_jobs=16, max_kf=11, max_lm=4096, max_obs=16384)
        c.low_latency(True)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    c.lm_trace(True)
    c.resident = resident
    return c


@pytest.fixture(scope="module", params=[(16, 1), (16, 0), (8, 1), (4, 0)], ids=lambda p: "%dshards-%s" % (p[0], "resident" if p[1] else "streaming"))
def ctx(svs, request):
    c = _make_ctx(svs, *request.param)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx_batch(svs):
    c = svs.Context(cm.W, cm.H, max_slots=1, max_jobs=16, max_kf=11, max_lm=4096, max_obs=16384)
    yield c
    c.close()


def _took_ll(ctx, n):
    assert ctx.host_counters()[6] == n, "the low-latency solver did not take the call"


def test_ll_on_captured_pipeline_problems(ctx, orc):
    """the committed answers of the 12 captured pipeline problems (K = 10: ~1700 landmarks / ~4000 edges; K = 7),
    tolerances of SURVEY 8d, every LM trial equal to the oracle's"""
    probs = lc.pipeline_problems()
    ctx.host_counters()
    for tag in ("k10", "k7"):
        sel = [p for p in probs if p["tag"] == tag]
        cam, ext_r = sel[0]["cam"], sel[0]["ext_r"]
        for i, p in enumerate(sel):
            job = (p["poses0"], p["pts0"], p["okf"], p["olm"], p["ori"], p["ouv"])
            assert np.array_equal(np.lexsort((job[2], job[3])), np.arange(len(job[2]))), "captured problems are landmark-major"
            (poses, pts, chi2, it), = ctx.local_ba([job], cam, cm.EXT_L, cam, ext_r)
            _took_ll(ctx, 1)
            sh = ctx.ll_shards(1)[0]
            if not ctx.resident:
                assert np.all(sh[:, 4] == 1)
            elif sh.shape[0] == 16:
                assert np.all(sh[:, 4] == 2), "a K <= 10 window over 16 workgroups fits the resident layout: %s" % sh[:, :5].tolist()
            assert it == int(p["iters"][0])
            assert np.allclose(poses[:, 4:], p["poses"][:, 4:], atol=1e-6), (tag, i, np.abs(poses - p["poses"]).max())
            assert np.allclose(poses[:, :4], p["poses"][:, :4], atol=1e-7)
            assert np.allclose(pts, p["pts"], rtol=1e-6, atol=1e-6), (tag, i, np.abs(pts - p["pts"]).max())
            assert np.allclose(chi2, p["chi2"], rtol=1e-5, atol=1e-6)
            tr = ctx.lm_trace(job=0)
            assert len(tr) == len(p["trace"])
            assert np.array_equal(tr[:, [0, 5]], p["trace"][:, [0, 5]])
            assert np.allclose(tr[:, 1:4], p["trace"][:, 1:4], rtol=1e-7)


def test_ll_equals_the_batch_kernel(ctx, ctx_batch):
    """same problems through both solvers: the sums are taken in another order (per shard, then over the shards),
    nothing else differs — agreement far inside the oracle tolerances; several problems in one low-latency call"""
    probs = [p for p in lc.pipeline_problems() if p["tag"] == "k10"][:6]
    cam, ext_r = probs[0]["cam"], probs[0]["ext_r"]
    jobs = [(p["poses0"], p["pts0"], p["okf"], p["olm"], p["ori"], p["ouv"]) for p in probs]
    ctx.host_counters()
    a = ctx.local_ba(jobs, cam, cm.EXT_L, cam, ext_r)
    _took_ll(ctx, len(jobs))
    b = ctx_batch.local_ba(jobs, cam, cm.EXT_L, cam, ext_r)
    for (pa, xa, ca, ia), (pb, xb, cb, ib) in zip(a, b):
        assert ia == ib
        assert np.allclose(pa, pb, atol=1e-9), np.abs(pa - pb).max()
        assert np.allclose(xa, xb, rtol=1e-8, atol=1e-8), np.abs(xa - xb).max()
        assert np.allclose(ca, cb, rtol=1e-6, atol=1e-8)
    # and it is deterministic: the same call again gives the same bits
    a2 = ctx.local_ba(jobs, cam, cm.EXT_L, cam, ext_r)
    for (pa, xa, ca, ia), (pb, xb, cb, ib) in zip(a, a2):
        assert np.array_equal(pa, pb) and np.array_equal(xa, xb) and np.array_equal(ca, cb) and ia == ib


def _ba_reject_cases():
    out = []
    for seed, iters in lc.BA_SYNTH_REJECT:
        out.append(("synth %d" % seed, cm.CAM, cm.EXT_R, _sorted_job(lc.ba_synth_case(seed)), iters))
    for (idx, pn, rot, of, ptn, iters) in lc.BA_PIPE_REJECT:
        cam, ext_r, job = lc.ba_pipe_case(idx, pn, rot, of, ptn)
        out.append(("pipeline %d (%.1f m, %.2f rad, %.0f %% outliers)" % (idx, pn, rot, 100 * of), cam, ext_r, _sorted_job(job), iters))
    return out


def test_ll_follows_the_oracle_through_rejected_trials(ctx, orc):
    """every LM trial — lambda, chi2 before / after, accept or reject — equals the oracle's on cases that reject
    trials: the shards take the same decision from the same all-shard sums"""
    total_rej = 0
    ctx.host_counters()
    for name, cam, ext_r, job, iters in _ba_reject_cases():
        (poses, pts, chi2, it), = ctx.local_ba([job], cam, cm.EXT_L, cam, ext_r, iters=iters)
        _took_ll(ctx, 1)
        tr = ctx.lm_trace(job=0)
        pr, xr, cr, itr, ref = orc.local_ba_trace(cam, cm.EXT_L, cam, ext_r, *job, iters=iters, jac_mode=0)
        n, nrej = lc.assert_traces_agree(tr, ref, need_rejected=1, what=name)
        total_rej += nrej
        if n == len(ref):
            assert it == itr and len(tr) == len(ref), name
            assert np.allclose(poses[:, 4:], pr[:, 4:], atol=2e-5) and np.allclose(poses[:, :4], pr[:, :4], atol=2e-6), (name, np.abs(poses - pr).max())
            assert np.allclose(pts, xr, rtol=1e-4, atol=1e-4), (name, np.abs(pts - xr).max())
    assert total_rej >= 20


def test_ll_small_and_ragged_problems(ctx, orc):
    """fewer landmarks than shards (shards without edges are masked out), landmarks without edges, a keyframe
    that nothing observes (every shard keeps all keyframes active: its pose must not move), unsorted edges
    (handed to the batch kernel)"""
    rng = np.random.default_rng(5)
    ctx.host_counters()
    # 7 landmarks, 4 keyframes: fewer landmarks than shards at 8 and 16 shards
    p = cm.make_ba_problem(rng, 4, 7, outlier_frac=0.0)
    job = _sorted_job((p["poses0"], p["pts0"], p["okf"], p["olm"], p["ori"], p["ouv"]))
    (poses, pts, chi2, it), = ctx.local_ba([job], cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)
    _took_ll(ctx, 1)
    assert int(np.count_nonzero(ctx.ll_shards(1)[0][:, 1])) <= 7
    pr, xr, cr, itr = orc.local_ba(cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R, *job, jac_mode=0)
    assert it == itr
    assert np.allclose(poses[:, 4:], pr[:, 4:], atol=1e-6) and np.allclose(poses[:, :4], pr[:, :4], atol=1e-7), np.abs(poses - pr).max()
    assert np.allclose(pts, xr, rtol=1e-6, atol=1e-6) and np.allclose(chi2, cr, rtol=1e-5, atol=1e-6)
    # landmarks without edges (first, middle, last) and a keyframe without edges
    p = cm.make_ba_problem(rng, 6, 200)
    okf, olm, ori, ouv = p["okf"], p["olm"], p["ori"], p["ouv"]
    keep = (okf != 2) & ~np.isin(olm, (0, 77, 199))
    job = _sorted_job((p["poses0"], p["pts0"], okf[keep], olm[keep], ori[keep], ouv[keep]))
    (poses, pts, chi2, it), = ctx.local_ba([job], cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)
    _took_ll(ctx, 1)
    pr, xr, cr, itr = orc.local_ba(cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R, *job, jac_mode=0)
    assert it == itr
    assert np.allclose(poses[2], p["poses0"][2], atol=1e-15), "an unobserved keyframe moved"
    for l in (0, 77, 199):
        assert np.array_equal(pts[l], p["pts0"][l]), "a landmark without edges moved"
    assert np.allclose(poses[:, 4:], pr[:, 4:], atol=1e-6) and np.allclose(poses[:, :4], pr[:, :4], atol=1e-7)
    assert np.allclose(pts, xr, rtol=1e-6, atol=1e-6) and np.allclose(chi2, cr, rtol=1e-5, atol=1e-6)
    # unsorted edges: not this path's business
    job_u = (p["poses0"], p["pts0"], okf[keep], olm[keep], ori[keep], ouv[keep])
    (poses_u, pts_u, chi2_u, it_u), = ctx.local_ba([job_u], cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)
    assert ctx.host_counters()[6] == 0
    assert it_u == itr and np.allclose(poses_u, pr, atol=1e-6)
    # eight different problems in one call
    jobs = []
    for s in range(8):
        q = cm.make_ba_problem(np.random.default_rng(100 + s), 5 + s % 4, 150 + 40 * s)
        jobs.append(_sorted_job((q["poses0"], q["pts0"], q["okf"], q["olm"], q["ori"], q["ouv"])))
    res = ctx.local_ba(jobs, cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)
    _took_ll(ctx, 8)
    for (poses, pts, chi2, it), job in zip(res, jobs):
        pr, xr, cr, itr = orc.local_ba(cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R, *job, jac_mode=0)
        assert it == itr
        assert np.allclose(poses[:, 4:], pr[:, 4:], atol=1e-6) and np.allclose(poses[:, :4], pr[:, :4], atol=1e-7)
        assert np.allclose(pts, xr, rtol=1e-6, atol=1e-6)
