#!/usr/bin/env python3
"""kernel micro-benchmarks on the GPU box: per-family HIP-event times for a batch of
jobs, plus the per-phase cycle profile of the BA kernel.  Development tool."""
import importlib
import sys
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common as cm

svs = importlib.import_module("stereovision-slam_amd")


def ba(nj=1, nkf=10, nlm=700, reps=3, ll=0):
    rng = np.random.default_rng(1)
    if ll:
        os.environ["SVSLAM_LL_SHARDS"] = str(ll)
    c = svs.Context(cm.W, cm.H, max_slots=1, max_jobs=max(nj, 1), max_kf=max(nkf, 10) + 1, max_lm=4096, max_obs=16384)
    if ll:
        c.low_latency(True)
        print("low-latency solver, %d shards per problem" % ll)
    if nkf == 0:
        # a local-BA problem captured from the pipeline itself (stream seed 3, full 10-keyframe window:
        # 1725 landmarks of which 78% are seen from one keyframe only, 4011 edges), replicated
        d = np.load(os.path.join(ROOT, "tools", "ba_pipeline_problem.npz"))
        jobs = [(d["poses"], d["pts"], d["okf"], d["olm"], d["ori"], d["uv"])] * nj
        nkf, nlm = len(d["poses"]), len(d["pts"])
    else:
        probs = [cm.make_ba_problem(rng, nkf, nlm) for _ in range(nj)]
        # thin the observations to a realistic ~4 per landmark
        jobs = []
        for p in probs:
            keep = rng.random(len(p["okf"])) < 0.3
            jobs.append((p["poses0"], p["pts0"], p["okf"][keep], p["olm"][keep], p["ori"][keep], p["ouv"][keep]))
    print("BA jobs=%d nkf=%d nlm=%d nobs=%d" % (nj, nkf, nlm, len(jobs[0][2])))
    c.ba_profile(True)
    c.timing(True)
    for r in range(reps):
        c.local_ba(jobs, cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)
        ms, n, _ = c.timing_get("local_ba")
        prof = c.ba_profile(True, read=True)
        c.timing(True)
        if ll and r == 0:
            print("  shards (landmarks, edges, blocks, tiles, solver 2=resident):", c.ll_shards(1)[0][:, :5].tolist())
        names = ["edge+J", "lm+pose", "dinv/Y/Sinit", "schur", "chol", "backsub", "errors"]
        tot = sum(prof[:7])
        print("  rep %d: %.3f ms/launch; trials %d; phase us (100MHz ticks/100): " % (r, ms / max(n, 1), prof[11]) +
              ", ".join("%s %.1f" % (nm, prof[i] / 100.0) for i, nm in enumerate(names)) + "  sum %.1f  [chol factor part %.1f; tile pass %.1f, tile schur diag %.1f + offdiag %.1f]" % ((tot + prof[8] + prof[9] + prof[10]) / 100.0, prof[7] / 100.0, prof[8] / 100.0, prof[10] / 100.0, prof[9] / 100.0))
    c.close()


def _inputs(npts, min_dist, stereo=False):
    """corners of the left image (and their stereo matches / landmarks) from the HIP kernels themselves: the tools
    do not touch the oracle"""
    l0, r0 = svs.synth_pair(3, 0)
    c = svs.Context(cm.W, cm.H, max_slots=2, max_jobs=2, max_pts=max(npts, 8), max_corners=max(npts, 8), max_kf=0, max_lm=0, max_obs=0)
    c.pyramid([0, 1], [l0, r0])
    pts = c.gftt([(0, None)], max_corners=npts, min_dist=min_dist)[0]
    out = (pts,)
    if stereo:
        q, st, _ = c.lk([(0, 1, pts, pts)])[0]
        xyz, ok = c.triangulate([(pts, q, None, 0.0)], cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)[0]
        out = (pts, q, st, xyz, ok)
    c.close()
    return out


def frontend(nj=1, npts=230):
    l0, r0 = svs.synth_pair(3, 0); l1, _ = svs.synth_pair(3, 1)
    pts, q, st, xyz, ok = _inputs(npts, 8.0, stereo=True)
    c = svs.Context(cm.W, cm.H, max_slots=3 * nj, max_jobs=3 * nj, max_kf=0, max_lm=0, max_obs=0)
    c.pyramid(list(range(3 * nj)), [l0, r0, l1] * nj)
    m = (st > 0) & (ok > 0)
    c.timing(True)
    for r in range(3):
        c.pyramid(list(range(0, 3 * nj, 3)), [l0] * nj)
        c.lk([(3 * i, 3 * i + 2, pts, pts) for i in range(nj)])
        c.gftt([(3 * i, pts[:80]) for i in range(nj)])
        q1 = c.lk([(3 * i, 3 * i + 2, pts, pts) for i in range(nj)])[0][0]
        c.pose_only([(cm.EXT_L, xyz[m], q1[m]) for i in range(nj)], cm.CAM)
        c.triangulate([(pts, q, None, 0.0) for i in range(nj)], cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)
    print("frontend jobs=%d npts=%d (edges %d): " % (nj, len(pts), m.sum()) +
          ", ".join("%s %.1f us" % (f, 1e3 * c.timing_get(f)[0] / max(c.timing_get(f)[1], 1)) for f in svs.FAMILIES if c.timing_get(f)[1]))
    c.close()


def lkbench(nj=512, npts=150):
    """LK throughput at bench scale: nj jobs x npts points, temporal and stereo pairs, by iteration cap."""
    l0, r0 = svs.synth_pair(3, 0); l1, _ = svs.synth_pair(3, 1)
    pts, = _inputs(npts, 20.0)
    c = svs.Context(cm.W, cm.H, max_slots=3, max_jobs=nj, max_pts=max(len(pts), 8), max_kf=0, max_lm=0, max_obs=0)
    c.pyramid([0, 1, 2], [l0, r0, l1])
    for name, dst in (("temporal", 2), ("stereo", 1)):
        for mc in ((int(os.environ["LKBENCH_MC"]),) if "LKBENCH_MC" in os.environ else (1, 2, 4, 30)):
            prm = svs.LkParams(3, mc, 0.01, 1e-4, 1)
            c.lk([(0, dst, pts, pts)] * nj, prm)
            c.timing_reset() if hasattr(c, "timing_reset") else None
            c.timing(True)
            for r in range(5):
                out = c.lk([(0, dst, pts, pts)] * nj, prm)
            t = c.timing_get("lk")
            print("lk %s jobs=%d pts=%d max_count=%2d: %.1f us/launch (%.2f ns/point), ok %.2f" %
                  (name, nj, len(pts), mc, 1e3 * t[0] / max(t[1], 1), 1e6 * t[0] / max(t[1], 1) / (nj * len(pts)), out[0][1].mean()))
            c.timing(False)
    c.close()


def gfttbench(nj=512):
    """GFTT and pyramid throughput at bench scale: nj images per launch, no mask / 80 / 230 mask squares"""
    l0, r0 = svs.synth_pair(3, 0)
    pts, = _inputs(230, 8.0)
    c = svs.Context(cm.W, cm.H, max_slots=nj, max_jobs=nj, max_pts=256, max_kf=0, max_lm=0, max_obs=0)
    c.pyramid(list(range(nj)), [l0 if i % 2 == 0 else r0 for i in range(nj)])
    for nr in (0, 80, 230):
        jobs = [(i, pts[:nr] if nr else None) for i in range(nj)]
        out = c.gftt(jobs)
        c.timing(True)
        for r in range(5):
            out = c.gftt(jobs)
        t = c.timing_get("gftt")
        if t[1] == 0:       # SVSLAM_TIMING_SPLIT=1: eig3 / select2 separately
            a, b = c.timing_get("dbg0"), c.timing_get("dbg1")
            print("gftt jobs=%d rects=%3d: eig3 %.1f us + select2 %.1f us per launch" % (nj, nr, 1e3 * a[0] / max(a[1], 1), 1e3 * b[0] / max(b[1], 1)))
            continue
        print("gftt jobs=%d rects=%3d: %.1f us/launch (%.3f us/image), corners %.1f" %
              (nj, nr, 1e3 * t[0] / max(t[1], 1), 1e3 * t[0] / max(t[1], 1) / nj, np.mean([len(o) for o in out])))
        c.timing(False)
    c.timing(True)
    for r in range(5):
        c.pyramid(list(range(nj)), [l0] * nj)
    t = c.timing_get("pyramid")
    print("pyramid jobs=%d (host images, upload not timed): %.1f us/launch (%.3f us/image)" % (nj, 1e3 * t[0] / max(t[1], 1), 1e3 * t[0] / max(t[1], 1) / nj))
    c.close()


def clock():
    import ctypes as C
    c = svs.Context(cm.W, cm.H, max_slots=1, max_jobs=1, max_kf=0, max_lm=0, max_obs=0)
    for blocks, ms in ((1, 0.05), (1, 1.0), (1, 20.0), (256, 1.0), (2048, 20.0), (1, 0.05)):
        mhz = C.c_double()
        c.L.svslam_debug_clock_mhz(c.h, blocks, C.c_double(ms), C.byref(mhz))
        print("clock probe: %4d blocks, %.2f ms spin -> %.0f MHz" % (blocks, ms, mhz.value))
    c.close()


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what == "lk":
        lkbench(); sys.exit(0)
    if what == "gftt":
        gfttbench(); sys.exit(0)
    if what == "ball":       # the captured pipeline problem on the low-latency solver (one problem over 4 / 8 / 16 workgroups)
        for w in (4, 8, 16):
            ba(1, 0, 0, reps=3, ll=w)
        ba(4, 0, 0, reps=2, ll=8)
        ba(1, 0, 0, reps=2)
        sys.exit(0)
    if what == "ba1":        # the captured pipeline problem: alone and 256 at a time, with the phase profile
        ba(1, 0, 0, reps=2); ba(256, 0, 0, reps=2); sys.exit(0)
    if what == "ba2":        # chip throughput of the BA kernel: more problems than CUs
        for nj in (256, 512, 1024):
            ba(nj, 0, 0, reps=2)
        sys.exit(0)
    if what == "tput":       # chip-time per family at bench scale
        for nj in (1, 64, 256, 512):
            ba(nj, 0, 0, reps=2)
        frontend(256); frontend(512)
        sys.exit(0)
    if what in ("clock", "all"):
        clock()
    if what in ("ba", "all"):
        ba(1, 10, 700); ba(1, 7, 300); ba(16, 10, 700)
    if what in ("fe", "all"):
        frontend(1); frontend(16); frontend(64)
