#!/usr/bin/env python3
"""How much would a misrecollected upstream convention matter?  (VERDICT r3 "next round" #8)

The oracle restates OpenCV 4.5.4 / g2o from published descriptions (oracle/svs_oracle.h: PARITY UNPINNED; SURVEY.md
Appendix A lists the recalled conventions).  For every convention that no independent test pins, the oracle has a
what-if knob (orc_set_whatif) that flips it to its plausible alternative.  This script runs the CPU twin of the host
pipeline (reference-shaped host logic over the oracle kernels, numeric BA Jacobians like g2o) over N seeded streams of
F frames with each knob flipped, one at a time, and reports against the oracle as declared:

  corners   first-frame GFTT lists that change (the detector alone, no mask / with a 40-point mask)
  identical streams whose whole trajectory stays bit-identical
  dpose     translation difference of the last frame's pose, median and maximum over the streams [m]
  ATE       trajectory RMSE against the renderer's ground truth: mean of the variant, paired difference to the
            declared oracle with its standard error, relative to the declared oracle's mean
  kf        keyframes per stream (mean), frames LOST

CPU only (about ten minutes on 8 cores).  Test infrastructure: drives oracle/ only, never the product.

  python tests/oracle_sensitivity.py [n_streams] [n_frames] > profiles/r4_oracle_sensitivity.txt"""
import ctypes as C
import importlib
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "tests")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)

KNOBS = [
    (0, "LK normal-equation sums in f32, pixel order (declared: exact integer sums, one cast)", "A.5"),
    (1, "g2o LM tau = 1e-3 (declared: 1e-5)", "A.6"),
    (2, "g2o rho denominator without + 1e-3 (declared: with)", "A.6"),
    (3, "Sobel scale 1/12 (declared: 1/3060 = 1/(4*3*255))", "A.2"),
    (4, "mask rectangle corners rounded half-up (declared: half-to-even)", "A.1"),
    (5, "box filter of the eigenvalue map in f32 (declared: f64 sums, one cast)", "A.2"),
    (6, "equal corner responses: address ascending (declared: descending)", "A.2"),
    (7, "Huber kernel bends at chi2 > 5.991 (declared: at sqrt(chi2) > 5.991, i.e. chi2 > 35.9)", "A.6"),
]


def main():
    n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    import oracle_lib as orc
    import pipe_cpu
    svs = importlib.import_module("stereovision-slam_amd")
    pl = importlib.import_module("stereovision-slam_amd.pipeline")
    twin = pipe_cpu.twin_lib()
    olib = orc.lib()
    for L in (twin, olib):
        L.orc_set_whatif.argtypes = [C.c_int, C.c_int]
        L.orc_set_whatif.restype = None
    threads = max(1, len(os.sched_getaffinity(0)))
    seeds = [0x5EED2000 + i for i in range(n_streams)]
    W, H = 620, 188
    t0 = time.time()
    left = np.zeros((n_streams, n_frames, H, W), np.uint8)
    right = np.zeros_like(left)

    def render(t):
        for s in range(t, n_streams, threads):
            for f in range(n_frames):
                left[s, f], right[s, f] = svs.synth_pair(seeds[s], f)
    th = [threading.Thread(target=render, args=(t,)) for t in range(threads)]
    [t_.start() for t_ in th]; [t_.join() for t_ in th]
    gt = np.array([[svs.synth_gt(seeds[s], f) for f in range(n_frames)] for s in range(n_streams)])
    print("# %d streams x %d frames of the synthetic KITTI-00-shaped stereo stream (seeds 0x5EED2000 ...), rendered in %.0f s; CPU twin with numeric BA Jacobians" %
          (n_streams, n_frames, time.time() - t0))

    def set_knob(k, v):
        for L in (twin, olib):
            L.orc_set_whatif(k, v)

    def run_all():
        poses = np.zeros((n_streams, n_frames, 7)); kf = np.zeros(n_streams); lost = np.zeros(n_streams)
        errs = []

        def work(t):
            try:
                for s in range(t, n_streams, threads):
                    p = pipe_cpu.make(nstreams=1)
                    for f in range(n_frames):
                        r = p.step([left[s, f]], [right[s, f]])
                        poses[s, f] = r["pose"][0]
                        lost[s] += int(r["status"][0] == 3)
                    kf[s] = p.counters()["keyframes"]
                    p.close()
            except Exception as e:   # noqa: BLE001
                errs.append(e)
        th = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
        [t_.start() for t_ in th]; [t_.join() for t_ in th]
        if errs:
            raise errs[0]
        ate = np.array([pl.ate_rmse(poses[s], gt[s]) for s in range(n_streams)])
        return poses, ate, kf, lost

    def corners(mask_pts):
        out = []
        for s in range(n_streams):
            c = orc.gftt(left[s, 0], mask_pts[s] if mask_pts is not None else None)
            out.append(c)
        return out

    base_c0 = corners(None)
    mask_pts = [c[:40] + np.float32(0.5) for c in base_c0]        # points on half-pixel positions: the rounding rule decides
    base_c1 = corners(mask_pts)
    t0 = time.time()
    bp, bate, bkf, blost = run_all()
    print("# the oracle as declared: ATE %.4f m (mean over the streams), %.1f keyframes per stream, %d frames LOST; one pass %.0f s" %
          (bate.mean(), bkf.mean(), int(blost.sum()), time.time() - t0))
    print("%-3s %-92s %-9s %-10s %-20s %-38s %-12s" % ("#", "what if (SURVEY appendix)", "corners", "identical", "dpose med / max [m]",
                                                      "ATE mean, paired diff +- s.e. (rel.)", "kf / LOST"))
    for k, what, app in KNOBS:
        set_knob(k, 1)
        try:
            c0 = corners(None); c1 = corners(mask_pts)
            ch0 = sum(1 for a, b in zip(c0, base_c0) if not (a.shape == b.shape and np.array_equal(a, b)))
            ch1 = sum(1 for a, b in zip(c1, base_c1) if not (a.shape == b.shape and np.array_equal(a, b)))
            p, ate, kf, lost = run_all()
        finally:
            set_knob(k, 0)
        ident = int(sum(np.array_equal(p[s], bp[s]) for s in range(n_streams)))
        dpo = np.linalg.norm(p[:, -1, 4:] - bp[:, -1, 4:], axis=1)
        d = ate - bate
        print("%-3d %-92s %3d / %-3d %4d / %-3d %8.2e / %8.2e  %.4f  %+.4f +- %.4f (%+.2f %%)          %.1f / %d" %
              (k, "%s [%s]" % (what, app), ch0, ch1, ident, n_streams, np.median(dpo), dpo.max(), ate.mean(), d.mean(),
               d.std(ddof=1) / np.sqrt(n_streams), 100 * d.mean() / bate.mean(), kf.mean(), int(lost.sum())))
        sys.stdout.flush()
    print("# corners: streams (of %d) whose first-frame GFTT list changes, without mask / with a mask of 40 points on half-pixel positions" % n_streams)


if __name__ == "__main__":
    main()
