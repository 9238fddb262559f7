#!/usr/bin/env python3
"""Where does the small, consistently negative HIP - twin ATE difference come from?

Rounds 3-5 measured HIP minus twin = -0.38 ... -0.78 % of the twin's mean ATE (s.e. 0.44-0.62 %) — five runs, all negative, all
over the SAME seeded streams.  This tool separates the candidates: the twin's side is run ONCE per seed set and kept; the HIP side
is run per library variant / setting on the same frames:
    base         the product build
    ieee         -DSVS_IEEE_DIV: IEEE divisions and roots instead of estimate + two Newton steps in the LM kernels
    nocontract   -DSVS_NO_CONTRACT: the f64 LM code without FMA contraction (and the explicit fma calls as mul + add)
    xtol0        the product build with SVSLAM_PO_XTOL=0 (g2o's pose-only schedule to the last trial)
    all3         ieee + nocontract + xtol0
and a SECOND, disjoint seed set answers whether the sign belongs to the streams or to the arithmetic.

  python tests/ate_bias.py twin  <seed0> <streams> <frames> <out.npz>        (needs the GPU: renders the frames)
  python tests/ate_bias.py hip   <seed0> <streams> <frames> <out.npz>        (the library currently in stereovision-slam_amd/lib)
  python tests/ate_bias.py report <twin.npz> <tag=hip.npz> ...

Test infrastructure (drives the CPU twin); tools/ate_bias.sh is the box-side driver."""
import importlib
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "tests")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)
W, H = 620, 188


def run_side(side, seed0, n_streams, n_frames, chunk=40, batch=2048, device=0):
    svs = importlib.import_module("stereovision-slam_amd")
    pl = importlib.import_module("stereovision-slam_amd.pipeline")
    ate = np.zeros(n_streams); kf = 0
    for b0 in range(0, n_streams, batch):
        nb = min(batch, n_streams - b0)
        seeds = [seed0 + b0 + i for i in range(nb)]
        gpu = pl.Pipeline(pl.default_config(W, H, host_threads=4, device_map=1), nstreams=nb if side == "hip" else 1, device=device)
        ctx = svs.Context.borrow(gpu.kernel_ctx(), W, H)
        img = W * H
        dl = ctx.dev_alloc(nb * chunk * img); dr = ctx.dev_alloc(nb * chunk * img)
        est = np.zeros((n_frames, nb, 7))
        if side == "twin":
            import pipe_cpu
            os.environ.pop("SVS_ORACLE_BA_JAC", None)             # numeric Jacobians: the reference-faithful twin
            twins = [pipe_cpu.make(nstreams=1) for _ in seeds]
            left = np.zeros((nb, chunk, H, W), np.uint8); right = np.zeros_like(left)
            threads = max(1, min(nb, len(os.sched_getaffinity(0))))
        for f0 in range(0, n_frames, chunk):
            n = min(chunk, n_frames - f0)
            svs.synth_render_streams_device(seeds, f0, chunk, W, H, dl, dr, device=device)
            if side == "hip":
                est[f0:f0 + n] = gpu.run_device(dl, dr, chunk * img, img, 0, n)["pose"]
                continue
            ctx.dev_download(dl, left); ctx.dev_download(dr, right)
            errs = []

            def work(t):
                try:
                    for s in range(t, nb, threads):
                        for f in range(n):
                            est[f0 + f, s] = twins[s].step([left[s, f]], [right[s, f]])["pose"][0]
                except Exception as e:   # noqa: BLE001
                    errs.append(e)
            th = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
            if errs:
                raise errs[0]
        ctx.dev_free(dl); ctx.dev_free(dr)
        for s, sd in enumerate(seeds):
            gt = np.array([svs.synth_gt(sd, f) for f in range(n_frames)])
            ate[b0 + s] = pl.ate_rmse(est[:, s], gt)
        kf += gpu.counters()["keyframes"] if side == "hip" else sum(t.counters()["keyframes"] for t in twins)
        gpu.close()
        if side == "twin":
            for t in twins:
                t.close()
        print("# %s: %d streams from seed 0x%X done" % (side, nb, seed0 + b0), file=sys.stderr, flush=True)
    return ate, kf


def report(twin_npz, hips):
    from ate_distribution import bootstrap
    t = np.load(twin_npz)
    b = t["ate"]
    print("twin (numeric-J BA, reference-faithful): %d streams x %d frames from seed 0x%X, mean ATE %.5f m, keyframes %d"
          % (len(b), int(t["frames"]), int(t["seed0"]), b.mean(), int(t["keyframes"])))
    print("%-12s %10s %22s %26s %16s %10s" % ("HIP variant", "mean ATE", "HIP - twin [% of twin]", "bootstrap 95 % CI", "better / worse", "keyframes"))
    base = None
    for tag, path in hips:
        h = np.load(path)
        a = h["ate"]
        assert len(a) == len(b) and int(h["seed0"]) == int(t["seed0"])
        d, lo, hi, se = bootstrap(a, b)
        print("%-12s %10.5f %+12.2f +- %.2f %+14.2f ... %+.2f %9d / %d %12d  %s"
              % (tag, a.mean(), 100 * d, 100 * se, 100 * lo, 100 * hi, int((a < b).sum()), int((a > b).sum()), int(h["keyframes"]), str(h["library"])))
        if base is None:
            base = a
        else:
            dd = a - base
            print("%-12s vs %s: paired %+.3f %% +- %.3f %% of the twin's mean; %d streams bit-identical in ATE"
                  % ("", hips[0][0], 100 * dd.mean() / b.mean(), 100 * dd.std(ddof=1) / np.sqrt(len(dd)) / b.mean(), int((dd == 0).sum())))


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "report":
        report(sys.argv[2], [a.split("=", 1) for a in sys.argv[3:]])
        sys.exit(0)
    seed0, ns, nf, out = int(sys.argv[2], 0), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    ate, kf = run_side(mode, seed0, ns, nf)
    svs = importlib.import_module("stereovision-slam_amd")
    np.savez(out, ate=ate, keyframes=kf, seed0=seed0, frames=nf, library=svs.load().svslam_build_info().decode(),
             po_xtol=os.environ.get("SVSLAM_PO_XTOL", "default"))
    print("%s: %d streams, mean ATE %.5f m, keyframes %d -> %s" % (mode, ns, ate.mean(), kf, out))
