#!/usr/bin/env python3
"""ATE distribution of the HIP pipeline against its reference-faithful CPU twin (numeric BA
Jacobians like g2o, pyramids rebuilt per LK call) over many seeded streams.

Over a sequence the two pipelines are chaotic in each other (LK's stopping rule is
discontinuous in its float start guess, the first chi2 within rounding of 5.991 flips an outlier
bit), so north_star's "ATE within 1 % of the reference" can only be a statement about the
DISTRIBUTION of the trajectory error, not about single runs.  This tool measures it: per stream
ATE (RMSE after rigid alignment) against the renderer's ground truth for both paths, the paired
difference, and a bootstrap confidence interval of the relative difference of the means.

  python tests/ate_distribution.py [n_streams] [n_frames] > profiles/r2_ate_distribution.txt

Used by tests/test_gpu_ate_distribution.py (-m gpu).  The twin is test infrastructure."""
import importlib
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "tests")):  # (this file lives in tests/: it drives the CPU twin, test infrastructure)
    if p_ not in sys.path:
        sys.path.insert(0, p_)

W, H = 620, 188


def run(n_streams=48, n_frames=320, seed0=0x5EED1000, chunk=80, threads=None, device=0, twin_jacobians="numeric", device_map=1,
        whatif=None, whatif_streams=0):
    """returns dict(ate_hip, ate_twin, ate_between, path_len) as arrays over streams.  twin_jacobians: "numeric"
    (g2o's central differences, what the reference runs) or "analytic" (isolates the effect of that choice).
    whatif = k: a second twin with the oracle's what-if knob k flipped (oracle/svs_oracle.h) runs the first whatif_streams
    streams on the same frames -> out["ate_whatif"] (the knob is a process global of the twin library: the two twins
    take turns per chunk of frames, never concurrently)"""
    import ctypes as C
    import pipe_cpu
    svs = importlib.import_module("stereovision-slam_amd")
    pl = importlib.import_module("stereovision-slam_amd.pipeline")
    if twin_jacobians == "analytic":
        os.environ["SVS_ORACLE_BA_JAC"] = "0"
    else:
        os.environ.pop("SVS_ORACLE_BA_JAC", None)                 # twin: numeric Jacobians (reference-faithful)
    seeds = [seed0 + i for i in range(n_streams)]
    threads = threads or max(1, min(n_streams, len(os.sched_getaffinity(0))))
    # (device_map: the map of every stream in HBM, the bench's default; bit-identical to the host-resident map)
    gpu = pl.Pipeline(pl.default_config(W, H, host_threads=min(4, threads), device_map=device_map), nstreams=n_streams, device=device)
    ctx = svs.Context.borrow(gpu.kernel_ctx(), W, H)
    twins = [pipe_cpu.make(nstreams=1) for _ in seeds]
    n_wi = min(n_streams, whatif_streams) if whatif is not None else 0
    twins_wi = [pipe_cpu.make(nstreams=1) for _ in range(n_wi)]
    tl = pipe_cpu.twin_lib()
    tl.orc_set_whatif.argtypes = [C.c_int, C.c_int]; tl.orc_set_whatif.restype = None
    ew = np.zeros((n_frames, n_wi, 7))
    img = W * H
    dl = ctx.dev_alloc(n_streams * chunk * img); dr = ctx.dev_alloc(n_streams * chunk * img)
    eg = np.zeros((n_frames, n_streams, 7)); ec = np.zeros((n_frames, n_streams, 7))
    left = np.zeros((n_streams, chunk, H, W), np.uint8); right = np.zeros_like(left)
    for f0 in range(0, n_frames, chunk):
        n = min(chunk, n_frames - f0)
        # the same rendered frames for both paths: rendered on the device, downloaded for the twin
        svs.synth_render_streams_device(seeds, f0, chunk, W, H, dl, dr, device=device)
        ctx.dev_download(dl, left); ctx.dev_download(dr, right)
        eg[f0:f0 + n] = gpu.run_device(dl, dr, chunk * img, img, 0, n)["pose"]
        errs = []

        def work(t, tw, dst, ns):
            try:
                for s in range(t, ns, threads):
                    for f in range(n):
                        dst[f0 + f, s] = tw[s].step([left[s, f]], [right[s, f]])["pose"][0]
            except Exception as e:   # noqa: BLE001
                errs.append(e)
        for tw, dst, ns, knob in ((twins, ec, n_streams, 0), (twins_wi, ew, n_wi, 1)):
            if ns == 0:
                continue
            if whatif is not None:
                tl.orc_set_whatif(int(whatif), knob)
            th = [threading.Thread(target=work, args=(t, tw, dst, ns)) for t in range(threads)]
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
            if whatif is not None:
                tl.orc_set_whatif(int(whatif), 0)
            if errs:
                raise errs[0]
    ctx.dev_free(dl); ctx.dev_free(dr)
    out = {k: np.zeros(n_streams) for k in ("ate_hip", "ate_twin", "ate_between", "path_len")}
    for s, sd in enumerate(seeds):
        gt = np.array([svs.synth_gt(sd, f) for f in range(n_frames)])
        out["path_len"][s] = float(np.linalg.norm(np.diff(pl.camera_centres(gt), axis=0), axis=1).sum())
        out["ate_hip"][s] = pl.ate_rmse(eg[:, s], gt)
        out["ate_twin"][s] = pl.ate_rmse(ec[:, s], gt)
        out["ate_between"][s] = pl.ate_rmse(eg[:, s], ec[:, s])
    if n_wi:
        out["ate_whatif"] = np.array([pl.ate_rmse(ew[:, s], np.array([svs.synth_gt(seeds[s], f) for f in range(n_frames)]))
                                      for s in range(n_wi)])
        out["whatif_identical"] = int(sum(np.array_equal(ew[:, s], ec[:, s]) for s in range(n_wi)))
    out["keyframes"] = (gpu.counters()["keyframes"], sum(t.counters()["keyframes"] for t in twins))
    gpu.close()
    for t in twins + twins_wi:
        t.close()
    return out


def run_batched(n_streams, n_frames, batch=2048, **kw):
    """run() over batches of streams (bounds host memory: frames of a chunk x streams live on the host for the twin);
    stream i has seed seed0 + i whatever the batch size"""
    seed0 = kw.pop("seed0", 0x5EED1000)
    wi_total = kw.pop("whatif_streams", 0)
    outs = []
    for b0 in range(0, n_streams, batch):
        nb = min(batch, n_streams - b0)
        outs.append(run(nb, n_frames, seed0=seed0 + b0, whatif_streams=max(0, min(nb, wi_total - b0)), **kw))
        print("# batch of %d streams from %d done" % (nb, b0), file=sys.stderr, flush=True)
    out = {}
    for k in outs[0]:
        if k == "keyframes":
            out[k] = tuple(sum(o[k][i] for o in outs) for i in range(2))
        elif k == "whatif_identical":
            out[k] = sum(o.get(k, 0) for o in outs)
        else:
            out[k] = np.concatenate([o[k] for o in outs if k in o])
    return out


def report_whatif(r, what):
    """paired difference twin-with-knob minus twin-as-declared over the streams that ran both"""
    w = r["ate_whatif"]; b = r["ate_twin"][:len(w)]
    d = w - b
    _, lo, hi, se = bootstrap(w, b)
    return ("what-if '%s': %d streams, twin as declared mean ATE %.4f m, with the knob %.4f m; paired difference %+.5f +- %.5f m "
            "= %+.2f %% +- %.2f %% of the declared twin's mean (bootstrap 95 %% CI [%+.2f %%, %+.2f %%]); %d streams bit-identical, "
            "%d better / %d worse with the knob"
            % (what, len(w), b.mean(), w.mean(), d.mean(), d.std(ddof=1) / np.sqrt(len(w)), 100 * d.mean() / b.mean(),
               100 * d.std(ddof=1) / np.sqrt(len(w)) / b.mean(), 100 * lo, 100 * hi, r.get("whatif_identical", 0),
               int((d < 0).sum()), int((d > 0).sum())))


def bootstrap(a_hip, a_twin, n_boot=20000, seed=1):
    """paired bootstrap over streams of d = (mean hip - mean twin) / mean twin; returns
    (d, lo95, hi95, standard error)"""
    rng = np.random.default_rng(seed)
    n = len(a_hip)
    idx = rng.integers(0, n, (n_boot, n))
    d = (a_hip[idx].mean(1) - a_twin[idx].mean(1)) / a_twin[idx].mean(1)
    d0 = (a_hip.mean() - a_twin.mean()) / a_twin.mean()
    lo, hi = np.percentile(d, [2.5, 97.5])
    return float(d0), float(lo), float(hi), float(d.std())


def report(r, n_frames, twin="reference-faithful CPU twin (numeric-J BA)"):
    a, b, L = r["ate_hip"], r["ate_twin"], r["path_len"]
    d, lo, hi, se = bootstrap(a, b)
    lines = ["ATE distribution, HIP pipeline vs %s, %d streams x %d frames "
             "(%.0f m mean path), config-00 parameters" % (twin, len(a), n_frames, L.mean()),
             "%-28s %10s %10s" % ("", "HIP", "CPU twin"),
             "%-28s %10.4f %10.4f" % ("mean ATE [m]", a.mean(), b.mean()),
             "%-28s %10.4f %10.4f" % ("median ATE [m]", np.median(a), np.median(b)),
             "%-28s %10.4f %10.4f" % ("p10 ATE [m]", np.percentile(a, 10), np.percentile(b, 10)),
             "%-28s %10.4f %10.4f" % ("p90 ATE [m]", np.percentile(a, 90), np.percentile(b, 90)),
             "%-28s %10.4f %10.4f" % ("max ATE [m]", a.max(), b.max()),
             "%-28s %9.4f%% %9.4f%%" % ("mean ATE / path length", 100 * (a / L).mean(), 100 * (b / L).mean()),
             "paired difference of the means: %+.2f %% of the twin's mean ATE; bootstrap 95 %% CI [%+.2f %%, %+.2f %%], "
             "standard error %.2f %%" % (100 * d, 100 * lo, 100 * hi, 100 * se),
             "difference of the path-normalised means: %+.5f %% of the path" % (100 * ((a / L).mean() - (b / L).mean())),
             "per stream |ATE_hip - ATE_twin| median %.4f m; HIP vs twin trajectories after alignment: median %.4f m, max %.4f m"
             % (np.median(np.abs(a - b)), np.median(r["ate_between"]), r["ate_between"].max()),
             "streams where HIP is better / worse than the twin: %d / %d; keyframes HIP %d, twin %d"
             % (int((a < b).sum()), int((a > b).sum()), r["keyframes"][0], r["keyframes"][1])]
    return "\n".join(lines)


if __name__ == "__main__":
    ns = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    nf = int(sys.argv[2]) if len(sys.argv) > 2 else 320
    if "--analytic-only" in sys.argv:
        # the SAME-ALGORITHM twin (analytic Jacobians like the HIP kernel) on the big sample —
        # any difference beyond the chaos of two valid runs would be a defect of the kernel, not of g2o's differentiation
        res = run(ns, nf, twin_jacobians="analytic", chunk=40)
        print(report(res, nf, "CPU twin with analytic BA Jacobians (same algorithm as the HIP kernel)"))
        sys.exit(0)
    batch = 2048
    for a in sys.argv:
        if a.startswith("--batch="):
            batch = int(a.split("=")[1])
    wi = [a for a in sys.argv if a.startswith("--whatif=")]          # --whatif=<knob>:<streams>
    if wi or ns > batch:
        kw = {}
        if wi:
            k_, n_ = wi[0].split("=")[1].split(":")
            kw = dict(whatif=int(k_), whatif_streams=int(n_))
        res = run_batched(ns, nf, batch=batch, chunk=40, **kw)
        print(report(res, nf))
        if wi:
            print()
            print(report_whatif(res, "oracle knob %s (tests/oracle_sensitivity.py KNOBS)" % k_))
        sys.exit(0)
    res = run(ns, nf)
    print(report(res, nf))
    if "--analytic-too" in sys.argv:
        # the same streams against the twin with ANALYTIC BA Jacobians: what is left is the chaos of two
        # equally valid runs, the difference to the line above is the numeric differentiation of g2o
        res2 = run(ns, nf, twin_jacobians="analytic")
        print()
        print(report(res2, nf, "CPU twin with analytic BA Jacobians"))
    print()
    print("per-stream ATE [m] hip / twin (numeric-J):")
    print(" ".join("%.3f/%.3f" % (x, y) for x, y in zip(res["ate_hip"], res["ate_twin"])))
