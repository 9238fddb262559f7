#!/usr/bin/env python3
"""bench.py — throughput of the stereo-SLAM hot path on MI355X.

A "step" is one pass of the hot path (Frontend::AddFrame: fused pyramid + LK +
pose-only every frame; GFTT + stereo LK + triangulation + local BA
on keyframes) over one batch of S synthetic stereo frames — one new frame for
each of the S independent streams a rank owns.  Frames are rendered into HBM
by the HIP generator before the timed region, so `value` is whole-job
frames/s with inputs resident in HBM.  One process per GPU; streams are
independent (no data-path collective); N>1 is weak scaling.

  python bench.py --gpus 1 --steps 200 --warmup 10
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# One HIP stream per host thread: by default ROCm multiplexes all streams of a process onto 4
# hardware queues, which serialises one group's ~3 ms BA kernel with the other groups' tracking.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

W, H = 620, 188                     # KITTI-00 1241x376 after the reference's 1/2 decimation (F3)
HBM_PEAK_GBS = 8000.0               # spec (MI355X_MICROARCH.md)
HBM_PEAK_MEASURED_GBS = 6290.0      # float4 copy on MI355X (MI355X_MICROARCH.md chip table); SURVEY 8d's denominator
LEVEL_PIX = (620 * 188, 310 * 94, 155 * 47, 78 * 24)


def algorithmic_bytes(fam, cnt, launches):
    """SURVEY.md §8d per-unit figures x the units the timed launches processed."""
    if fam == "lk":          # 4 levels x (14x14 I patch + 20x20 J region) + 21 B point I/O
        return (cnt["track_pts"] + cnt["right_pts"]) * (2384 + 21)
    if fam == "pose_only":   # 40 B per edge (xyz f64 + uv f32 + flags) + 56 B pose
        return cnt["pose_edges"] * 40 + launches * 56
    if fam == "pyramid":     # image read once + levels 1..3 written once
        return (cnt["pyr_left"] + cnt["pyr_right"]) * sum(LEVEL_PIX)
    if fam == "gftt":        # image read once + rect list + corners out
        return cnt["gftt_calls"] * LEVEL_PIX[0] + cnt["gftt_rects"] * 8 + cnt["corners"] * 8
    if fam == "triangulate":
        return cnt["tri_pts"] * (16 + 24 + 1)
    if fam == "local_ba":    # iters x (E x 40 B obs+ids + K x 56 B + M x 24 B)
        it = max(cnt["ba_iters"], 1) / max(cnt["ba_calls"], 1)
        return it * (cnt["ba_edges"] * 40 + cnt["ba_kf"] * 56 + cnt["ba_lm"] * 24)
    return 0


def measured_traffic(fam, cnt, launches):
    """HBM bytes per launch of the family from the committed PMC passes (profiles/pmc_traffic.json,
    collected with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs); None if unknown."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(fam)
    except (OSError, ValueError):
        return None
    if not t or not t.get("bytes"):
        return None
    units = {"local_ba": cnt["ba_calls"], "lk": cnt["track_pts"] + cnt["right_pts"], "pose_only": cnt["frames"],
             "gftt": cnt["gftt_calls"], "pyramid": cnt["pyr_left"] + cnt["pyr_right"], "triangulate": cnt["tri_pts"]}.get(fam)
    if not units or not launches:
        return None
    return t["bytes"] * units / launches


F64_VECTOR_PEAK_TFLOPS = 78.6       # MI355X f64 vector peak (half the f32 vector rate, MI355X_MICROARCH.md chip table)
SIMDS = 256 * 4


def compute_side(fam, cnt, fam_ms):
    """The compute-side ruler of the two kernels that own most of the step (the HBM roofline says
    nothing about a kernel bound by f64 latency or by integer issue).
      local_ba: counted f64 flops of the timed problems / the family's HIP-event time, against the f64 vector peak.
                Flop model: per LM trial E * 500 (residual, Jacobians, normal-equation sums of an edge)
                + P * 324 (a block pair of the Schur complement: Y = W Dinv, Y W^T) + (6K)^3 / 3 (Cholesky), with
                E edges, P block pairs, K keyframes and the trial count as the library logged them for the timed
                problems (svslam_dmap_job::ba_npair / ba_ntrial).
      lk:       VALU wave-instructions issued / time, against the issue peak of its instruction class
                (v_dot2 / v_perm / v_mad_i24 class: 1.75 ns per wave-instruction per SIMD at full occupancy,
                profiles/r3_ubench_valu_issue_rates.txt); instructions per point from the committed PMC pass.
    valu_busy comes from the committed PMC passes (profiles/pmc_valu.json: tools/pmc_ba.sh, tools/pmc_lk.sh), like
    `traffic` it is a constant of the build, not something this run measured."""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_valu.json")))
    except (OSError, ValueError):
        pmc = {}
    t = fam_ms / 1e3
    if t <= 0:
        return None
    if fam == "local_ba":
        calls = max(cnt["ba_calls"], 1)
        trials = cnt.get("ba_trials", 0) or cnt["ba_iters"]
        k = cnt["ba_kf"] / calls
        flops = (trials / calls) * (cnt["ba_edges"] * 500.0 + cnt.get("ba_pairs", 0) * 324.0 + calls * (6 * k) ** 3 / 3.0)
        p = pmc.get("local_ba", {})
        return {"kernel": "local_ba", "flops": round(flops), "flops_per_problem": round(flops / calls),
                "lm_trials_per_problem": round(trials / calls, 2), "block_pairs_per_problem": round(cnt.get("ba_pairs", 0) / calls, 1),
                "achieved_tflops": round(flops / t / 1e12, 3), "peak_tflops": F64_VECTOR_PEAK_TFLOPS,
                "frac": round(flops / t / 1e12 / F64_VECTOR_PEAK_TFLOPS, 5),
                "valu_busy": p.get("valu_busy"), "valu_insts_per_problem": p.get("valu_insts_per_unit"),
                "valu_source": p.get("source", "no committed PMC pass")}
    if fam == "lk":
        p = pmc.get("lk", {})
        ipp = p.get("valu_insts_per_unit")
        pts = cnt["track_pts"] + cnt["right_pts"]
        if not ipp or not pts:
            return None
        peak = SIMDS / 1.75e-9 / 1e9           # G wave-instructions / s
        ach = ipp * pts / t / 1e9
        return {"kernel": "lk", "valu_wave_insts": round(ipp * pts), "valu_insts_per_point": ipp,
                "achieved_ginst_s": round(ach, 1), "peak_ginst_s": round(peak, 1), "frac": round(ach / peak, 4),
                "valu_busy": p.get("valu_busy"), "valu_source": p.get("source", "no committed PMC pass")}
    return None


def valu_step(cnt, elapsed_s):
    """The whole step on the ruler that fits it (round 5): the step is VALU-issue work — k_lk alone is ~64 % of all VALU
    wave-instructions, the local BA ~17 % — so the meaningful roofline of the JOB is instructions issued per second against the
    chip's VALU issue peak.  Instructions = the timed region's units x the per-unit SQ_INSTS_VALU constants of the committed
    PMC pass (profiles/pmc_valu_step.json, tools/pmc_valu_step.sh); peak = 1024 SIMDs / 1.75 ns (the 4-cycle instruction class
    k_lk is made of; plain f32 / u32 adds issue faster, f64 slower: the fraction is against LK's class)."""
    try:
        c = json.load(open(os.path.join(ROOT, "profiles", "pmc_valu_step.json")))
    except (OSError, ValueError):
        return None
    units = {"local_ba": cnt["ba_calls"], "lk": cnt["track_pts"] + cnt["right_pts"], "pose_only": cnt["frames"],
             "pyramid": cnt["pyr_left"] + cnt["pyr_right"], "gftt": cnt["gftt_calls"], "triangulate": cnt["tri_pts"], "map": cnt["keyframes"]}
    per = {f: units[f] * c[f]["valu_insts"] for f in units if f in c}
    tot = sum(per.values())
    if tot <= 0 or elapsed_s <= 0:
        return None
    peak = SIMDS / 1.75e-9 / 1e9
    return {"valu_wave_insts_per_frame": round(tot / max(cnt["frames"], 1)), "achieved_ginst_s": round(tot / elapsed_s / 1e9, 1),
            "peak_ginst_s": round(peak, 1), "frac": round(tot / elapsed_s / 1e9 / peak, 4),
            "share": {f: round(v / tot, 3) for f, v in per.items()},
            "source": "profiles/pmc_valu_step.json (committed PMC pass, per-unit SQ_INSTS_VALU) x the units of this run's first timed window, per GPU"}


def effective_cpus():
    """CPUs this process may actually burn: min(affinity, cgroup v2 cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return n


def thread_cpu_seconds():
    """CPU seconds per thread name of this process (Linux /proc), for the host-cost breakdown"""
    out = {}
    tick = os.sysconf("SC_CLK_TCK")
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                comm = open("/proc/self/task/%s/comm" % tid).read().strip()
                f = open("/proc/self/task/%s/stat" % tid).read().rsplit(")", 1)[1].split()
                out[comm] = out.get(comm, 0.0) + (int(f[11]) + int(f[12])) / tick
            except (OSError, IndexError, ValueError):
                pass
    except OSError:
        pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("SVS_BENCH_STREAMS", "0")),
                    help="independent stereo streams per GPU, advanced in lockstep (0 = the largest of 12288 / 8192 / "
                         "6144 whose frame buffers fit the GPU's memory for this --warmup + --steps)")
    ap.add_argument("--groups", type=int, default=int(os.environ.get("SVS_BENCH_GROUPS", "0")),
                    help="host threads per GPU, each driving streams/groups streams through its own "
                         "svslam context (own HIP stream): one group's BA overlaps the others' tracking")
    ap.add_argument("--host-threads", type=int, default=int(os.environ.get("SVS_BENCH_HOST_THREADS", "0")),
                    help="threads per group for the per-stream host bookkeeping (Frontend/Map/Backend glue)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pin", action="store_true", help="do not restrict the host threads to the GPU's NUMA node")
    ap.add_argument("--low-latency", action="store_true",
                    help="latency shape of the serial kernels (svslam_set_low_latency); for a few streams per GPU")
    ap.add_argument("--backend-mode", type=int, default=1, choices=(1, 2),
                    help="1 (default): local BA completes before the next frame; 2: it runs beside the next "
                         "frame like the reference's backend thread and lands exactly one frame late "
                         "(measured: no throughput gain, the GPU is already saturated by the other streams)")
    ap.add_argument("--backend-lag", type=int, default=1,
                    help="--backend-mode 2: frames a local BA may stay in flight before its result is applied (1 = one frame, "
                         "round 2; 6 hides a lone camera's 1.3-ms BA behind the next frames' tracking)")
    ap.add_argument("--cpu-frames", type=int, default=1200, help="timed frames per CPU-baseline thread (after its pre-roll)")
    ap.add_argument("--ring-frames", type=int, default=0,
                    help="frames per stream kept in HBM (0 = warmup + steps: one timed block).  Smaller than warmup + steps: "
                         "the timed region runs in blocks, every block bracketed by barrier + synchronize, the frames of the "
                         "next block rendered in between OUTSIDE the timing — for long runs (--steps 10000)")
    ap.add_argument("--preroll", type=int, default=-1,
                    help="untimed steps before the warm-up so that the timed region is the steady state (every stream's "
                         "active window holds num_active_keyframes keyframes); -1 = automatic: blocks of warmup+steps "
                         "frames until the local-BA problems of a block average a full window; 0 = none")
    ap.add_argument("--spread-windows", type=int, default=5,
                    help="further windows of --steps steps timed after the reported one in the same process (each behind "
                         "its own untimed render + barrier): the line carries their values as value_spread (0 = none)")
    ap.add_argument("--host-input-steps", type=int, default=10,
                    help="steps of an extra leg that reads the frames from PINNED HOST memory (the pyramid kernel pulls "
                         "them over PCIe, no staging copy): value_host_input, the rate a host-buffer boundary gets (0 = skip)")
    ap.add_argument("--solo-steps", type=int, default=6,
                    help="steps of an extra leg in which ONE group runs alone on the GPU: every kernel of its chain then "
                         "has the chip to itself, so the HIP-event durations are solo durations (roofline_solo; 0 = skip)")
    ap.add_argument("--host-map", action="store_true",
                    help="keep every stream's map (window, features, landmarks, observations) on the HOST as in rounds 1-2; "
                         "default: the map lives in HBM and the keyframe path is one chain of kernels (svslam_dmap_*) — "
                         "bit-identical results, a fraction of the host CPU")
    ap.add_argument("--full-res-streams", type=int, default=-1,
                    help="streams of the value_full_res leg: after the reported run the script runs itself once more with the "
                         "frames stored at the camera's 1241x376 (BASELINE's metric names that size) and the 1/2 decimation "
                         "fused into the pyramid, at most 20 + 5 steps so that the full-size frame ring fits (0 = skip; -1 = 8192 when "
                         "--streams is left to the script, i.e. the headline operating point, else skip; one GPU only, never under torchrun's N > 1)")
    ap.add_argument("--full-res", action="store_true",
                    help="keep the frames in HBM at the camera's 1241x376 and fuse the reference's 1/2 "
                         "decimation (Dataset::NextFrame) into the pyramid's level 0 (SURVEY 8 row f3); 4x the "
                         "frame bytes, so fewer streams fit")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: the launch contract only — rank discovery, stream partition, barriers, max-over-ranks timing, "
                         "the JSON line (dry_run: true) — with a rank-dependent sleep as the step; what tests/test_abi_and_host.py "
                         "runs at world 8 over gloo on a CPU box")
    args = ap.parse_args()
    if args.dry_run:
        return dry_run(args)
    if args.full_res_streams < 0:         # (a small --streams run of a test or a latency script must not spawn an 8192-stream child)
        args.full_res_streams = 8192 if args.streams <= 0 else 0

    import torch
    svs = importlib.import_module("stereovision-slam_amd")
    pl = importlib.import_module("stereovision-slam_amd.pipeline")
    if os.environ.get("SVS_MALLOC_TUNING", "1") == "1":
        pl.tune_allocator()
    sdist = importlib.import_module("stereovision-slam_amd.dist")
    rk = sdist.init(os.environ.get("SVS_DIST_BACKEND", "nccl"))   # RCCL; one process per GPU
    rank, local_rank, world = rk.rank, rk.local_rank, rk.world
    if "SVS_FORCE_DEVICE" in os.environ:            # dry run of the N-rank path on a 1-GPU box (with gloo)
        local_rank = int(os.environ["SVS_FORCE_DEVICE"])
    if world == 1 and torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    svs.load()                                      # fails loudly if the HIP library is missing

    S, Wm, K = args.streams, args.warmup, args.steps
    # The synthetic frames live in HBM.  One buffer of FB = warmup + steps frames per stream is rendered
    # block by block (pre-roll blocks first, then the block that holds the warm-up and the timed steps),
    # always outside the timed region: keep it under ~190 GB.
    SW, SH = (1241, 376) if args.full_res else (W, H)     # stored frame size
    FB = Wm + K if args.ring_frames <= 0 else max(Wm + 1, min(Wm + K, args.ring_frames))
    budget = 225e9                                   # of the MI355X's 288 GB; the pyramids and work buffers need ~1 MB per stream
    if torch.cuda.is_available():
        budget = min(budget, 0.8 * torch.cuda.mem_get_info(local_rank)[0])
    cap = int(budget // (2 * SW * SH * FB + (1 << 20)))
    if S <= 0:
        # more streams per launch fill the chip better (measured: 6144 / 8192 / 12288 streams = 1.00 / 1.04 / 1.08),
        # the frame ring of warmup + steps frames per stream decides what fits
        S = next((c for c in (12288, 8192, 6144) if c <= cap), cap)
        # a rank that may use only a few cores (N ranks sharing one CPU quota) cannot feed that many streams:
        # ~27 us of host CPU per frame; keep its memory footprint in proportion
        # (only with the map on the host: the device-resident map costs the host < 1 core per 12 288 streams)
        lw = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        if args.host_map:
            S = min(S, 1024 * max(1, effective_cpus() // max(1, lw)))
    if S > cap:
        S = max(512, cap // 512 * 512) if cap >= 512 else max(1, cap)
    # host layout from the cores this rank may actually use (cgroup quota / ranks on the node):
    # about two threads per core (half of them are waiting on the GPU at any time), at most 12
    # groups x at most 4 bookkeeping threads
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    cores = max(1, effective_cpus() // max(1, local_world))
    pinned = set() if args.no_pin else sdist.pin_to_device_numa(local_rank, min_cpus=cores)
    # (about 1024 streams per group: 8 groups up to 8192 streams, 12 beyond; tools/sweep.sh)
    dev_map = not args.host_map       # (round 4: also with the backend beside the frontend, --backend-mode 2)
    # with the map on the device a group's thread only enqueues and waits: the group count follows the GPU (kernels
    # of different groups overlap), not the cores, and one bookkeeping thread per group is plenty
    # (measured, tools/sweep_devmap.sh: 4 / 6 / 8 / 12 / 16 groups at 12 288 streams = 483 / 487 / 500 / 495 / 480 k frames/s,
    # 24 groups 450 k — the GPU is the limit whatever the layout; 6 keeps fewer kernels co-resident per launch)
    # (round 5, profiles/r5_bench_groups_sweep.txt: 2 / 3 / 4 / 6 / 8 groups = 487 / 503 / 538 / 542 / 546 k frames/s — flat from 4 on;
    #  with 4 fewer kernels share the chip per launch, so the per-family HIP-event durations the roofline objects are computed
    #  from are closer to what a kernel takes: local_ba roofline.frac 0.016 at 4 groups, 0.013 at 6, 0.009 at 8)
    G = args.groups if args.groups > 0 else (4 if dev_map else min(12 if S >= 12288 else 8, cores))
    G = max(1, min(G, S))
    S -= S % G                                       # whole groups (8192 streams in 12 groups: 12 x 682)
    if args.host_threads <= 0:
        args.host_threads = 1 if dev_map else max(1, min(4, (2 * cores + G - 1) // G))
    Sg = S // G
    cfg = pl.default_config(W, H, host_threads=max(1, args.host_threads), backend_on=args.backend_mode,
                            src_width=SW if args.full_res else 0, src_height=SH if args.full_res else 0,
                            low_latency=1 if args.low_latency else 0,
                            device_map=0 if args.host_map else 1, backend_lag=max(1, args.backend_lag))
    pipes = [pl.Pipeline(cfg, nstreams=Sg, device=local_rank) for _ in range(G)]
    ctxs = [svs.Context.borrow(p.kernel_ctx(), W, H) for p in pipes]   # alloc / timing through the pipelines' contexts
    ctx = ctxs[0]
    sep_backend = args.backend_mode == 2 and not dev_map      # host map: the backend has its own context (second HIP stream) per pipeline
    if sep_backend:
        ctxs = ctxs + [svs.Context.borrow(p.backend_ctx(), W, H) for p in pipes]

    img = SW * SH
    cam_r = tuple(2 * v for v in svs.KITTI00_HALF_CAM) if args.full_res else svs.KITTI00_HALF_CAM
    d_left = ctx.dev_alloc(S * FB * img)
    d_right = ctx.dev_alloc(S * FB * img)
    seeds = list(rk.stream_seeds(S))
    twin_of_0 = (S // G) * (G - 1) if G > 1 else (S - 1 if S > 1 else 0)   # first stream of the last group
    if twin_of_0 > 0:
        seeds[twin_of_0] = seeds[0]     # one deliberate duplicate: must come out bit-identical (checked below)

    def render_block(frame0):
        """frames [frame0, frame0 + FB) of every stream -> HBM, layout [stream][FB][img]"""
        svs.synth_render_streams_device(seeds, frame0, FB, SW, SH, d_left, d_right, device=local_rank, cam=cam_r)

    def barrier():
        rk.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    import threading

    def run_all(first, nframes, want, lbase=None, rbase=None, fb=None, groups=None, out_bufs=None):
        """every group advances its streams by nframes steps; groups run concurrently (ctypes
        releases the GIL inside the C++ loop), each on its own HIP stream.  lbase / rbase / fb: another frame
        store laid out [stream][fb][img] (the pinned host ring of the host-input leg); groups: a subset"""
        outs = [None] * G
        errs = []
        lb = d_left if lbase is None else lbase
        rb = d_right if rbase is None else rbase
        fbn = FB if fb is None else fb

        def work(g):
            try:
                import ctypes
                ctypes.CDLL(None).prctl(15, b"svs-group", 0, 0, 0)      # PR_SET_NAME, for the CPU breakdown
                base = g * Sg * fbn * img
                outs[g] = pipes[g].run_device(lb + base, rb + base, fbn * img, img, first, nframes,
                                              want_results=want, out=None if out_bufs is None else out_bufs[g])
                pipes[g].flush()     # a backend optimisation still in flight completes inside the timed region
            except Exception as e:   # noqa: BLE001
                errs.append(e)
        th = [threading.Thread(target=work, args=(g,)) for g in (range(G) if groups is None else groups)]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        if errs:
            raise errs[0]
        return outs

    def counters_sum():
        tot = {}
        for p in pipes:
            for k, v in p.counters().items():
                tot[k] = tot.get(k, 0) + v
        return tot

    # ---- pre-roll (untimed): StereoInit of every stream, then frames until the sliding window of
    #      local BA is full everywhere, so that the timed region is the steady-state workload whatever
    #      --warmup / --steps are
    pre = 0
    window_full = cfg.num_active_keyframes
    preroll_kf = 0.0
    if args.preroll != 0:
        while True:
            n = FB if args.preroll < 0 else min(FB, args.preroll - pre)
            if n <= 0:
                break
            cb = counters_sum()
            render_block(pre)
            run_all(0, n, False)
            pre += n
            ca = counters_sum()
            calls = ca["ba_calls"] - cb["ba_calls"]
            preroll_kf = (ca["ba_kf"] - cb["ba_kf"]) / max(calls, 1)
            if args.preroll < 0 and (preroll_kf >= window_full - 0.05 or pre >= 400):
                break
    # ---- warmup, untimed
    render_block(pre)
    run_all(0, Wm, False)
    c0 = counters_sum()
    for c in ctxs:
        c.timing(True)
    # the per-frame results of the timed region (88 B per frame: the bench's own log) are allocated and touched here, so
    # that rss_growth_bytes_per_frame below is the growth of the pipeline / library, not of this script's arrays
    res_bufs = [np.zeros((K, Sg), pl.RESULT_DTYPE) for _ in range(G)]
    for b_ in res_bufs:
        b_.view(np.uint8).fill(0)
    # the K timed steps: one block when the ring holds warmup + steps frames (the default), else block by block — each
    # block between barrier + synchronize on both sides, the next block's frames rendered outside the timing
    blocks, first_, left_ = [], Wm, K
    while left_ > 0:
        n_ = min(FB - first_, left_)
        blocks.append((first_, n_))
        left_ -= n_
        first_ = 0
    import resource
    rss0 = int(open("/proc/self/statm").read().split()[1]) * os.sysconf("SC_PAGE_SIZE")
    t_timed = cpu_timed = 0.0
    minflt_timed = 0
    tc_acc = {}
    done_ = 0
    for bi, (first_, n_) in enumerate(blocks):
        if bi > 0:
            render_block(pre + FB + (bi - 1) * FB)
        barrier()
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        tc0 = thread_cpu_seconds()
        t0 = time.perf_counter()
        run_all(first_, n_, True, out_bufs=[b_[done_:done_ + n_] for b_ in res_bufs])
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        t1 = time.perf_counter()
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        tc1 = thread_cpu_seconds()
        barrier()
        t_timed += t1 - t0
        cpu_timed += (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
        minflt_timed += ru1.ru_minflt - ru0.ru_minflt
        for k in tc1:
            tc_acc[k] = tc_acc.get(k, 0.0) + tc1[k] - tc0.get(k, 0.0)
        done_ += n_
    res_g = res_bufs
    rss1 = int(open("/proc/self/statm").read().split()[1]) * os.sysconf("SC_PAGE_SIZE")
    cpu_by_thread = {k: round(v / max(t_timed, 1e-9), 2) for k, v in tc_acc.items() if v > 0.005 * t_timed}
    cpu_busy = cpu_timed / max(t_timed, 1e-9)
    t0, t1 = 0.0, t_timed                                # (the code below uses t1 - t0)
    elapsed = rk.max_over_ranks(t_timed)
    if len(blocks) > 1:                                  # the extra legs assume a ring of warmup + steps frames
        args.spread_windows = 0
        args.host_input_steps = 0
    c1 = counters_sum()
    cnt = {k: c1[k] - c0[k] for k in c1}
    fam_t = {}
    for f in svs.FAMILIES:
        parts = [c.timing_get(f) for c in ctxs]
        fam_t[f] = (sum(p[0] for p in parts), sum(p[1] for p in parts), sum(p[2] for p in parts))
    for c in ctxs:
        c.timing(False)
    frame_pos = pre + FB + (len(blocks) - 1) * FB  # next frame of every stream (a multi-block run leaves part of the last ring unused)

    # ---- value_spread: further windows of K steps, same process, same operating point (the 0.6-s
    #      window of the driver's 20 steps scatters by +-10 % from run to run; here is the scatter inside one run)
    spread = []
    for _ in range(max(0, args.spread_windows)):
        render_block(frame_pos)
        barrier()
        ts0 = time.perf_counter()
        run_all(0, K, False)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        ts1 = time.perf_counter()
        barrier()
        spread.append(S * K * world / rk.max_over_ranks(ts1 - ts0))
        frame_pos += K

    # ---- value_host_input: the same step with the frames in pinned host memory (what a host-buffer boundary
    #      hands over): k_pyr_fused reads them across PCIe itself, nothing is staged or copied by the CPU
    host_input = None
    Kh = max(0, min(args.host_input_steps, FB))
    if world > 1:
        Kh = 0          # a one-GPU figure (PCIe of one device; 2 x 14 GB of pinned host memory per rank): not taken at N > 1
    if Kh > 0:
        try:
            svs.synth_render_streams_device(seeds, frame_pos, Kh, SW, SH, d_left, d_right, device=local_rank, cam=cam_r)
            hl = torch.empty(S * Kh * img, dtype=torch.uint8, pin_memory=True)
            hr = torch.empty(S * Kh * img, dtype=torch.uint8, pin_memory=True)
            ctx.L.svslam_dev_download(ctx.h, _vp(hl.data_ptr()), _vp(d_left), S * Kh * img)
            ctx.L.svslam_dev_download(ctx.h, _vp(hr.data_ptr()), _vp(d_right), S * Kh * img)
            barrier()
            th0 = time.perf_counter()
            run_all(0, Kh, False, lbase=hl.data_ptr(), rbase=hr.data_ptr(), fb=Kh)
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            th1 = time.perf_counter()
            barrier()
            eh = rk.max_over_ranks(th1 - th0)
            host_input = {"value": round(S * Kh * world / eh, 2), "unit": "frames/s", "steps": Kh,
                          "ms_per_step": round(1e3 * eh / Kh, 4),
                          "pcie_gbs": round(2 * S * Kh * img / eh / 1e9, 2),
                          "how": "frames in pinned host memory (torch pin_memory), read by k_pyr_fused over PCIe; "
                                 "no CPU staging copy; same streams, same operating point, directly after the timed region"}
            frame_pos += Kh
            del hl, hr
        except Exception as e:   # noqa: BLE001
            host_input = {"error": repr(e)[:300]}

    # ---- roofline_solo: group 0 alone on the GPU, so each kernel of its chain runs by itself
    solo = {}
    Ks = max(0, min(args.solo_steps, FB))
    if Ks > 0 and host_input is not None and "error" not in host_input or Ks > 0 and Kh == 0:
        render_block(frame_pos)
        cs0 = pipes[0].counters()
        ctxs[0].timing(True)
        if sep_backend:
            ctxs[G].timing(True)
        run_all(0, Ks, False, groups=[0])
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        cs1 = pipes[0].counters()
        cnt_s = {k: cs1[k] - cs0[k] for k in cs1}
        for f in svs.FAMILIES:
            parts = [ctxs[0].timing_get(f)] + ([ctxs[G].timing_get(f)] if sep_backend else [])
            fms, fl = sum(p[0] for p in parts), sum(p[1] for p in parts)
            fb_ = algorithmic_bytes(f, cnt_s, fl)
            if fl and fms > 0:
                solo[f] = {"achieved": round(fb_ / (fms / 1e3) / 1e9, 2), "frac": round(fb_ / (fms / 1e3) / 1e9 / HBM_PEAK_GBS, 6),
                           "avg_launch_us": round(1e3 * fms / fl, 2), "launches": fl,
                           "algorithmic_bytes_per_launch": round(fb_ / fl, 1)}
        solo["_how"] = ("one group of %d streams runs %d steps alone after the timed region: its kernels are launched back "
                        "to back on one HIP stream, so every launch has the whole chip (no co-resident kernels of other "
                        "groups); same accounting as roofline_by_family" % (Sg, Ks))
    res = np.concatenate(res_g, axis=1)
    import ctypes as _C
    hostns = np.zeros(8)
    for c in ctxs:
        o = (_C.c_longlong * 8)()
        c.L.svslam_debug_host_ns(c.h, o)
        hostns += np.array(list(o), float)

    ok_frames = int((res["status"] != 3).sum())      # not LOST
    # full-size checks that need no oracle: (1) the duplicated stream, processed by another group /
    # context / batch position, reproduces stream 0 bit for bit; (2) trajectory error against the
    # renderer's ground truth on a sample of streams (timed region only, aligned at its first frame)
    replica_ok = bool(twin_of_0 == 0 or (np.array_equal(res["pose"][:, 0], res["pose"][:, twin_of_0]) and
                                         np.array_equal(res["n_inliers"][:, 0], res["n_inliers"][:, twin_of_0])))
    ate = []
    for s_ in range(0, S, max(1, S // 16))[:16]:
        gt = np.array([svs.synth_gt(seeds[s_], pre + Wm + f) for f in range(K)])
        ate.append(pl.ate_rmse(res["pose"][:, s_], gt))
    rank_rows = rank_table(rk, sdist, local_rank, pinned, seeds)
    total_frames = S * K * world
    value_first = total_frames / elapsed
    # `value` is the MEDIAN window: every window is exactly K steps between barrier + synchronize on both
    # sides, max over ranks; the first one carries the per-family HIP events and the result log, the others do not.  With an
    # even count the lower of the two middle windows is taken, so that ms_per_step belongs to a window that was really timed.
    windows = sorted([value_first] + list(spread))
    value = windows[(len(windows) - 1) // 2]
    elapsed_value = total_frames / value
    ranks_seen = int(round(rk.sum_over_ranks(1.0)))      # an all-reduce of ones over the job's communicator (1 without a process group)

    if rank == 0:
        dom = max(fam_t, key=lambda f: fam_t[f][0])
        ms, launches, _ = fam_t[dom]
        abytes = algorithmic_bytes(dom, cnt, launches)
        avg_s = (ms / 1e3) / max(launches, 1)
        achieved = (abytes / max(launches, 1)) / max(avg_s, 1e-12) / 1e9
        by_fam = {}
        for f, (fms, fl, _) in fam_t.items():
            fb = algorithmic_bytes(f, cnt, fl)
            if fl and fms > 0:
                by_fam[f] = {"achieved": round(fb / (fms / 1e3) / 1e9, 2), "frac": round(fb / (fms / 1e3) / 1e9 / HBM_PEAK_GBS, 6),
                             "avg_launch_us": round(1e3 * fms / fl, 2), "launches": fl}
        all_bytes = sum(algorithmic_bytes(f, cnt, fam_t[f][1]) for f in fam_t)
        out = {
            "metric": "stereo frames/sec (track + local BA), KITTI-00-shaped synthetic stereo 1241x376 "
                      "(620x188 after the reference's 1/2 decimation)",
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "ranks_seen": ranks_seen,
            "rank_exchange": "RCCL (torch.distributed nccl): barrier + max-reduction of the elapsed time, no data-path collective"
                             if rk.dist is not None and os.environ.get("SVS_DIST_BACKEND", "nccl") == "nccl" else
                             ("gloo (dry run)" if rk.dist is not None else "none (a single process outside torch.distributed.run)"),
            "ranks": rank_rows,
            "steps": K, "warmup": Wm,
            "ms_per_step": round(1e3 * elapsed_value / K, 4), "higher_is_better": True, "scaling": "weak",
            "value_windows": {"n": len(windows), "median": round(value, 2), "min": round(windows[0], 2), "max": round(windows[-1], 2),
                              "first": round(value_first, 2), "first_ms_per_step": round(1e3 * elapsed / K, 4),
                              "how": "value = median of the first timed window (the one the per-family HIP events, the roofline "
                                     "objects and the checks belong to) and the value_spread windows; each window is exactly "
                                     "--steps steps between barrier + synchronize, max over ranks"},
            "vs_baseline": None, "dtype": "u8/i32 fixed-point (pyramid, LK), f32 (GFTT), f64 (LM, BA)",
            "data": "synthetic",
            "config": {"workload": "configs[1..3] on synthetic input: full Frontend::AddFrame hot path on HIP "
                                   "(GFTT + pyramidal LK + triangulation + pose-only LM) with HIP local BA per "
                                   "keyframe (%s), config-00.yaml hyper-parameters (150 features, 10 active "
                                   "keyframes)" % ("completes before the next frame" if args.backend_mode == 1 else
                                                   "runs beside the next frames like the reference's backend thread, "
                                                   "lands %d frame(s) late, all of it inside the timed region" % max(1, args.backend_lag)),
                       **({"timed_blocks": "%d blocks of <= %d steps, each between barrier + synchronize; the next block's frames are "
                                           "rendered into the HBM ring in between, outside the timing" % (len(blocks), FB)} if len(blocks) > 1 else {}),
                       "map": "host (Frontend/Map/Backend bookkeeping on the CPU)" if cfg.device_map == 0 else
                              "device-resident (svslam_dmap_*: window, features, landmarks, observation counts in HBM; the host keeps ids and poses of the window)",
                       "preroll_steps": pre, "preroll_last_block_ba_keyframes_mean": round(preroll_kf, 2),
                       "streams_per_gpu": S, "host_threads_per_gpu": G, "bookkeeping_threads_per_group": args.host_threads, "frame": "%dx%d u8 stereo pair" % (W, H) + (" decimated on the fly from %dx%d frames in HBM" % (SW, SH) if args.full_res else ""),
                       "keyframes_in_timed_region": cnt["keyframes"],
                       "pose_only_xtol": float(os.environ.get("SVSLAM_PO_XTOL", "1e-12")),   # svslam_set_pose_only_xtol (0 = g2o's schedule to the last trial)
                       "ba_problem_mean": {"keyframes": round(cnt["ba_kf"] / max(cnt["ba_calls"], 1), 1),
                                           "landmarks": round(cnt["ba_lm"] / max(cnt["ba_calls"], 1), 1),
                                           "edges": round(cnt["ba_edges"] / max(cnt["ba_calls"], 1), 1),
                                           "lm_iterations": round(cnt["ba_iters"] / max(cnt["ba_calls"], 1), 2)},
                       "per_frame_mean": {"tracked_points": round(cnt["track_pts"] / max(cnt["frames"], 1), 1),
                                          "pose_edges": round(cnt["pose_edges"] / max(cnt["frames"], 1), 1)}, "tracked_ok_fraction": ok_frames / (S * K),
                       "checks": {"duplicate_stream_bit_identical": replica_ok,
                                  "ate_rmse_m_mean_of_%d_streams" % len(ate): round(float(np.mean(ate)), 4),
                                  "ate_rmse_m_max": round(float(np.max(ate)), 4)},
                       "parallelism": "%d independent streams/GPU x %d GPU(s), no collective" % (S, world)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "peak_measured": HBM_PEAK_MEASURED_GBS, "frac_of_peak_measured": round(achieved / HBM_PEAK_MEASURED_GBS, 6),
                         "traffic": measured_traffic(dom, cnt, launches),
                         "traffic_source": "committed PMC pass (profiles/pmc_traffic.json, tools/pmc_traffic.sh), not measured by this run",
                         "avg_launch_us": round(avg_s * 1e6, 2), "launches": launches,
                         "algorithmic_bytes_per_launch": round(abytes / max(launches, 1), 1)},
            # every kernel family priced the same way (launches of different groups overlap, so each
            # family's duration is its own HIP-event time, not a share of the wall clock), and the
            # whole step: all algorithmic bytes of the timed region over its wall time
            "roofline_by_family": by_fam,
            "roofline_compute": {f: compute_side(f, cnt, fam_t[f][0]) for f in ("local_ba", "lk") if fam_t[f][1]},
            "roofline_valu_step": valu_step(cnt, t_timed),
            "roofline_solo": solo,
            "value_spread": {"windows": [round(v, 1) for v in spread], "steps_each": K,
                             "min": round(min(spread), 1) if spread else None, "max": round(max(spread), 1) if spread else None,
                             "mean": round(float(np.mean(spread)), 1) if spread else None,
                             "rel_std": round(float(np.std(spread) / np.mean(spread)), 4) if spread else None,
                             "how": "further windows of the same length timed in the same process after the reported one"},
            "value_host_input": host_input,
            "roofline_whole_step": {"achieved": round(all_bytes * world / elapsed / 1e9, 2), "unit": "GB/s (all ranks)",
                                    "frac_of_peak_per_gpu": round(all_bytes / elapsed / 1e9 / HBM_PEAK_GBS, 6),
                                    "algorithmic_bytes_per_frame": round(all_bytes / max(cnt["frames"], 1), 1)},
            "kernel_ms": {f: round(fam_t[f][0], 3) for f in fam_t},
            # unit counts since process start (pre-roll and warm-up included): what a PMC pass over the whole
            # process divides its per-kernel totals by (tools/pmc_traffic.sh)
            "units_whole_process": {k: c1[k] for k in ("frames", "keyframes", "ba_calls", "gftt_calls", "pyr_left", "pyr_right",
                                                       "track_pts", "right_pts", "tri_pts", "pose_edges") if k in c1},
            "host_ms_per_step": {"in_step": round(cnt["ns_step"] / 1e6 / K / G, 3),
                                 "in_abi_calls": round(cnt["ns_kernel_calls"] / 1e6 / K / G, 3),
                                 "h2d_enqueue": round(hostns[0] / 1e6 / K / G, 3), "d2h_enqueue": round(hostns[1] / 1e6 / K / G, 3),
                                 "stream_wait": round(hostns[2] / 1e6 / K / G, 3), "event_collect": round(hostns[5] / 1e6 / K / G, 3), "ba_host_prep": round(hostns[4] / 1e6 / K / G, 3),
                                 "cpus_busy": round(cpu_busy, 2), "cpus_allowed": effective_cpus(),
                                 "pinned_to_gpu_numa_cpus": len(pinned),
                                 "minor_page_faults_per_step": round(minflt_timed / K, 1),
                                 "rss_growth_bytes_per_frame": round((rss1 - rss0) / max(S * K, 1), 1),
                                 "rss_gb": round(rss1 / 1e9, 2),
                                 "cpus_busy_by_thread_name": cpu_by_thread,
                             "capacity_events": {"corners_dropped": cnt.get("corners_dropped", 0), "ba_skipped": cnt.get("ba_skipped", 0)}},
        }
        if world == 1 and not args.no_cpu_baseline:
            base = cpu_baseline(svs, pl, ctx, cfg, seeds, max(pre, 0), args.cpu_frames, local_rank, cam_r, (SW, SH),
                                effective_cpus())
            out["cpu_baseline"] = base["one_thread"]
            out["cpu_baseline_all_cores"] = base["all_cores"]
    run_full_res = rank == 0 and world == 1 and args.full_res_streams > 0 and not args.full_res
    if run_full_res:
        ctx.dev_free(d_left); ctx.dev_free(d_right)     # (before the contexts go: the leg below needs the memory)
    for p in pipes:
        p.close()
    if rank == 0:
        # ---- value_full_res: BASELINE.json's metric names 1241x376 frames; the reported run keeps the frames in HBM already
        #      decimated (the reference's working resolution).  This leg stores them at full size and fuses the decimation
        #      (src/dataset.cpp:126-129) into the pyramid's level 0: the same hot path plus the 4x larger frame read.
        if run_full_res:
            import subprocess
            # (at most 20 + 5 steps: a full-size frame ring of 8192 streams then fits the GPU's memory)
            cmd = [sys.executable, os.path.abspath(__file__), "--full-res", "--streams", str(args.full_res_streams), "--steps", str(min(K, 20)),
                   "--warmup", str(min(Wm, 5)), "--no-cpu-baseline", "--spread-windows", "0", "--host-input-steps", "0", "--solo-steps", "0",
                   "--full-res-streams", "0"]
            # the child is a single process of its own: it must not join the parent's rendezvous
            env = {k: v for k, v in os.environ.items()
                   if not (k.startswith("TORCHELASTIC_") or k.startswith("MASTER_") or k.startswith("TORCH_NCCL") or
                           k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_NAME",
                                 "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE", "NCCL_ASYNC_ERROR_HANDLING", "OMP_NUM_THREADS"))}
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=env)
                if r.returncode != 0 or not r.stdout.strip():
                    raise RuntimeError("child rc %d: %s" % (r.returncode, (r.stderr or "").strip()[-200:]))
                d = json.loads(r.stdout.strip().splitlines()[-1])
                out["value_full_res"] = {"value": d["value"], "unit": "frames/s", "streams": d["config"]["streams_per_gpu"], "steps": d["steps"],
                                         "ms_per_step": d["ms_per_step"], "frame": d["config"]["frame"],
                                         "checks": d["config"]["checks"],
                                         "how": "the same script run once more after the reported measurement: frames kept in HBM at "
                                                "1241x376, 1/2 decimation fused into the pyramid kernel (SURVEY 8 row f3); fewer streams "
                                                "and at most 20 + 5 steps because a frame is 4x the bytes"}
            except Exception as e:   # noqa: BLE001
                out["value_full_res"] = {"error": repr(e)[:300]}
        print(json.dumps(out), flush=True)
    rk.close()


def _vp(v):
    import ctypes
    return ctypes.c_void_p(int(v))


def rank_table(rk, sdist, local_rank, pinned, seeds):
    """one row per rank — device, its NUMA node, CPUs the rank may use, CPUs it pinned itself to, its stream seeds — gathered
    with one all-reduce (every rank fills its own slots of a zero vector), so that an N-GPU run is diagnosable from its line"""
    w = rk.world
    v = np.zeros(w * 6)
    o = 6 * rk.rank
    v[o:o + 6] = [local_rank, sdist.device_numa_node(local_rank), effective_cpus(), len(pinned), seeds[0] if seeds else -1, len(seeds)]
    v = rk.allreduce(v)
    return [{"rank": r, "device": int(v[6 * r]), "numa_node": int(v[6 * r + 1]), "cpus_allowed": int(v[6 * r + 2]),
             "cpus_pinned": int(v[6 * r + 3]), "first_stream_seed": int(v[6 * r + 4]), "streams": int(v[6 * r + 5])} for r in range(w)]


def dry_run(args):
    """bench.py --dry-run: everything of the launch contract that needs no GPU.  One process per rank
    under torch.distributed.run (gloo unless SVS_DIST_BACKEND says otherwise), rank r owns streams [rS, (r+1)S), W warm-up
    steps, then exactly K steps between barrier on both sides, max over ranks, rank 0 prints the one JSON line.  The step is
    a sleep of (1 + rank % 3) ms: the slowest rank must set the reported time."""
    sdist = importlib.import_module("stereovision-slam_amd.dist")
    rk = sdist.init(os.environ.get("SVS_DIST_BACKEND", "gloo"))
    S, K, Wm = (args.streams if args.streams > 0 else 16), args.steps, args.warmup
    seeds = list(rk.stream_seeds(S))
    step_s = 1e-3 * (1 + rk.rank % 3)
    for _ in range(Wm):
        time.sleep(step_s)
    rk.barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        time.sleep(step_s)
    t1 = time.perf_counter()
    rk.barrier()
    elapsed = rk.max_over_ranks(t1 - t0)
    ranks_seen = int(round(rk.sum_over_ranks(1.0)))
    rows = rank_table(rk, sdist, rk.local_rank, set(), seeds)
    if rk.rank == 0:
        print(json.dumps({
            "metric": "stereo frames/sec (track + local BA), KITTI-00-shaped synthetic stereo 1241x376 "
                      "(620x188 after the reference's 1/2 decimation)",
            "dry_run": True, "value": round(S * K * rk.world / elapsed, 2), "unit": "frames/s", "n_gpus": rk.world, "ranks_seen": ranks_seen,
            "rank_exchange": "gloo (dry run)" if rk.dist is not None else "none (a single process outside torch.distributed.run)",
            "ranks": rows, "steps": K, "warmup": Wm, "ms_per_step": round(1e3 * elapsed / K, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "none (dry run: no kernel ran)", "data": "none (dry run)",
            "config": {"workload": "dry run of the launch contract: a rank-dependent sleep per step, no GPU work",
                       "streams_per_gpu": S, "parallelism": "%d independent streams/GPU x %d rank(s), no collective" % (S, rk.world)}}),
              flush=True)
    rk.close()


def cpu_baseline(svs, pl, ctx, cfg, seeds, preroll, timed_frames, device, cam_r, src, cores):
    """The CPU twin (reference-shaped host logic over the oracle kernels: pyramids rebuilt per LK call,
    numeric BA Jacobians like g2o) on a bounded sample of the same workload, at the same operating
    point: every thread runs ONE stream through the same pre-roll (untimed) and then `timed_frames`
    frames (timed).  Two legs: 1 thread, and one stream per usable core.  kind = "port": the real
    OpenCV + g2o binary cannot be built on either box."""
    import threading
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pipe_cpu
    sw, sh = src
    cfg1 = pl.default_config(W, H, backend_on=cfg.backend_on, src_width=cfg.src_width, src_height=cfg.src_height)
    F = preroll + timed_frames
    nsrc = max(1, min(4, cores, len(seeds)))           # distinct streams, shared round-robin by the threads
    img = sw * sh
    dl = ctx.dev_alloc(nsrc * F * img); dr = ctx.dev_alloc(nsrc * F * img)
    svs.synth_render_streams_device(seeds[:nsrc], 0, F, sw, sh, dl, dr, device=device, cam=cam_r, block=1024)
    left = np.zeros((nsrc, F, sh, sw), np.uint8); right = np.zeros((nsrc, F, sh, sw), np.uint8)
    ctx.dev_download(dl, left); ctx.dev_download(dr, right)
    ctx.dev_free(dl); ctx.dev_free(dr)

    def leg(nthreads):
        bar = threading.Barrier(nthreads + 1)
        errs = []

        def work(t):
            try:
                twin = pipe_cpu.make(cfg1, nstreams=1)
                L, R = left[t % nsrc], right[t % nsrc]
                for f in range(preroll):
                    twin.step([L[f]], [R[f]])
                bar.wait()
                for f in range(preroll, F):
                    twin.step([L[f]], [R[f]])
                twin.flush()
                bar.wait()
                twin.close()
            except Exception as e:   # noqa: BLE001
                errs.append(e)
                bar.abort()
        th = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
        for t_ in th:
            t_.start()
        bar.wait()
        t0 = time.perf_counter()
        bar.wait()
        dt = time.perf_counter() - t0
        for t_ in th:
            t_.join()
        if errs:
            raise errs[0]
        return nthreads * timed_frames / dt

    what = ("%%d stream(s) x %d frames after a %d-frame untimed pre-roll (same synthetic workload and operating "
            "point as the GPU run), CPU restatement (oracle kernels: pyramids rebuilt per LK call, numeric BA "
            "Jacobians like g2o), %%d thread(s), %s" % (timed_frames, preroll, cpu_model()))
    v1 = leg(1)
    vn = leg(cores) if cores > 1 else v1
    return {"one_thread": {"value": round(v1, 2), "unit": "frames/s", "cores": 1, "kind": "port", "sample": what % (1, 1)},
            "all_cores": {"value": round(vn, 2), "unit": "frames/s", "cores": cores, "kind": "port",
                          "sample": what % (cores, cores) + "; one stream per thread"}}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip() + " (%d logical cores)" % os.cpu_count()
    except OSError:
        pass
    return "unknown CPU"


if __name__ == "__main__":
    main()
