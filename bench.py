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
FW, FH = 1241, 376                  # the camera's frame (BASELINE.json's metric names this size): stored in HBM by default
HBM_PEAK_GBS = 8000.0               # spec (MI355X_MICROARCH.md)
HBM_PEAK_MEASURED_GBS = 6290.0      # float4 copy on MI355X (MI355X_MICROARCH.md chip table); SURVEY 8d's denominator
LEVEL_PIX = (620 * 188, 310 * 94, 155 * 47, 78 * 24)


SRC_PIX = 0     # pixels of a stored frame when the 1/2 decimation is fused into the pyramid (0: frames arrive decimated)


def algorithmic_bytes(fam, cnt, launches):
    """SURVEY.md §8d per-unit figures x the units the timed launches processed."""
    if fam == "lk":          # 4 levels x (14x14 I patch + 20x20 J region) + 21 B point I/O
        return (cnt["track_pts"] + cnt["right_pts"]) * (2384 + 21)
    if fam == "pose_only":   # 40 B per edge (xyz f64 + uv f32 + flags) + 56 B pose
        return cnt["pose_edges"] * 40 + launches * 56
    if fam == "pyramid":     # image read once + levels 1..3 written once; from a full-resolution frame: the
        #                      rows of the 1241x376 frame that the 1/2 nearest decimation samples (every second one, 233 308 B)
        #                      read once + all four levels written once (level 0 IS the decimation's output)
        return (cnt["pyr_left"] + cnt["pyr_right"]) * (sum(LEVEL_PIX) + SRC_PIX)
    if fam == "gftt":        # image read once + rect list + corners out
        return cnt["gftt_calls"] * LEVEL_PIX[0] + cnt["gftt_rects"] * 8 + cnt["corners"] * 8
    if fam == "triangulate":
        return cnt["tri_pts"] * (16 + 24 + 1)
    if fam in ("local_ba", "ba_solve"):    # iters x (E x 40 B obs+ids + K x 56 B + M x 24 B); ba_solve: the solver kernel of the family alone
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
    units = {"local_ba": cnt["ba_calls"], "ba_solve": cnt["ba_calls"], "lk": cnt["track_pts"] + cnt["right_pts"], "pose_only": cnt["frames"],
             "gftt": cnt["gftt_calls"], "pyramid": cnt["pyr_left"] + cnt["pyr_right"], "triangulate": cnt["tri_pts"]}.get(fam)
    if not units or not launches:
        return None
    return t["bytes"] * units / launches


def pmc_stamp(name, build_info):
    """which build a committed PMC summary under profiles/ was measured on (its "build_info" field, written by tools/pmc_*.sh
    from svslam_build_info()) and whether that is the library this run loaded"""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
    except (OSError, ValueError):
        return {"file": "profiles/" + name, "build_info": None, "matches_loaded_library": False}
    b = d.get("build_info")
    return {"file": "profiles/" + name, "build_info": b, "operating_point": d.get("operating_point"),
            "matches_loaded_library": bool(b) and b == build_info}


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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("SVS_BENCH_STREAMS", "0")),
                    help="independent stereo streams per GPU, advanced in lockstep (0 = the largest of 12288 / 8192 / "
                         "6144 whose frame ring of warmup + steps frames fits the GPU's memory; when none does — full-size "
                         "frames and many steps — 8192 streams with a shorter ring, the timed region then runs in blocks)")
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
    ap.add_argument("--pre-decimated", action="store_true",
                    help="keep the frames in HBM already decimated to 620x188 (the reference's working resolution, what "
                         "Frontend::AddFrame sees; rounds 1-5 reported this as `value`).  Default since round 6: frames stored at "
                         "the camera's 1241x376 — the size BASELINE.json's metric names — and the reference's 1/2 decimation "
                         "(Dataset::NextFrame, src/dataset.cpp:126-129) fused into the pyramid's level 0 (SURVEY 8 row f3)")
    ap.add_argument("--full-res", action="store_true", help="(the default since round 6; accepted for older scripts)")
    ap.add_argument("--predecimated-streams", "--full-res-streams", dest="secondary_streams", type=int, default=-1,
                    help="streams of the value_predecimated leg: after the reported run the script runs itself once more with "
                         "--pre-decimated (0 = skip; -1 = 12288 when --streams is left to the script, i.e. the headline "
                         "operating point, else skip; one GPU only, never under torchrun's N > 1)")
    ap.add_argument("--super-windows", type=int, default=6,
                    help="when a window of --steps steps lasts < 1 s: one more timed region of this many x --steps steps back to "
                         "back (value_super_window; 0 = skip)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: the launch contract only — rank discovery, stream partition, barriers, max-over-ranks timing, "
                         "the JSON line (dry_run: true) — with a rank-dependent sleep as the step; what tests/test_abi_and_host.py "
                         "runs at world 8 over gloo on a CPU box")
    args = ap.parse_args()
    if args.dry_run:
        return dry_run(args)
    full_res = not args.pre_decimated
    if args.secondary_streams < 0:        # (a small --streams run of a test or a latency script must not spawn a 12288-stream child)
        args.secondary_streams = 12288 if (args.streams <= 0 and full_res) else 0

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
    # The synthetic frames live in HBM.  One ring of FB frames per stream is rendered block by block (pre-roll blocks first,
    # then the block that holds the warm-up and the first timed steps), always outside the timed region.  FB = warmup + steps
    # when that fits (one timed block), else as many frames as fit: the timed region then runs in blocks (see timed()).
    SW, SH = (FW, FH) if full_res else (W, H)             # stored frame size
    global SRC_PIX
    # INTER_NEAREST at 1/2 (src/dataset.cpp:126-129) samples the even pixels of the even rows: the rows that hold samples are
    # read (whole rows: the samples sit in every 64-byte line of them), the odd rows are never touched
    SRC_ROWS = (SH + 1) // 2
    SRC_PIX = SW * SRC_ROWS if full_res else 0
    budget = 225e9                                   # of the MI355X's 288 GB; the pyramids, maps and work buffers need ~1 MB per stream
    if torch.cuda.is_available():
        budget = min(budget, 0.8 * torch.cuda.mem_get_info(local_rank)[0])
    per_stream = 1 << 20

    def cap_for(fb):
        return int(budget // (2 * SW * SH * fb + per_stream))

    def ring_for(streams):
        return int((budget / max(streams, 1) - per_stream) // (2 * SW * SH))
    FB = Wm + K if args.ring_frames <= 0 else max(Wm + 1, min(Wm + K, args.ring_frames))
    if S <= 0:
        # more streams per launch fill the chip better (measured: 6144 / 8192 / 12288 streams = 1.00 / 1.04 / 1.08),
        # the frame ring decides what fits
        S = next((c for c in (12288, 8192, 6144) if c <= cap_for(FB)), 0)
        if S == 0:
            S = 8192
            FB = max(Wm + 1, min(FB, ring_for(S)))
        # a rank that may use only a few cores (N ranks sharing one CPU quota) cannot feed that many streams:
        # ~27 us of host CPU per frame; keep its memory footprint in proportion
        # (only with the map on the host: the device-resident map costs the host < 1 core per 12 288 streams)
        lw = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        if args.host_map:
            S = min(S, 1024 * max(1, effective_cpus() // max(1, lw)))
    if S > cap_for(FB):
        if args.ring_frames <= 0 and ring_for(S) >= Wm + 4:
            FB = min(FB, ring_for(S))                # an explicit stream count keeps its streams and gets a shorter ring
        else:
            cap = cap_for(FB)
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
                            src_width=SW if full_res else 0, src_height=SH if full_res else 0,
                            low_latency=1 if args.low_latency else 0,
                            device_map=0 if args.host_map else 1, backend_lag=max(1, args.backend_lag))
    pipes = [pl.Pipeline(cfg, nstreams=Sg, device=local_rank) for _ in range(G)]
    ctxs = [svs.Context.borrow(p.kernel_ctx(), W, H) for p in pipes]   # alloc / timing through the pipelines' contexts
    ctx = ctxs[0]
    sep_backend = args.backend_mode == 2 and not dev_map      # host map: the backend has its own context (second HIP stream) per pipeline
    if sep_backend:
        ctxs = ctxs + [svs.Context.borrow(p.backend_ctx(), W, H) for p in pipes]

    img = SW * SH
    cam_r = tuple(2 * v for v in svs.KITTI00_HALF_CAM) if full_res else svs.KITTI00_HALF_CAM
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

    # ---- the frame ring: frames [ring["base"], ring["base"] + FB) of every stream are in HBM; ring["next"] is the frame the
    #      next step of every stream reads.  advance() renders the ring again from that frame on (never inside a timed region).
    ring = {"base": None, "next": 0}

    def advance():
        render_block(ring["next"])
        ring["base"] = ring["next"]

    def ring_left():
        return 0 if ring["base"] is None else ring["base"] + FB - ring["next"]

    import resource

    # shader clock under load: one wave on a HIP stream of its own spins through (most of) a window and reports the engine
    # clock it saw (clock64 against the constant 100 MHz counter) — what a window's kernels ran at
    probe = None
    if torch.cuda.is_available() and os.environ.get("SVS_BENCH_CLOCK_PROBE", "1") == "1":
        try:
            probe = svs.Context(W, H, max_slots=1, max_jobs=1, max_pts=64, max_corners=16, max_kf=0, max_lm=0, max_obs=0, device=local_rank)
        except Exception:   # noqa: BLE001
            probe = None

    def with_clock(fn, expect_s):
        """runs fn() with the clock probe spinning beside it for ~80 % of the expected duration; returns (fn(), MHz or None)"""
        if probe is None or expect_s <= 0:
            return fn(), None
        box = {}

        def spin():
            try:
                box["mhz"] = probe.clock_mhz(1, max(1.0, min(2000.0, 800.0 * expect_s)))
            except Exception:   # noqa: BLE001
                box["mhz"] = None
        th_ = threading.Thread(target=spin)
        th_.start()
        out_ = fn()
        th_.join()
        return out_, (round(box["mhz"], 1) if box.get("mhz") else None)

    def timed(nsteps, want=False, bufs=None, account=None):
        """`nsteps` steps of every stream, timed.  One block when the ring holds them (re-rendered first if it only holds a part
        and could hold all), else block by block: every block between barrier + synchronize on both sides, the next block's
        frames rendered in between, outside the timing.  Returns the summed wall time of the blocks (this rank)."""
        left, done, t_sum, nblocks = nsteps, 0, 0.0, 0
        if ring_left() < left and (left <= FB or ring_left() <= 0):
            advance()
        while left > 0:
            if ring_left() <= 0:
                advance()
            n = min(ring_left(), left)
            first = ring["next"] - ring["base"]
            barrier()
            if account is not None:
                ru0 = resource.getrusage(resource.RUSAGE_SELF)
                tc0 = thread_cpu_seconds()
            t0 = time.perf_counter()
            run_all(first, n, want, out_bufs=None if bufs is None else [b_[done:done + n] for b_ in bufs])
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            t1 = time.perf_counter()
            if account is not None:
                ru1 = resource.getrusage(resource.RUSAGE_SELF)
                tc1 = thread_cpu_seconds()
                account["cpu"] += (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
                account["minflt"] += ru1.ru_minflt - ru0.ru_minflt
                for k in tc1:
                    account["threads"][k] = account["threads"].get(k, 0.0) + tc1[k] - tc0.get(k, 0.0)
            barrier()
            t_sum += t1 - t0
            ring["next"] += n
            left -= n
            done += n
            nblocks += 1
        return t_sum, nblocks

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
            advance()
            run_all(0, n, False)
            ring["next"] += n
            pre += n
            ca = counters_sum()
            calls = ca["ba_calls"] - cb["ba_calls"]
            preroll_kf = (ca["ba_kf"] - cb["ba_kf"]) / max(calls, 1)
            if args.preroll < 0 and (preroll_kf >= window_full - 0.05 or pre >= 400):
                break
    # ---- warmup, untimed (in the ring block the first timed steps come from)
    advance()
    run_all(0, min(Wm, FB - 1), False)
    ring["next"] += min(Wm, FB - 1)
    for _ in range(Wm - min(Wm, FB - 1)):          # (a ring shorter than the warm-up: the rest one frame at a time)
        advance()
        run_all(0, 1, False)
        ring["next"] += 1
    c0 = counters_sum()
    for c in ctxs:
        c.timing(True)
    # the per-frame results of the timed region (88 B per frame: the bench's own log) are allocated and touched here, so
    # that rss_growth_bytes_per_frame below is the growth of the pipeline / library, not of this script's arrays
    res_bufs = [np.zeros((K, Sg), pl.RESULT_DTYPE) for _ in range(G)]
    for b_ in res_bufs:
        b_.view(np.uint8).fill(0)
    rss0 = int(open("/proc/self/statm").read().split()[1]) * os.sysconf("SC_PAGE_SIZE")
    first_timed_frame = ring["next"]
    acct = {"cpu": 0.0, "minflt": 0, "threads": {}}
    # ---- the K timed steps (the window the per-family HIP events, the result log and the checks belong to)
    t_timed, nblocks = timed(K, True, res_bufs, acct)
    clocks = []
    blocks_hint = [0] * nblocks      # (the clock probe runs beside single-block windows only: a block boundary renders frames)
    cpu_timed, minflt_timed, tc_acc = acct["cpu"], acct["minflt"], acct["threads"]
    res_g = res_bufs
    rss1 = int(open("/proc/self/statm").read().split()[1]) * os.sysconf("SC_PAGE_SIZE")
    cpu_by_thread = {k: round(v / max(t_timed, 1e-9), 2) for k, v in tc_acc.items() if v > 0.005 * t_timed}
    cpu_busy = cpu_timed / max(t_timed, 1e-9)
    elapsed = rk.max_over_ranks(t_timed)
    c1 = counters_sum()
    cnt = {k: c1[k] - c0[k] for k in c1}
    fam_t = {}
    for f in list(svs.FAMILIES) + list(svs.KERNEL_FAMILIES):
        parts = [c.timing_get(f) for c in ctxs]
        fam_t[f] = (sum(p[0] for p in parts), sum(p[1] for p in parts), sum(p[2] for p in parts))
    for c in ctxs:
        c.timing(False)

    # ---- value_spread: further windows of K steps, same process, same operating point (the 0.6-s
    #      window of the driver's 20 steps scatters by +-10 % from run to run; here is the scatter inside one run).
    #      Every window also logs how many keyframes (= local-BA problems) fell into it: the windows differ in their work.
    spread, spread_kf = [], []
    for _ in range(max(0, args.spread_windows)):
        ck0 = counters_sum()["keyframes"]
        (ts, _nb), mhz_ = with_clock(lambda: timed(K), t_timed if len(blocks_hint) == 1 else 0)
        spread.append(S * K * world / rk.max_over_ranks(ts))
        spread_kf.append(counters_sum()["keyframes"] - ck0)
        clocks.append(mhz_)

    # ---- value_super_window: when a window is shorter than a second, super_windows x K steps back to back
    #      as ONE measurement (with a full-size frame ring it still runs in blocks of at most FB steps, each between barrier +
    #      synchronize; the time is the sum of the blocks)
    super_window = None
    if args.super_windows > 0 and elapsed < 1.0:
        ck0 = counters_sum()["keyframes"]
        Ksw = args.super_windows * K
        ts, nb_sw = timed(Ksw)
        e_sw = rk.max_over_ranks(ts)
        super_window = {"value": round(S * Ksw * world / e_sw, 2), "unit": "frames/s", "steps": Ksw, "seconds": round(e_sw, 4),
                        "ms_per_step": round(1e3 * e_sw / Ksw, 4), "timed_blocks": nb_sw,
                        "keyframes_per_step": round((counters_sum()["keyframes"] - ck0) / Ksw, 1),
                        "how": "%d x --steps steps as one measurement directly after the value_spread windows, no HIP events; "
                               "blocks of at most %d steps (the frame ring), each between barrier + synchronize, time = sum of the blocks" % (args.super_windows, FB)}

    # ---- value_host_input: the same step with the frames in pinned host memory (what a host-buffer boundary
    #      hands over): k_pyr_fused reads them across PCIe itself, nothing is staged or copied by the CPU
    host_input = None
    # (at most ~14 GB of pinned host memory per eye)
    Kh = max(0, min(args.host_input_steps, FB, int(14e9 // max(S * img, 1))))
    if world > 1:
        Kh = 0          # a one-GPU figure (PCIe of one device; 2 x 14 GB of pinned host memory per rank): not taken at N > 1
    if Kh > 0:
        try:
            svs.synth_render_streams_device(seeds, ring["next"], Kh, SW, SH, d_left, d_right, device=local_rank, cam=cam_r)
            ring["base"] = None                      # (the ring's layout is gone: the next leg renders it again)
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
                          "pcie_gbs": round(2 * S * Kh * (SRC_PIX if full_res else img) / eh / 1e9, 2),
                          "pcie_bytes_per_frame": 2 * (SRC_PIX if full_res else img),
                          "how": "frames in pinned host memory (torch pin_memory), read by k_pyr_fused over PCIe; "
                                 "no CPU staging copy; same streams, same operating point, directly after the timed region"}
            ring["next"] += Kh
            del hl, hr
        except Exception as e:   # noqa: BLE001
            host_input = {"error": repr(e)[:300]}

    # ---- roofline_solo: group 0 alone on the GPU, so each kernel of its chain runs by itself
    solo = {}
    Ks = max(0, min(args.solo_steps, FB))
    if Ks > 0 and (host_input is None or "error" not in host_input):
        advance()
        cs0 = pipes[0].counters()
        ctxs[0].timing(True)
        if sep_backend:
            ctxs[G].timing(True)
        run_all(0, Ks, False, groups=[0])
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        cs1 = pipes[0].counters()
        cnt_s = {k: cs1[k] - cs0[k] for k in cs1}
        for f in list(svs.FAMILIES) + list(svs.KERNEL_FAMILIES):
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
        gt = np.array([svs.synth_gt(seeds[s_], first_timed_frame + f) for f in range(K)])
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
        # The dominant KERNEL: the interval with the largest summed HIP-event time among those that enclose one kernel (ba_solve =
        # k_local_ba_t alone; pyramid, lk, pose_only, triangulate are one kernel each; gftt is two short ones and counted as one).
        # "local_ba" (gather + build + solver + scatter) is a family of four kernels: priced in roofline_by_family, not here.
        single = [f for f in fam_t if f != "local_ba"]
        dom = max(single, key=lambda f: fam_t[f][0])
        ms, launches, _ = fam_t[dom]
        abytes = algorithmic_bytes(dom, cnt, launches)
        avg_s = (ms / 1e3) / max(launches, 1)
        achieved = (abytes / max(launches, 1)) / max(avg_s, 1e-12) / 1e9
        dom_traffic_fam = dom if measured_traffic(dom, cnt, launches) is not None else ("local_ba" if dom == "ba_solve" else dom)
        by_fam = {}
        for f, (fms, fl, _) in fam_t.items():
            fb = algorithmic_bytes(f, cnt, fl)
            if fl and fms > 0:
                by_fam[f] = {"achieved": round(fb / (fms / 1e3) / 1e9, 2), "frac": round(fb / (fms / 1e3) / 1e9 / HBM_PEAK_GBS, 6),
                             "avg_launch_us": round(1e3 * fms / fl, 2), "launches": fl, "kernels": svs.FAMILY_KERNELS.get(f),
                             "algorithmic_bytes_per_launch": round(fb / fl, 1)}
        all_bytes = sum(algorithmic_bytes(f, cnt, fam_t[f][1]) for f in svs.FAMILIES)
        build_info = svs.load().svslam_build_info().decode()
        ctx_xtol = ctx.pose_only_xtol_effective()
        out = {
            "metric": "stereo frames/sec (track + local BA), KITTI-00-shaped synthetic stereo 1241x376 "
                      "(620x188 after the reference's 1/2 decimation)",
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "ranks_seen": ranks_seen, "library": build_info,
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
                       "frame_ring": "%d frames per stream in HBM (%.1f GB)" % (FB, 2.0 * S * FB * img / 1e9),
                       **({"timed_blocks": "%d blocks of <= %d steps, each between barrier + synchronize; the next block's frames are "
                                           "rendered into the HBM ring in between, outside the timing" % (nblocks, FB)} if nblocks > 1 else {}),
                       "map": "host (Frontend/Map/Backend bookkeeping on the CPU)" if cfg.device_map == 0 else
                              "device-resident (svslam_dmap_*: window, features, landmarks, observation counts in HBM; the host keeps ids and poses of the window)",
                       "preroll_steps": pre, "preroll_last_block_ba_keyframes_mean": round(preroll_kf, 2),
                       "streams_per_gpu": S, "host_threads_per_gpu": G, "bookkeeping_threads_per_group": args.host_threads, "frame": ("%dx%d u8 stereo pair in HBM, the reference's 1/2 decimation to %dx%d fused into the pyramid kernel" % (SW, SH, W, H)) if full_res else
                                ("%dx%d u8 stereo pair in HBM (already decimated: the reference's working resolution)" % (W, H)),
                       "keyframes_in_timed_region": cnt["keyframes"],
                       "pose_only_xtol": ctx_xtol,   # read back from the context (svslam_get_pose_only_xtol; 0 = g2o's schedule to the last trial)
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
            "roofline": {"bound": "hbm", "kernel": "+".join(svs.FAMILY_KERNELS.get(dom, [dom])), "interval": dom,
                         "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "peak_measured": HBM_PEAK_MEASURED_GBS, "frac_of_peak_measured": round(achieved / HBM_PEAK_MEASURED_GBS, 6),
                         "traffic": measured_traffic(dom_traffic_fam, cnt, launches),
                         "traffic_source": "committed PMC passes (profiles/pmc_traffic.json, tools/pmc_traffic.sh: FETCH_SIZE and WRITE_SIZE in "
                                           "separate runs over the launches of a timed window, FETCH_SIZE x 2 on gfx950), not measured by this run"
                                           + ("; the figure is the whole local_ba family's (gather + build + solver + scatter)" if dom_traffic_fam != dom else ""),
                         "traffic_stamp": pmc_stamp("pmc_traffic.json", build_info),
                         "avg_launch_us": round(avg_s * 1e6, 2), "launches": launches,
                         "algorithmic_bytes_per_launch": round(abytes / max(launches, 1), 1),
                         "how": "HIP events on the launching stream around this kernel only, summed over the first timed window's launches "
                                "(kernels of the other host threads' streams run beside it); the rocprofv3 --kernel-trace --stats row of the "
                                "same command is profiles/r6_bench_default_kernel_stats.csv"},
            # every kernel family priced the same way (launches of different groups overlap, so each
            # family's duration is its own HIP-event time, not a share of the wall clock), and the
            # whole step: all algorithmic bytes of the timed region over its wall time
            "roofline_by_family": by_fam,
            "roofline_compute": {f: compute_side(f, cnt, fam_t[f][0]) for f in ("local_ba", "lk") if fam_t[f][1]},
            "roofline_valu_step": valu_step(cnt, t_timed),
            "roofline_solo": solo,
            "pmc_stamps": {f: pmc_stamp(f, build_info) for f in ("pmc_traffic.json", "pmc_valu.json", "pmc_valu_step.json")},
            "value_super_window": super_window,
            "value_spread": {"windows": [round(v, 1) for v in spread], "keyframes": spread_kf, "keyframes_first_window": cnt["keyframes"], "steps_each": K,
                             "shader_clock_mhz": clocks,
                             "hip_events": "only the first window records the per-family HIP events (pyramid, lk, gftt, triangulate, pose_only, local_ba, "
                                           "ba_solve) and writes the per-frame result log; the spread windows and the super-window record none",
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
            # the same counts for the first timed window alone, with the launches of every family in it: a PMC pass over a run whose
            # LAST launches are this window (no further legs) prices the steady state, not the pre-roll's growing windows (tools/pmc_reduce.py)
            "units_timed_window": {**{k: cnt[k] for k in ("frames", "keyframes", "ba_calls", "gftt_calls", "pyr_left", "pyr_right",
                                                          "track_pts", "right_pts", "tri_pts", "pose_edges") if k in cnt},
                                   "launches": {f: fam_t[f][1] for f in fam_t}},
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
    run_secondary = rank == 0 and world == 1 and args.secondary_streams > 0 and full_res
    if run_secondary:
        ctx.dev_free(d_left); ctx.dev_free(d_right)     # (before the contexts go: the leg below needs the memory)
    for p in pipes:
        p.close()
    if probe is not None:
        probe.close()
    if rank == 0:
        # ---- value_predecimated: rounds 1-5 reported the run whose frames are stored in HBM already decimated (620x188: what the
        #      reference's Frontend::AddFrame sees) as `value`; since round 6 `value` is the 1241x376 configuration
        #      BASELINE.json's metric names and the pre-decimated run is this named secondary (more streams fit: 12288).
        if run_secondary:
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), "--pre-decimated", "--streams", str(args.secondary_streams), "--steps", str(K),
                   "--warmup", str(Wm), "--no-cpu-baseline", "--spread-windows", "2", "--super-windows", "0", "--host-input-steps", "0", "--solo-steps", "0",
                   "--predecimated-streams", "0"]
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
                out["value_predecimated"] = {"value": d["value"], "unit": "frames/s", "streams": d["config"]["streams_per_gpu"], "steps": d["steps"],
                                             "ms_per_step": d["ms_per_step"], "frame": d["config"]["frame"], "value_windows": d["value_windows"],
                                             "roofline": d["roofline"], "kernel_ms": d["kernel_ms"],
                                             "checks": d["config"]["checks"],
                                             "how": "the same script run once more after the reported measurement with --pre-decimated: frames "
                                                    "kept in HBM at 620x188 (a quarter of the frame bytes, so 12288 streams fit); what rounds "
                                                    "1-5 reported as `value`"}
            except Exception as e:   # noqa: BLE001
                out["value_predecimated"] = {"error": repr(e)[:300]}
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
    # the host layout of a real rank: G group threads, each waiting for "its GPU work" the way the library waits for a HIP
    # stream (sleep-polling, csrc/svslam_hip.hip wait_stream), + this thread.  What the dry run checks about it:
    # N ranks x (G + 1) threads on one node do not oversubscribe the CPUs a rank may use — the threads sleep, they do not spin.
    import resource
    import threading
    G = args.groups if args.groups > 0 else 4
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(rk.world)))
    cpus_per_rank = effective_cpus() / max(1, local_world)

    def steps(n):
        def group():
            for _ in range(n):
                t_end = time.perf_counter() + step_s
                while time.perf_counter() < t_end:          # poll every ~150 us like wait_stream's nap
                    time.sleep(min(150e-6, max(0.0, t_end - time.perf_counter())))
        th = [threading.Thread(target=group) for _ in range(G)]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
    steps(Wm)
    rk.barrier()
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    steps(K)
    t1 = time.perf_counter()
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    rk.barrier()
    elapsed = rk.max_over_ranks(t1 - t0)
    busy = ((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)) / max(t1 - t0, 1e-9)
    busy_max = rk.max_over_ranks(busy)
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
            "host": {"threads_per_rank": G + 1, "ranks_on_node": local_world, "cpus_allowed": effective_cpus(),
                     "cpus_per_rank": round(cpus_per_rank, 2), "cpu_cores_busy_per_rank_max": round(busy_max, 3),
                     # measured on the GPU box at the headline operating point (bench.py host_ms_per_step.cpus_busy, profiles/r6_bench_default.json)
                     "product_cpu_cores_busy_per_rank": 0.6,
                     "oversubscribed": bool(busy_max > cpus_per_rank or 0.6 > cpus_per_rank),
                     "how": "every rank runs %d group threads that wait the way the library waits for a HIP stream (sleep-polling) + the main "
                            "thread; busy = CPU seconds / wall seconds of the timed steps, max over ranks" % G},
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
