"""BASELINE config 4 at its stated length: "synthetic 1241x376 stereo stream, 10k frames, 8 independent streams on
8 x MI355X" — the per-GPU share is one 10 000-frame stream; here the ONE GPU of the test box carries all eight
streams for the full 10 000 frames (src/visual_odometry.cpp:164-171 runs a whole sequence).  No oracle at this
size: the checks are properties — never LOST, trajectory error against the renderer's ground truth, no capacity
event, bounded host memory (the evicting landmark store of host/slam_host.h) — and the run's own frame rate."""
import importlib
import json
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H = 620, 188


def _rss():
    return int(open("/proc/self/statm").read().split()[1]) * os.sysconf("SC_PAGE_SIZE")


# The shape DESIGN quotes for config 4 (a few cameras per GPU): map in HBM + low-latency kernel shapes, at the full 10 000
# frames; the host-resident map with the batch shapes (rounds 1-3) stays at a reduced length.
# Round 6: that shape now reads the config's own frames — 1241x376 in HBM, the reference's 1/2 decimation
# fused into the pyramid kernel (k_pyr_fused<true>).
@pytest.mark.parametrize("shape,frames", [("device_map_low_latency", 10000), ("host_map_batch_shapes", 2500)])
def test_config4_eight_streams_ten_thousand_frames(svs, shape, frames):
    pl = importlib.import_module("stereovision-slam_amd.pipeline")
    S, N, CH = 8, int(os.environ.get("SVS_LONG_FRAMES", str(frames))), 500
    seeds = [0xC0F40000 + i for i in range(S)]
    product = shape == "device_map_low_latency"
    SW, SH = (1241, 376) if product else (W, H)
    cam_r = tuple(2 * v for v in svs.KITTI00_HALF_CAM) if product else svs.KITTI00_HALF_CAM
    cfg = pl.default_config(W, H, host_threads=2, device_map=1, low_latency=1, src_width=SW, src_height=SH) if product else pl.default_config(W, H, host_threads=2)
    pipe = pl.Pipeline(cfg, nstreams=S)
    ctx = svs.Context.borrow(pipe.kernel_ctx(), W, H)
    img = SW * SH
    dl = ctx.dev_alloc(S * CH * img); dr = ctx.dev_alloc(S * CH * img)
    poses = np.zeros((N, S, 7)); status = np.zeros((N, S), np.int32)
    t_run = 0.0
    rss_at = {}
    for f0 in range(0, N, CH):
        n = min(CH, N - f0)
        svs.synth_render_streams_device(seeds, f0, CH, SW, SH, dl, dr, cam=cam_r)
        t0 = time.perf_counter()
        r = pipe.run_device(dl, dr, CH * img, img, 0, n)
        t_run += time.perf_counter() - t0
        poses[f0:f0 + n] = r["pose"]; status[f0:f0 + n] = r["status"]
        rss_at[f0 + n] = _rss()
    cnt = pipe.counters()
    hc = ctx.host_counters()
    ctx.dev_free(dl); ctx.dev_free(dr)
    pipe.close()
    if product:      # every keyframe's local BA went to the multi-workgroup solver, none had to be repeated
        assert hc[6] == cnt["ba_calls"] and hc[7] == 0, (hc, cnt["ba_calls"])
    assert (status != 3).all(), "a stream was LOST at frame %d" % int(np.argmax((status == 3).any(1)))
    assert cnt["corners_dropped"] == 0 and cnt["ba_skipped"] == 0
    rel = []
    for s, sd in enumerate(seeds):
        gt = np.array([svs.synth_gt(sd, f) for f in range(N)])
        path = float(np.linalg.norm(np.diff(pl.camera_centres(gt), axis=0), axis=1).sum())
        rel.append(pl.ate_rmse(poses[:, s], gt) / path)
    # host memory: the landmark store evicts what nothing can reach (16 B of archive per landmark stay for
    # landmarks.pcd): growth per frame per stream between frame 1000 and the end
    first = min(k for k in rss_at if k >= min(1000, N // 2))
    growth = (rss_at[N] - rss_at[first]) / max(1, (N - first) * S)
    fps = N * S / t_run
    line = {"shape": shape, "input": "%dx%d" % (SW, SH), "streams": S, "frames_per_stream": N, "frames_per_s": round(fps, 1), "ms_per_frame_per_stream": round(1e3 * t_run / N, 4),
            "keyframes": int(cnt["keyframes"]), "ba_calls": int(cnt["ba_calls"]),
            "ate_over_path_pct_mean": round(100 * float(np.mean(rel)), 4), "ate_over_path_pct_max": round(100 * float(np.max(rel)), 4),
            "rss_growth_bytes_per_frame_per_stream": round(growth, 1), "rss_mb_end": round(rss_at[N] / 1e6, 1)}
    print("config4 long run:", json.dumps(line))
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "config4_long_run_%s.json" % shape), "w") as f:
            f.write(json.dumps(line) + "\n")
    # Drift of a stereo VO WITHOUT loop closure (SURVEY: LoopClosure is out of scope) grows faster than the path
    # (heading random walk): 0.06-0.21 % over 1.3 km and 0.3 % over 2.6 km in round 1, ~1 % over these 8.5 km.
    # What has to hold is that the HIP path drifts like the reference-shaped CPU twin on the same frames:
    # two streams go through the twin as well (numeric-J BA like g2o, one thread each).
    assert max(rel) <= 0.02, rel
    assert growth <= 700.0, growth
    # (both shapes: ADVICE r5 — the host-map run used to return before the twin comparison)
    twin_rel = _twin_drift(svs, pl, seeds[:2], N)
    line["twin_ate_over_path_pct"] = [round(100 * v, 4) for v in twin_rel]
    line["hip_ate_over_path_pct_same_streams"] = [round(100 * v, 4) for v in rel[:2]]
    print("config4 long run:", json.dumps(line))
    if os.path.isdir(out):
        with open(os.path.join(out, "config4_long_run_%s.json" % shape), "w") as f:
            f.write(json.dumps(line) + "\n")
    # per-stream drift scatters by a factor ~1.5 between two valid runs of the same stream (chaotic in
    # each other); the bound says "same regime", the distribution test (test_gpu_ate_distribution) says "same mean"
    for a, b in zip(rel[:2], twin_rel):
        assert a <= 2.0 * b + 0.002, (a, b)


def _twin_drift(svs, pl, seeds, N, cfg=None, w=W, h=H, cam=None):
    import threading
    import pipe_cpu
    out = [None] * len(seeds)
    errs = []
    cam = cam or svs.KITTI00_HALF_CAM

    def work(k):
        try:
            tw = pipe_cpu.make(cfg or pl.default_config(W, H), nstreams=1)
            est = np.zeros((N, 7))
            for f in range(N):
                l, r = svs.synth_pair(seeds[k], f, w, h, cam)
                est[f] = tw.step([l], [r])["pose"][0]
            tw.close()
            gt = np.array([svs.synth_gt(seeds[k], f) for f in range(N)])
            path = float(np.linalg.norm(np.diff(pl.camera_centres(gt), axis=0), axis=1).sum())
            out[k] = pl.ate_rmse(est, gt) / path
        except Exception as e:   # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=work, args=(k,)) for k in range(len(seeds))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise errs[0]
    return out


def _long_single_stream(svs, pl, pipe, seed, N, w, h, cam, tag, extra=None):
    """one stream, N frames rendered into HBM in chunks of 500; returns poses, status, counters, frames/s"""
    CH = 500
    ctx = svs.Context.borrow(pipe.kernel_ctx(), w, h)
    img = w * h
    dl = ctx.dev_alloc(CH * img); dr = ctx.dev_alloc(CH * img)
    poses = np.zeros((N, 7)); status = np.zeros(N, np.int32)
    t_run = 0.0
    rss0 = None
    for f0 in range(0, N, CH):
        n = min(CH, N - f0)
        svs.synth_render_streams_device([seed], f0, CH, w, h, dl, dr, cam=cam)
        t0 = time.perf_counter()
        r = pipe.run_device(dl, dr, CH * img, img, 0, n)
        t_run += time.perf_counter() - t0
        poses[f0:f0 + n] = r["pose"][:, 0]; status[f0:f0 + n] = r["status"][:, 0]
        if f0 + n >= min(1000, N // 2) and rss0 is None:
            rss0 = (_rss(), f0 + n)
    growth = (_rss() - rss0[0]) / max(1, N - rss0[1])
    cnt = pipe.counters()
    ctx.dev_free(dl); ctx.dev_free(dr)
    gt = np.array([svs.synth_gt(seed, f) for f in range(N)])
    path = float(np.linalg.norm(np.diff(pl.camera_centres(gt), axis=0), axis=1).sum())
    rel = pl.ate_rmse(poses, gt) / path
    line = {"config": tag, "frames": N, "frame": "%dx%d" % (w, h), "frames_per_s": round(N / t_run, 1), "keyframes": int(cnt["keyframes"]),
            "ba_calls": int(cnt["ba_calls"]), "ba_keyframes_mean": round(cnt["ba_kf"] / max(cnt["ba_calls"], 1), 2),
            "ba_landmarks_mean": round(cnt["ba_lm"] / max(cnt["ba_calls"], 1), 1),
            "ate_over_path_pct": round(100 * rel, 4), "path_m": round(path, 1), "rss_growth_bytes_per_frame": round(growth, 1)}
    line.update(extra or {})
    assert (status != 3).all(), "LOST at frame %d" % int(np.argmax(status == 3))
    assert cnt["corners_dropped"] == 0 and cnt["ba_skipped"] == 0 and cnt["ba_calls"] == cnt["keyframes"]
    assert growth <= 1500.0, growth
    return line, rel, cnt


def _publish(name, line):
    print(name + ":", json.dumps(line))
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, name + ".json"), "w") as f:
            f.write(json.dumps(line) + "\n")


def test_config2_full_length_hip_frontend_cpu_backend(svs):
    """BASELINE config 2 at the length SURVEY 8d gives it: 4541 frames (KITTI-00), one stream, HIP GFTT + LK (+ triangulation,
    pose-only) frontend with the backend on the CPU (the oracle's g2o-shaped BA, numeric Jacobians).  Properties: never LOST, one
    BA per keyframe, no capacity event, bounded host memory, drift in the regime of the all-CPU twin on the same stream."""
    import pipe_cpu
    pl = importlib.import_module("stereovision-slam_amd.pipeline")
    os.environ.pop("SVS_ORACLE_BA_JAC", None)
    N, seed = int(os.environ.get("SVS_LONG_FRAMES_C2", "4541")), 0xC0F20000
    hyb = pipe_cpu.make_hybrid(pl.default_config(W, H), nstreams=1)
    assert hyb.kernel_ctx()
    line, rel, cnt = _long_single_stream(svs, pl, hyb, seed, N, W, H, svs.KITTI00_HALF_CAM, "2: HIP frontend + CPU (oracle) local BA")
    hyb.close()
    twin_rel = _twin_drift(svs, pl, [seed], N)[0]
    line["twin_ate_over_path_pct"] = round(100 * twin_rel, 4)
    _publish("config2_full_length", line)
    assert rel <= 0.02 and rel <= 2.0 * twin_rel + 0.002, (rel, twin_rel)


def test_config3_full_length_seq05_shape_seven_keyframes(svs):
    """BASELINE config 3 at its length: 2761 frames of KITTI-05-shaped input (1226x370 -> 613x185: odd width), HIP frontend + HIP
    local BA over the last 7 keyframes, one stream on the one-camera product shape (device map, low-latency kernels)."""
    pl = importlib.import_module("stereovision-slam_amd.pipeline")
    os.environ.pop("SVS_ORACLE_BA_JAC", None)
    w, h = 613, 185
    cam = (353.5455, 353.5455, 300.9435, 91.55515)
    N, seed = int(os.environ.get("SVS_LONG_FRAMES_C3", "2761")), 0xC0F30000
    cfg = pl.default_config(w, h, cam=cam, num_active_keyframes=7, device_map=1, low_latency=1)
    pipe = pl.Pipeline(cfg, nstreams=1)
    line, rel, cnt = _long_single_stream(svs, pl, pipe, seed, N, w, h, cam, "3: HIP frontend + HIP local BA, 7-keyframe window")
    pipe.close()
    assert cnt["ba_kf"] <= 7 * cnt["ba_calls"] and cnt["ba_kf"] >= 6.9 * (cnt["ba_calls"] - 7)
    twin_rel = _twin_drift(svs, pl, [seed], N, cfg=pl.default_config(w, h, cam=cam, num_active_keyframes=7), w=w, h=h, cam=cam)[0]
    line["twin_ate_over_path_pct"] = round(100 * twin_rel, 4)
    _publish("config3_full_length", line)
    assert rel <= 0.02 and rel <= 2.0 * twin_rel + 0.002, (rel, twin_rel)
