"""The drop-in facade (host/slam_facade.h: Frontend / Backend / Dataset / VisualOdometry with the
reference's names and signatures) on a KITTI-layout sequence directory: calib.txt with the four 3x4
projection matrices, image_0/ and image_1/ with 8-bit PNG files at the camera's full resolution.
The C++ test program (tests/cpp/facade_kitti.cpp) drives it the way the reference's VisualOdometry::run
does; this file writes the fixture, runs the program and compares every frame with the batched host
pipeline fed with the same frames.  CPU variant (oracle kernels) and GPU variant (HIP kernels)."""
import importlib
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX, CX, CY, B = 718.856, 607.1928, 185.2157, 0.537166          # KITTI-00 calibration


def _png_gray(path, img, filt):
    """minimal PNG writer (8-bit grey, filter type `filt` on every row) — exercises the reader's un-filtering"""
    h, w = img.shape
    a = img.astype(np.int32)
    left = np.concatenate([np.zeros((h, 1), np.int32), a[:, :-1]], 1)
    up = np.concatenate([np.zeros((1, w), np.int32), a[:-1]], 0)
    ul = np.concatenate([np.zeros((h, 1), np.int32), up[:, :-1]], 1)
    if filt == 0:
        r = a
    elif filt == 1:
        r = a - left
    elif filt == 2:
        r = a - up
    elif filt == 3:
        r = a - ((left + up) >> 1)
    else:
        p = left + up - ul
        pa, pb, pc = np.abs(p - left), np.abs(p - up), np.abs(p - ul)
        pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, ul))
        r = a - pred
    raw = np.concatenate([np.full((h, 1), filt, np.uint8), (r & 255).astype(np.uint8)], 1).tobytes()

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def _make_sequence(svs, root, seed, nframes, extra_yaml=""):
    seq = os.path.join(root, "sequences", "00")
    os.makedirs(os.path.join(seq, "image_0")); os.makedirs(os.path.join(seq, "image_1"))
    P = lambda tx: "%.12e 0 %.12e %.12e 0 %.12e %.12e 0 0 0 1 0" % (FX, CX, tx, FX, CY)
    with open(os.path.join(seq, "calib.txt"), "w") as f:
        f.write("P0: " + P(0.0) + "\nP1: " + P(-FX * B) + "\nP2: " + P(0.0) + "\nP3: " + P(-FX * B) + "\n")
    frames = []
    for i in range(nframes):
        l, r = svs.synth_pair(seed, i, w=1241, h=376, cam=(FX, FX, CX, CY), baseline=B)
        _png_gray(os.path.join(seq, "image_0", "%06d.png" % i), l, i % 5)
        if i % 2 == 0:
            _png_gray(os.path.join(seq, "image_1", "%06d.png" % i), r, (i + 2) % 5)
        else:                                    # the reader also takes binary PGM
            with open(os.path.join(seq, "image_1", "%06d.pgm" % i), "wb") as f:
                f.write(b"P5\n# kitti-shaped fixture\n1241 376\n255\n" + r.tobytes())
        frames.append((l, r))
    cfg = os.path.join(root, "config.yaml")
    with open(cfg, "w") as f:
        f.write("%YAML:1.0\n# written by tests/test_facade_kitti.py\ndataset_dir: \"" + seq + "\"\nleft_cam_index: 0\nright_cam_index: 1\n"
                "is_color_input: 0\noutput_dir: " + root + "\nnum_features: 150\nnum_features_init: 50\nnum_features_tracking: 50\n"
                "num_features_tracking_bad: 20\nnum_features_needed_for_keyframe: 80\nmax_triangulation_depth: 300.0\n"
                "keypoint_feature_detector: GFTT\nnum_active_keyframes: 10\nbackend_on: 1\nchi2_th: 5.991\nloopclosure_on: 0\nvisualizer_on: 0\n" + extra_yaml)
    return cfg, seq, frames


def _run_facade(exe, cfg, out):
    r = subprocess.run([exe, cfg, out], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "facade ok" in r.stdout
    rows = [l.split() for l in r.stdout.splitlines() if l.startswith("frame ")]
    meta = np.array([[int(x[1]), int(x[3]), int(x[5]), int(x[7]), int(x[9]), int(x[11])] for x in rows])
    poses = np.array([[float(v) for v in x[13:20]] for x in rows])
    cams = [l.split() for l in r.stdout.splitlines() if l.startswith("cam")]
    return meta, poses, cams, r.stdout


def _check(meta, poses, cams, frames, pl, make_pipeline, low_latency=0):
    # Dataset::initialize: K halved, baseline = |K^-1 t| (src/dataset.cpp:63-77)
    assert [float(v) for v in cams[0][1:6]] == pytest.approx([FX / 2, FX / 2, CX / 2, CY / 2, 0.0], abs=1e-6)
    assert [float(v) for v in cams[1][1:6]] == pytest.approx([FX / 2, FX / 2, CX / 2, CY / 2, B], abs=1e-6)
    assert [float(v) for v in cams[1][7:10]] == pytest.approx([-B, 0.0, 0.0], abs=1e-9)
    n = len(frames)
    assert len(meta) == n and (meta[:, 0] == np.arange(n)).all()
    assert meta[0, 2] == 1 and meta[0, 1] == 1 and meta[:, 1].min() >= 1 and meta[:, 1].max() <= 2   # tracking good / bad, never lost
    # against the batched pipeline on the same frames up to the pause (Backend::PauseRequest at frame 8 has
    # no counterpart in the C API): identical metadata and poses, bit for bit — same host code, same kernels
    half = (0.5 * FX, 0.5 * FX, 0.5 * CX, 0.5 * CY)
    p = make_pipeline(pl.default_config(620, 188, cam=half, baseline=B, src_width=1241, src_height=376, resident_track=0,
                                        low_latency=low_latency))
    kf_before_pause = int(meta[:8, 2].sum())
    for i in range(8):
        r = p.step([frames[i][0]], [frames[i][1]])
        assert (int(r["status"][0]), int(r["is_keyframe"][0]), int(r["n_features"][0]), int(r["n_inliers"][0])) == \
            (meta[i, 1], meta[i, 2], meta[i, 4], meta[i, 5]), i
        assert np.array_equal(r["pose"][0], poses[i]), (i, r["pose"][0], poses[i])
    assert p.counters()["keyframes"] == kf_before_pause
    p.close()


def _build(tmp_path, oracle):
    exe = str(tmp_path / ("facade_" + ("cpu" if oracle else "hip")))
    cmd = ["g++", "-O3", "-march=native", "-ffp-contract=off", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "facade_kitti.cpp")]
    if oracle:
        import oracle_lib
        oracle_lib.build()
        cmd += ["-DFACADE_ORACLE"] + [os.path.join(ROOT, "oracle", "_build", o) for o in ("orc_image.o", "orc_gftt.o", "orc_geom.o")]
    else:
        lib = os.path.join(ROOT, "stereovision-slam_amd", "lib")
        cmd += ["-L" + lib, "-lsvslam_hip", "-Wl,-rpath," + lib]
    subprocess.check_call(cmd + ["-o", exe, "-lz", "-lm"])
    return exe


def test_facade_on_kitti_layout_sequence_cpu(svs, tmp_path):
    import pipe_cpu
    pl = importlib.import_module("stereovision-slam_amd.pipeline")
    cfg, seq, frames = _make_sequence(svs, str(tmp_path), 41, 16)
    meta, poses, cams, out = _run_facade(_build(tmp_path, True), cfg, str(tmp_path))
    _check(meta, poses, cams, frames, pl, lambda c: pipe_cpu.make(c, nstreams=1))
    # the outputs DenseReconstruction reads (src/visual_odometry.cpp:198-310)
    kf = open(os.path.join(str(tmp_path), "keyframes.txt")).read().splitlines()
    assert kf[0] == seq and kf[1] == "0" and len(kf) == 2 + int(meta[:, 2].sum())
    assert open(os.path.join(str(tmp_path), "landmarks.pcd")).read().startswith("# .PCD v0.7")


@pytest.mark.gpu
def test_facade_on_kitti_layout_sequence_gpu(svs, tmp_path):
    """the classes a user of the reference links against, on the HIP kernels — by default with the stream's map in device
    memory (FrontendOptions::device_map = 1, the configuration every throughput figure uses), and once more with
    `device_map: 0` in the YAML: every frame line, Backend::UpdateMap() from outside, keyframes.txt and landmarks.pcd
    must agree bit for bit"""
    pl = importlib.import_module("stereovision-slam_amd.pipeline")
    cfg, seq, frames = _make_sequence(svs, str(tmp_path), 42, 24)
    exe = _build(tmp_path, False)
    meta, poses, cams, out = _run_facade(exe, cfg, str(tmp_path))
    assert "map: device" in out
    # one camera by construction: the facade selects the low-latency kernel shapes — pose-only on four
    # waves, every keyframe's local BA on the multi-workgroup solver (none repeated by the batch solver on an idle GPU)
    print([l for l in out.splitlines() if l.startswith(("shape", "local BA"))])
    assert "shape: low-latency; local BA over 16 workgroups per problem" in out
    ll = [l.split() for l in out.splitlines() if l.startswith("local BA:")][0]
    assert int(ll[2]) >= 1 and int(ll[8]) == 0, ll
    _check(meta, poses, cams, frames, pl, lambda c: pl.Pipeline(c, nstreams=1), low_latency=1)
    gt = np.array([svs.synth_gt(42, f) for f in range(len(frames))])
    assert pl.ate_rmse(poses, gt) < 0.1
    files = {f: open(os.path.join(str(tmp_path), f)).read() for f in ("keyframes.txt", "landmarks.pcd")}
    # the host-resident map on the same sequence
    cfg_h = os.path.join(str(tmp_path), "config_hostmap.yaml")
    open(cfg_h, "w").write(open(cfg).read() + "device_map: 0\n")
    out_h_dir = os.path.join(str(tmp_path), "hostmap"); os.makedirs(out_h_dir)
    meta_h, poses_h, cams_h, out_h = _run_facade(exe, cfg_h, out_h_dir)
    assert "map: host" in out_h
    assert np.array_equal(meta, meta_h) and np.array_equal(poses, poses_h)
    pick = lambda o: [l for l in o.splitlines() if l.startswith(("frame ", "update_map", "frames "))]
    assert pick(out) == pick(out_h)
    for f, txt in files.items():
        assert open(os.path.join(out_h_dir, f)).read() == txt, f
    # `low_latency: 0`: the batch shapes — the results of the batched pipeline's default configuration, bit for bit, and
    # within rounding of the low-latency run (another summation order)
    cfg_b = os.path.join(str(tmp_path), "config_batch.yaml")
    open(cfg_b, "w").write(open(cfg).read() + "low_latency: 0\n")
    out_b_dir = os.path.join(str(tmp_path), "batch"); os.makedirs(out_b_dir)
    meta_b, poses_b, cams_b, out_b = _run_facade(exe, cfg_b, out_b_dir)
    assert "shape: batch" in out_b and "local BA: 0 problems on the low-latency solver" in out_b
    _check(meta_b, poses_b, cams_b, frames, pl, lambda c: pl.Pipeline(c, nstreams=1), low_latency=0)
    assert np.array_equal(meta_b[:, :4], meta[:, :4])
    # (per call the shapes agree at the LM tolerances; over frames the difference feeds back through LK's stopping rule)
    assert np.abs(poses_b[:8] - poses[:8]).max() < 5e-5 and np.abs(poses_b - poses).max() < 5e-2
