"""GPU tests for the edges of the C ABI: limits, error returns, ragged batches,
maximum sizes, and the >64-row reduced camera system."""
import numpy as np
import pytest

import common as cm

pytestmark = pytest.mark.gpu


def test_error_returns_not_crashes(svs):
    c = svs.Context(cm.W, cm.H, max_slots=2, max_jobs=2, max_pts=64, max_corners=50, max_kf=3, max_lm=64, max_obs=256)
    img = np.zeros((cm.H, cm.W), np.uint8)
    with pytest.raises(RuntimeError, match="slot"):
        c.pyramid([5], [img])
    with pytest.raises(RuntimeError, match="jobs"):
        c.pyramid([0, 1, 0], [img, img, img])
    c.pyramid([0, 1], [img, img])
    p = np.zeros((10, 2), np.float32)
    with pytest.raises(RuntimeError, match="slot"):
        c.lk([(0, 7, p, p)])
    with pytest.raises(RuntimeError):
        c.gftt([(0, None)], max_corners=51)                      # above the context's max_corners
    big = np.zeros((65, 3)); uv = np.zeros((65, 2), np.float32)
    with pytest.raises(RuntimeError, match="pose_only"):
        c.pose_only([(cm.EXT_L, big, uv)], cm.CAM)              # 65 points > max_pts
    rng = np.random.default_rng(0)
    pr = cm.make_ba_problem(rng, 3, 40)
    bad_kf = pr["okf"].copy(); bad_kf[3] = 9
    with pytest.raises(RuntimeError, match="out of range"):
        c.local_ba([(pr["poses0"], pr["pts0"], bad_kf, pr["olm"], pr["ori"], pr["ouv"])], cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)
    pr2 = cm.make_ba_problem(rng, 4, 40)
    with pytest.raises(RuntimeError, match="limits"):
        c.local_ba([(pr2["poses0"], pr2["pts0"], pr2["okf"], pr2["olm"], pr2["ori"], pr2["ouv"])], cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)
    # the context is still usable after errors
    res = c.local_ba([(pr["poses0"], pr["pts0"], pr["okf"], pr["olm"], pr["ori"], pr["ouv"])], cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)
    assert res[0][3] >= 1
    c.close()
    with pytest.raises(RuntimeError):
        svs.Context(8, 8)                                        # below the minimum image size
    with pytest.raises(RuntimeError):
        svs.Context(cm.W, cm.H, max_pts=4096)                    # above what the pose-only kernel holds


def test_ragged_batch_and_max_sizes(svs, orc):
    l0, r0 = svs.synth_pair(31, 0)
    l1, _ = svs.synth_pair(31, 1)
    c = svs.Context(cm.W, cm.H, max_slots=4, max_jobs=6, max_pts=512, max_corners=1024, max_kf=2, max_lm=8, max_obs=8)
    c.pyramid([0, 1, 2], [l0, r0, l1])
    many = orc.gftt(l0, None, 1024, 0.0005, 2.0)
    assert len(many) >= 512
    pts512 = many[:512]
    sizes = [512, 1, 0, 77, 3, 512]
    jobs = [(0, 2 if i % 2 else 1, pts512[:n], pts512[:n] + 0.5) for i, n in enumerate(sizes)]
    res = c.lk(jobs)
    for (q, st, err), (ps, ns, p, g) in zip(res, jobs):
        q_ref, st_ref, err_ref = orc.lk(l0, l1 if ns == 2 else r0, p, g)
        assert np.array_equal(st, st_ref) and np.array_equal(q.view(np.uint32), q_ref.view(np.uint32))
    # GFTT with the largest corner budget and a tiny min distance
    (cr,) = c.gftt([(0, None)], max_corners=1024, quality=0.0005, min_dist=2.0)
    assert np.array_equal(cr, many[:len(cr)]) and len(cr) == min(1024, len(many))
    # pose-only at the 512-edge limit
    rng = np.random.default_rng(1)
    P = np.stack([rng.uniform(-8, 8, 512), rng.uniform(-3, 1.5, 512), rng.uniform(5, 50, 512)], 1)
    T_true = cm.random_pose(rng, 0.5, 0.03)
    uv, _ = cm.project(cm.CAM, T_true, cm.EXT_L, P)
    uv = (uv + rng.normal(0, 0.4, uv.shape)).astype(np.float32)
    (T, outl, ninl), (T2, o2, n2) = c.pose_only([(cm.EXT_L, P, uv), (cm.EXT_L, P[:1], uv[:1])], cm.CAM)
    T_ref, outl_ref, ninl_ref = orc.pose_only(cm.CAM, cm.EXT_L, P, uv)
    assert np.allclose(T, T_ref, atol=1e-6) and np.array_equal(outl, outl_ref) and ninl == ninl_ref
    T2r, o2r, n2r = orc.pose_only(cm.CAM, cm.EXT_L, P[:1], uv[:1])
    assert np.allclose(T2, T2r, atol=1e-6) and n2 == n2r
    c.close()


def test_ba_more_than_64_rows_and_inactive_vertices(svs, orc):
    rng = np.random.default_rng(2)
    c = svs.Context(cm.W, cm.H, max_slots=1, max_jobs=3, max_kf=12, max_lm=3000, max_obs=20000)
    big = cm.make_ba_problem(rng, 12, 2500)            # np = 72 > 64 rows; 2500 landmarks
    keep = rng.random(len(big["okf"])) < 0.25
    small = cm.make_ba_problem(rng, 2, 6)
    gap = cm.make_ba_problem(rng, 6, 200)              # remove all edges of one pose and a few landmarks
    m = (gap["okf"] != 2) & (gap["olm"] % 17 != 0)
    jobs = [(big["poses0"], big["pts0"], big["okf"][keep], big["olm"][keep], big["ori"][keep], big["ouv"][keep]),
            (small["poses0"], small["pts0"], small["okf"], small["olm"], small["ori"], small["ouv"]),
            (gap["poses0"], gap["pts0"], gap["okf"][m], gap["olm"][m], gap["ori"][m], gap["ouv"][m])]
    res = c.local_ba(jobs, cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)
    for (poses, pts, chi2, it), job in zip(res, jobs):
        pa, xa, ca, ia = orc.local_ba(cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R, *job, jac_mode=0)
        assert it == ia
        assert np.allclose(poses[:, 4:], pa[:, 4:], atol=1e-6) and np.allclose(poses[:, :4], pa[:, :4], atol=1e-7)
        assert np.allclose(pts, xa, rtol=1e-6, atol=1e-6)
        assert np.allclose(chi2, ca, rtol=1e-5, atol=1e-6)
    # vertices without edges are untouched (g2o: not part of the active problem)
    assert np.array_equal(res[2][0][2], gap["poses0"][2])
    assert np.array_equal(res[2][1][0], gap["pts0"][0]) and np.array_equal(res[2][1][17], gap["pts0"][17])
    # zero-edge job
    (p0, x0, ch0, it0), = c.local_ba([(small["poses0"], small["pts0"], np.zeros(0, np.int32), np.zeros(0, np.int32),
                                       np.zeros(0, np.uint8), np.zeros((0, 2), np.float32))], cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)
    assert it0 == 0 and np.array_equal(p0, small["poses0"]) and np.array_equal(x0, small["pts0"])
    c.close()


def test_full_resolution_decimating_path_matches_host_decimation(svs, orc):
    """f3: 1241x376 frames, the reference's 1/2 INTER_NEAREST resize fused into the level-0 write"""
    rng = np.random.default_rng(3)
    from scipy import ndimage
    full = ndimage.gaussian_filter(rng.random((376, 1241)), 1.5)
    full = np.clip((full - full.min()) / (full.max() - full.min()) * 255, 0, 255).astype(np.uint8)
    c = svs.Context(620, 188, max_slots=2, max_jobs=2, max_kf=0, max_lm=0, max_obs=0)
    c.pyramid([0], [full], decimate_from=(1241, 376))
    dec = orc.decimate(full)
    c.pyramid([1], [dec])
    for lvl in range(4):
        assert np.array_equal(c.pyramid_read(0, lvl), c.pyramid_read(1, lvl))
    (a, b) = c.gftt([(0, None), (1, None)])
    assert np.array_equal(a, b) and np.array_equal(a, orc.gftt(dec))
    # device-resident source
    d = c.dev_alloc(full.size)
    c.dev_upload(d, full)
    c.pyramid([0], [d], device=True, strides=[1241], decimate_from=(1241, 376))
    assert np.array_equal(c.pyramid_read(0, 2), orc.pyramid(dec)[2])
    c.dev_free(d)
    c.close()


def test_contexts_sharing_one_stream_stay_independent(svs, monkeypatch):
    """With the stream pool capped at 1 every context lands on the same HIP stream; two threads
    driving their own contexts concurrently must still get their own (correct) results, because
    a context waits on its completion event and never on the stream."""
    import threading
    import oracle_lib as orc
    monkeypatch.setenv("SVSLAM_MAX_STREAMS", "1")
    frames = [svs.synth_pair(11, f) for f in range(2)]
    l0, r0 = frames[0]
    pts = orc.gftt(l0)
    want_t = orc.lk(l0, frames[1][0], pts, pts)
    want_s = orc.lk(l0, r0, pts, pts)
    out = {}

    def work(name, second, want):
        c = svs.Context(cm.W, cm.H, max_slots=2, max_jobs=2, max_kf=0, max_lm=0, max_obs=0)
        ok = True
        for _ in range(20):
            c.pyramid([0, 1], [l0, second])
            q, st, _ = c.lk([(0, 1, pts, pts)])[0]
            ok = ok and np.array_equal(q.view(np.uint32), want[0].view(np.uint32)) and np.array_equal(st, want[1])
        c.close()
        out[name] = ok

    th = [threading.Thread(target=work, args=("t", frames[1][0], want_t)),
          threading.Thread(target=work, args=("s", r0, want_s))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert out == {"t": True, "s": True}


def test_resident_track_edge_cases(svs):
    """empty resident list, a frame on which every point is lost, and the count check"""
    import oracle_lib as orc
    l0, _ = svs.synth_pair(12, 0)
    l1, _ = svs.synth_pair(12, 1)
    pts = orc.gftt(l0)[:40]
    c = svs.Context(cm.W, cm.H, max_slots=4, max_jobs=2, max_kf=0, max_lm=0, max_obs=0, max_streams=3)
    c.pyramid([0, 2], [l0, l0])
    T = cm.EXT_L.copy()
    # stream 0: nothing resident; stream 2: 40 features without map points
    c.rtrack_upload([(2, pts, np.full(len(pts), -1, np.int32), np.zeros((len(pts), 3)))])
    r0, r2 = c.rtrack([(0, 0, 1, l1, T, T, 0), (2, 2, 3, l1, T, T, len(pts))], cm.CAM)
    assert r0["n_tracked"] == 0 and r0["n_edges"] == 0 and np.array_equal(r0["pose"], T)
    q, st, _ = orc.lk(l0, l1, pts, pts)
    inb = (q[:, 0] >= 0) & (q[:, 0] < cm.W) & (q[:, 1] >= 0) & (q[:, 1] < cm.H)
    keep = (st > 0) & inb
    assert r2["n_tracked"] == keep.sum() and r2["n_edges"] == 0
    assert np.array_equal(r2["xy"].view(np.uint32), q[keep].view(np.uint32)) and (r2["mp"] == -1).all()
    # next frame is blank: every point fails the min-eigenvalue test, the list becomes empty
    blank = np.full_like(l0, 77)
    c.pyramid([2], [blank])      # previous pyramid of stream 2 replaced by a textureless one
    (r,) = c.rtrack([(2, 2, 3, blank, T, T, int(keep.sum()))], cm.CAM)
    assert r["n_tracked"] == 0
    (r,) = c.rtrack([(2, 3, 2, l1, T, T, 0)], cm.CAM)
    assert r["n_tracked"] == 0
    with pytest.raises(RuntimeError, match="holds 0"):
        c.rtrack([(2, 2, 3, l1, T, T, 5)], cm.CAM)
    with pytest.raises(RuntimeError, match="out of range"):
        c.rtrack([(7, 2, 3, l1, T, T, 0)], cm.CAM)
    c.close()


def test_full_size_batch_properties_without_oracle(svs):
    """192 KITTI-shaped frames in one batch per call (the size the oracle does not finish in seconds):
    domain properties that need no reference — GFTT corners are integer, at least min_dist apart,
    outside the mask squares and at most max_corners; tracking an image onto itself is the identity;
    stereo matches triangulate to points that re-project onto both pixels."""
    n = 192
    frames = [svs.synth_pair(500 + i, i % 7) for i in range(n)]
    c = svs.Context(cm.W, cm.H, max_slots=2 * n, max_jobs=2 * n, max_pts=256, max_corners=150, max_kf=0, max_lm=0, max_obs=0)
    c.pyramid(list(range(2 * n)), [f[0] for f in frames] + [f[1] for f in frames])
    rng = np.random.default_rng(3)
    rects = [rng.uniform([20, 20], [cm.W - 20, cm.H - 20], (40, 2)).astype(np.float32) for _ in range(n)]
    corners = c.gftt([(i, rects[i]) for i in range(n)])
    for i, k in enumerate(corners):
        assert 20 <= len(k) <= 150
        assert np.array_equal(k, np.round(k)) and (k >= 1).all() and (k[:, 0] <= cm.W - 2).all() and (k[:, 1] <= cm.H - 2).all()
        d = np.linalg.norm(k[:, None, :] - k[None, :, :], axis=2) + 1e9 * np.eye(len(k))
        assert d.min() >= 20.0
        lo, hi = np.rint(rects[i] - 10), np.rint(rects[i] + 10)          # cv::rectangle on rounded corners, inclusive
        inside = ((k[:, None, :] >= lo[None]) & (k[:, None, :] <= hi[None])).all(2)
        assert not inside.any()
    # identity: I -> I from the exact positions
    res = c.lk([(i, i, corners[i], corners[i]) for i in range(n)])
    for (q, st, err), k in zip(res, corners):
        assert st.all() and np.array_equal(q, k) and (err == 0).all()
    # stereo: left -> right, triangulate, re-project
    res = c.lk([(i, n + i, corners[i], corners[i]) for i in range(n)])
    tri = c.triangulate([(corners[i], res[i][0], None, 0.0) for i in range(n)], cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)
    nok = 0
    for i in range(n):
        q, st, _ = res[i]
        xyz, ok = tri[i]
        m = (st > 0) & (ok > 0)
        nok += int(m.sum())
        ident = np.array([0, 0, 0, 1, 0, 0, 0], float)
        ul, zl = cm.project(cm.CAM, ident, cm.EXT_L, xyz[m])
        ur, zr = cm.project(cm.CAM, ident, cm.EXT_R, xyz[m])
        assert (zl > 0).all() and (zr > 0).all()
        # the DLT point minimises the algebraic error: it re-projects within a fraction of the
        # vertical disparity the two measurements disagree by
        dv = np.abs(corners[i][m][:, 1] - q[m][:, 1])
        assert (np.abs(ul - corners[i][m]).max(1) <= 0.6 * dv + 1e-3).all()
        assert (np.abs(ur - q[m]).max(1) <= 0.6 * dv + 1e-3).all()
    assert nok > 60 * n
    c.close()


def test_local_ba_batch_properties_without_oracle(svs):
    """48 full-size problems (10 keyframes, ~1200 landmarks, ~8000 edges) in one launch: a problem gives
    bit-identical results wherever it sits in the batch, the same result (to rounding) when its edges
    arrive in another order, a lower robust cost than it started with, and — without noise — the
    reprojection error of the truth."""
    rng = np.random.default_rng(77)
    base = [cm.make_ba_problem(rng, 10, 1200) for _ in range(15)] + [cm.make_ba_problem(rng, 10, 1200, noise=0.0, outlier_frac=0.0)]
    job = lambda p, perm=None: (p["poses0"], p["pts0"]) + tuple(p[k] if perm is None else p[k][perm] for k in ("okf", "olm", "ori", "ouv"))
    jobs = [job(p) for p in base] + [job(p) for p in base[::-1]] + [job(p, rng.permutation(len(p["okf"]))) for p in base]
    nobs = max(len(p["okf"]) for p in base)
    c = svs.Context(cm.W, cm.H, max_slots=1, max_jobs=len(jobs), max_kf=10, max_lm=1200, max_obs=nobs)
    res = c.local_ba(jobs, cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)
    huber = lambda chi2: np.where(chi2 <= 5.991 ** 2, chi2, 2 * 5.991 * np.sqrt(chi2) - 5.991 ** 2).sum()
    for i, p in enumerate(base):
        poses, pts, chi2, it = res[i]
        poses_d, pts_d, chi2_d, it_d = res[2 * len(base) - 1 - i]
        assert np.array_equal(poses, poses_d) and np.array_equal(pts, pts_d) and np.array_equal(chi2, chi2_d) and it == it_d
        poses_s, pts_s, chi2_s, it_s = res[2 * len(base) + i]
        assert it_s == it and np.allclose(poses_s, poses, atol=1e-9) and np.allclose(pts_s, pts, atol=1e-8)
        # cost at the start: reprojection of the perturbed state (numpy), against the kernel's final edge chi2
        start = 0.0
        for cam_i, ext in enumerate((cm.EXT_L, cm.EXT_R)):
            m = p["ori"] == cam_i
            for k in range(10):
                mk = m & (p["okf"] == k)
                uv, _ = cm.project(cm.CAM, p["poses0"][k], ext, p["pts0"][p["olm"][mk]])
                start += huber(((uv - p["ouv"][mk]) ** 2).sum(1))
        # (5 % gross outliers at sigma 25 px keep a Huber floor of ~60 % of the initial cost)
        assert huber(chi2) < 0.75 * start and np.median(chi2) < 1.0, (huber(chi2), start, np.median(chi2))
    # the noise-free problem ends at (numerically) zero reprojection error: f32 pixel rounding only
    assert res[len(base) - 1][2].max() < 1e-4 and np.median(res[len(base) - 1][2]) < 1e-7
    c.close()


def test_triangulate_exactly_singular_matches(svs, orc):
    """A stereo match whose row equals the left feature's to the last bit (LK left the y coordinate untouched: about one match
    in 10 000 of the synthetic streams) makes two rows of the DLT matrix equal: the system is exactly singular and the Jacobi
    sweeps of the oracle never report convergence (60 sweeps of identity rotations).  The kernel leaves at the first sweep that
    changes nothing — the outputs must still be the oracle's: flags exact, positions of the accepted points to 1e-9, and with
    the SVD compiled without FMA contraction (round 5) bit for bit.  Mixed with ordinary matches in one job, plus identical
    left / right pixels (zero disparity)."""
    rng = np.random.default_rng(77)
    n = 192
    fx, fy, cx, cy = cm.CAM
    Z = rng.uniform(3, 60, n); X = rng.uniform(-8, 8, n); Y = rng.uniform(-2, 1.5, n)
    l = np.stack([fx * X / Z + cx, fy * Y / Z + cy], 1)
    r = np.stack([fx * (X - cm.BASELINE) / Z + cx, fy * Y / Z + cy], 1)
    l += rng.normal(0, 0.3, l.shape); r += rng.normal(0, 0.3, r.shape)
    l = l.astype(np.float32); r = r.astype(np.float32)
    l[::3] = np.round(l[::3])                  # GFTT corners are integers
    r[::3, 1] = l[::3, 1]                      # ... and these matches kept the row exactly
    r[5] = l[5]                                # zero disparity
    c = svs.Context(cm.W, cm.H, max_slots=1, max_jobs=2, max_pts=256, max_kf=0, max_lm=0, max_obs=0)
    T = cm.random_pose(np.random.default_rng(2))
    jobs = [(l, r, None, 0.0), (l, r, T, 40.0)]
    res = c.triangulate(jobs, cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)
    for (xyz, ok), (ul, ur, Tj, zmax) in zip(res, jobs):
        xyz_ref, ok_ref = orc.triangulate(cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R, ul, ur, Tj, zmax)
        assert np.array_equal(ok, ok_ref)
        m = ok_ref > 0
        assert m.sum() > 100
        assert np.allclose(xyz[m], xyz_ref[m], rtol=1e-9, atol=1e-9)
        if Tj is None:
            assert np.array_equal(xyz[m], xyz_ref[m])           # camera frame == world frame: nothing but the SVD and three quotients
    c.close()
