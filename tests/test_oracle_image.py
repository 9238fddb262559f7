"""CPU tests pinning the oracle's image code (pyrDown, Scharr, decimation, LK) against
independent numpy/scipy computations.  The reference ships no tests or vectors
(SURVEY §4), so these are what anchors the restatement."""
import numpy as np
import pytest
from scipy import ndimage

import common as cm


def test_pyrdown_matches_scipy_mirror(orc):
    rng = np.random.default_rng(0)
    k = np.array([1, 4, 6, 4, 1], np.int64)
    for (h, w) in ((188, 620), (47, 155), (24, 78), (13, 17), (12, 12)):
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        # full-resolution 5x5 binomial with REFLECT_101 (scipy 'mirror'), then take even samples
        f = ndimage.correlate1d(img.astype(np.int64), k, axis=0, mode="mirror")
        f = ndimage.correlate1d(f, k, axis=1, mode="mirror")
        ref = ((f[::2, ::2] + 128) >> 8).astype(np.uint8)
        got = orc.pyrdown(img)
        assert got.shape == ((h + 1) // 2, (w + 1) // 2)
        assert np.array_equal(got, ref)


def test_pyramid_level_rule(orc):
    # buildOpticalFlowPyramid stops before a level whose size is <= winSize (11)
    assert [l.shape for l in orc.pyramid(np.zeros((188, 620), np.uint8))] == [(188, 620), (94, 310), (47, 155), (24, 78)]
    assert len(orc.pyramid(np.zeros((48, 64), np.uint8))) == 3      # 64x48, 32x24, 16x12; next 8x6 stops
    assert len(orc.pyramid(np.zeros((22, 30), np.uint8))) == 1      # next would be 15x11: 11 <= 11
    assert len(orc.pyramid(np.zeros((188, 620), np.uint8), max_level=1)) == 2


def test_scharr_matches_scipy(orc):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    a = img.astype(np.int64)
    sm = np.array([3, 10, 3]); df = np.array([-1, 0, 1])
    dx = ndimage.correlate1d(ndimage.correlate1d(a, sm, axis=0, mode="mirror"), df, axis=1, mode="mirror")
    dy = ndimage.correlate1d(ndimage.correlate1d(a, df, axis=0, mode="mirror"), sm, axis=1, mode="mirror")
    got = orc.scharr(img)
    assert np.array_equal(got[..., 0], dx) and np.array_equal(got[..., 1], dy)


def test_decimate_is_even_sampling(orc):
    rng = np.random.default_rng(2)
    full = rng.integers(0, 256, (376, 1241), dtype=np.uint8)
    dec = orc.decimate(full)
    assert dec.shape == (188, 620)            # cvRound(620.5) = 620 (half to even), SURVEY F3
    assert np.array_equal(dec, full[0:376:2, 0:1240:2])


def _ref_lk_single_level(I, J, pts, guess, max_iter=30, eps=0.01, win=11):
    """independent numpy restatement of one LK level (OpenCV LKTrackerInvoker, level 0 only,
    float accumulators exactly like the scalar C++ code)"""
    W_BITS = 14
    h, w = I.shape
    B = win
    Ib = np.pad(I.astype(np.int64), B, mode="reflect")
    Jb = np.pad(J.astype(np.int64), B, mode="reflect")
    a = I.astype(np.int64)
    sm = np.array([3, 10, 3]); df = np.array([-1, 0, 1])
    dx = ndimage.correlate1d(ndimage.correlate1d(a, sm, axis=0, mode="mirror"), df, axis=1, mode="mirror")
    dy = ndimage.correlate1d(ndimage.correlate1d(a, df, axis=0, mode="mirror"), sm, axis=1, mode="mirror")
    dxb = np.pad(dx, B); dyb = np.pad(dy, B)
    half = np.float32((win - 1) * 0.5)
    out = guess.astype(np.float32).copy(); status = np.ones(len(pts), np.uint8)

    def weights(fx, fy):
        ix, iy = int(np.floor(fx)), int(np.floor(fy))
        a_ = np.float32(fx - np.float32(ix)); b_ = np.float32(fy - np.float32(iy))
        one = np.float32(1)
        w00 = int(np.rint(np.float32(np.float32((one - a_) * (one - b_)) * np.float32(1 << W_BITS))))
        w01 = int(np.rint(np.float32(np.float32(a_ * (one - b_)) * np.float32(1 << W_BITS))))
        w10 = int(np.rint(np.float32(np.float32((one - a_) * b_) * np.float32(1 << W_BITS))))
        return ix, iy, w00, w01, w10, (1 << W_BITS) - w00 - w01 - w10

    def patch(img_b, ix, iy, ws, shift):
        y0, x0 = iy + B, ix + B
        p = img_b[y0:y0 + win + 1, x0:x0 + win + 1]
        v = p[:-1, :-1] * ws[0] + p[:-1, 1:] * ws[1] + p[1:, :-1] * ws[2] + p[1:, 1:] * ws[3]
        return (v + (1 << (shift - 1))) >> shift

    for n, (p, g) in enumerate(zip(pts.astype(np.float32), guess.astype(np.float32))):
        px, py = np.float32(p[0] - half), np.float32(p[1] - half)
        ix, iy, *ws = weights(px, py)
        if ix < -win or ix >= w or iy < -win or iy >= h:
            status[n] = 0
            continue
        Iw = patch(Ib, ix, iy, ws, W_BITS - 5); Ix = patch(dxb, ix, iy, ws, W_BITS); Iy = patch(dyb, ix, iy, ws, W_BITS)
        sc = np.float32(1.0 / (1 << 20))
        A11 = np.float32(np.float32((Ix * Ix).sum()) * sc); A12 = np.float32(np.float32((Ix * Iy).sum()) * sc)
        A22 = np.float32(np.float32((Iy * Iy).sum()) * sc)
        D = np.float32(A11 * A22 - A12 * A12)
        mine = (A22 + A11 - np.sqrt(np.float32((A11 - A22) ** 2 + np.float32(4) * A12 * A12))) / np.float32(2 * win * win)
        if mine < 1e-4 or D < np.finfo(np.float32).eps:
            status[n] = 0
            continue
        D = np.float32(1) / D
        nx, ny = np.float32(g[0] - half), np.float32(g[1] - half)
        pdx = pdy = np.float32(0)
        for j in range(max_iter):
            jx, jy, *wj = weights(nx, ny)
            if jx < -win or jx >= w or jy < -win or jy >= h:
                status[n] = 0
                break
            diff = patch(Jb, jx, jy, wj, W_BITS - 5) - Iw
            b1 = np.float32(np.float32((diff * Ix).sum()) * sc); b2 = np.float32(np.float32((diff * Iy).sum()) * sc)
            ddx = np.float32(np.float32(A12 * b2 - A22 * b1) * D); ddy = np.float32(np.float32(A12 * b1 - A11 * b2) * D)
            nx = np.float32(nx + ddx); ny = np.float32(ny + ddy)
            out[n] = (nx + half, ny + half)
            if float(ddx) ** 2 + float(ddy) ** 2 <= eps * eps:
                break
            if j > 0 and abs(float(ddx + pdx)) < 0.01 and abs(float(ddy + pdy)) < 0.01:
                out[n] -= np.array([ddx, ddy], np.float32) * np.float32(0.5)
                break
            pdx, pdy = ddx, ddy
    return out, status


def test_lk_level0_matches_independent_numpy(orc):
    rng = np.random.default_rng(5)
    I = cm.textured(rng, 80, 120)
    J = ndimage.shift(I.astype(np.float64), (0.8, -1.3), order=3, mode="mirror")
    J = np.clip(np.rint(J), 0, 255).astype(np.uint8)
    pts = np.stack([rng.uniform(2, 118, 60), rng.uniform(2, 78, 60)], 1).astype(np.float32)
    pts[:4] = [[0, 0], [119, 79], [-20, 5], [60, 200]]
    guess = pts + rng.normal(0, 0.7, pts.shape).astype(np.float32)
    q, st, _ = orc.lk(I, J, pts, guess, params=orc.lk_params(max_level=0))
    q_ref, st_ref = _ref_lk_single_level(I, J, pts, guess)
    assert np.array_equal(st, st_ref)
    ok = st > 0
    # the only declared deviation (exact integer sums vs float accumulation) is far below this
    assert np.abs(q[ok] - q_ref[ok]).max() < 2e-3
    assert ok.sum() > 40


def test_lk_recovers_known_subpixel_shift(orc):
    rng = np.random.default_rng(6)
    I = cm.textured(rng, 188, 620, sigma=3.0)
    for (sy, sx) in ((0.0, 3.25), (-2.5, 6.75), (5.5, -11.0)):
        J = ndimage.shift(I.astype(np.float64), (sy, sx), order=3, mode="mirror")
        J = np.clip(np.rint(J), 0, 255).astype(np.uint8)
        pts = orc.gftt(I, max_corners=120)
        pts = pts[(pts[:, 0] > 40) & (pts[:, 0] < 580) & (pts[:, 1] > 30) & (pts[:, 1] < 158)]
        q, st, err = orc.lk(I, J, pts, pts, params=orc.lk_params(use_initial_flow=0))
        assert st.mean() > 0.9
        d = (q - pts)[st > 0]
        assert np.abs(np.median(d[:, 0]) - sx) < 0.05 and np.abs(np.median(d[:, 1]) - sy) < 0.05
        assert np.percentile(np.abs(d - [sx, sy]).max(1), 90) < 0.15


def test_lk_status_semantics(orc):
    rng = np.random.default_rng(7)
    I = cm.textured(rng, 64, 96)
    flat = np.full((64, 96), 90, np.uint8)
    pts = np.array([[48, 32], [10, 10]], np.float32)
    # flat image: min eigenvalue below threshold at level 0 -> status 0, point unchanged
    q, st, _ = orc.lk(flat, flat, pts, pts + 0.5)
    assert st.tolist() == [0, 0]
    # identical images, guess == truth: converges in place
    q, st, err = orc.lk(I, I, pts, pts.copy())
    assert st.tolist() == [1, 1] and np.abs(q - pts).max() < 0.02 and err.max() < 1.0
    # zero points
    q, st, _ = orc.lk(I, I, np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32))
    assert q.shape == (0, 2)
