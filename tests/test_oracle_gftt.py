"""CPU tests pinning the oracle's GFTT against an independent numpy float32 computation
of the declared operation order, and against the defining properties of the detector."""
import numpy as np

import common as cm


def _eig_numpy(img):
    f = np.float32
    p = np.pad(img.astype(np.float32), 1, mode="reflect")
    s1 = f(1.0 / 3060.0); s2 = f(2.0 * (1.0 / 3060.0))
    c = lambda dy, dx: p[1 + dy:p.shape[0] - 1 + dy, 1 + dx:p.shape[1] - 1 + dx]
    d0 = c(-1, 1) - c(-1, -1); d1 = c(0, 1) - c(0, -1); d2 = c(1, 1) - c(1, -1)
    Dx = (d0 + d2) * s1 + d1 * s2
    c0 = (s1 * c(-1, -1) + s2 * c(-1, 0)) + s1 * c(-1, 1)
    c2 = (s1 * c(1, -1) + s2 * c(1, 0)) + s1 * c(1, 1)
    Dy = c2 - c0
    assert Dx.dtype == np.float32 and Dy.dtype == np.float32
    out = []
    for m in (Dx * Dx, Dx * Dy, Dy * Dy):
        q = np.pad(m, 1, mode="reflect").astype(np.float64)
        s = np.zeros_like(m, dtype=np.float64)
        for j in range(3):           # same accumulation order as the oracle (rows outer)
            for i in range(3):
                s = s + q[j:j + m.shape[0], i:i + m.shape[1]]
        out.append(s.astype(np.float32))
    a = out[0] * f(0.5); b = out[1]; cc = out[2] * f(0.5)
    t = a - cc
    return (a + cc) - np.sqrt(t * t + b * b)


def test_min_eig_map_bit_exact_vs_numpy(orc):
    rng = np.random.default_rng(0)
    for (h, w) in ((48, 64), (188, 620), (17, 23)):
        img = cm.textured(rng, h, w) if h > 20 else rng.integers(0, 256, (h, w), dtype=np.uint8)
        e = orc.min_eig_map(img)
        r = _eig_numpy(img)
        assert r.dtype == np.float32
        assert np.array_equal(e.view(np.uint32), r.view(np.uint32)), np.abs(e - r).max()


def test_mask_rounding_half_even_inclusive_clipped(orc):
    m = orc.gftt_mask(64, 48, np.array([[20.5, 10.5], [0.0, 0.0], [63.0, 47.0], [-30, -30]], np.float32))
    # 20.5-10 = 10.5 -> 10 (half to even), 20.5+10 = 30.5 -> 30 ; 10.5-10 = 0.5 -> 0 ; 10.5+10 = 20.5 -> 20
    assert m[0:21, 10:31].max() == 0 and m[21, 20] == 255 and m[10, 31] == 255 and m[10, 9] == 0
    assert m[0:11, 0:11].max() == 0            # clipped square around (0,0): [-10,10] -> [0,10]
    assert m[37:48, 53:64].max() == 0
    assert m[30, 40] == 255


def _greedy_python(eig, mask, max_corners, quality, min_dist):
    h, w = eig.shape
    mx = eig[mask > 0].max() if (mask > 0).any() else 0.0
    thr = np.float32(float(mx) * quality)
    cand = []
    for y in range(1, h - 1):
        for x in range(1, w - 1):
            v = eig[y, x]
            if not (v > thr) or v == 0 or not mask[y, x]:
                continue
            nb = eig[y - 1:y + 2, x - 1:x + 2]
            if (np.where(nb > thr, nb, 0) > v).any():
                continue
            cand.append((float(v), y * w + x))
    cand.sort(key=lambda t: (-t[0], -t[1]))
    acc = []
    for v, idx in cand:
        y, x = divmod(idx, w)
        if min_dist >= 1 and any((x - ax) ** 2 + (y - ay) ** 2 < min_dist * min_dist for ax, ay in acc):
            continue
        acc.append((x, y))
        if len(acc) == max_corners:
            break
    return np.array(acc, np.float32).reshape(-1, 2)


def test_gftt_matches_python_selection(orc):
    rng = np.random.default_rng(3)
    img = cm.textured(rng, 60, 90)
    eig = orc.min_eig_map(img)
    rect = np.array([[30.2, 20.7], [70.5, 40.5]], np.float32)
    mask = orc.gftt_mask(90, 60, rect)
    for (mc, q, md) in ((40, 0.01, 8.0), (500, 0.002, 3.0), (25, 0.05, 12.0), (60, 0.01, 0.0)):
        got = orc.gftt(img, rect, mc, q, md)
        ref = _greedy_python(eig, mask, mc, q, md)
        assert np.array_equal(got, ref), (mc, q, md)


def test_gftt_properties_on_kitti_shaped_frame(orc, svs):
    l0, _ = svs.synth_pair(11, 0)
    c = orc.gftt(l0)
    assert len(c) == 150                                     # textured frame: the cap binds
    assert np.array_equal(c, np.rint(c))                     # integer pixel coordinates
    assert c[:, 0].min() >= 1 and c[:, 0].max() <= 618 and c[:, 1].min() >= 1 and c[:, 1].max() <= 186
    d = np.linalg.norm(c[:, None] - c[None], axis=2) + 1e9 * np.eye(len(c))
    assert d.min() >= 20.0
    e = orc.min_eig_map(l0)
    q = e[c[:, 1].astype(int), c[:, 0].astype(int)]
    assert np.all(np.diff(q) <= 0)                           # quality-descending
    # masked detection never returns a corner inside an exclusion square
    c2 = orc.gftt(l0, c[:50])
    m = orc.gftt_mask(620, 188, c[:50])
    assert m[c2[:, 1].astype(int), c2[:, 0].astype(int)].min() == 255
    # flat image: nothing
    assert len(orc.gftt(np.full((188, 620), 50, np.uint8))) == 0
