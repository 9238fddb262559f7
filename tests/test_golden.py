"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the
oracle): the oracle must keep reproducing them (CPU), the HIP path must match them (GPU)."""
import os

import numpy as np
import pytest

import common as cm

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(G, name))


def _beq(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def test_oracle_reproduces_frontend_golden(orc):
    g = _load("frontend_620x188.npz")
    assert np.array_equal(orc.gftt(g["l0"]), g["corners"])
    assert np.array_equal(orc.gftt(g["l1"], g["rect"]), g["corners_masked"])
    q, st, err = orc.lk(g["l0"], g["r0"], g["corners"], g["corners"])
    assert _beq(q, g["q_r"]) and np.array_equal(st, g["st_r"]) and _beq(err, g["err_r"])
    q, st, err = orc.lk(g["l0"], g["l1"], g["corners"], g["guess"])
    assert _beq(q, g["q_t"]) and np.array_equal(st, g["st_t"]) and _beq(err, g["err_t"])
    p = orc.pyramid(g["l0"])
    assert np.array_equal(p[1], g["pyr1"]) and np.array_equal(p[2], g["pyr2"]) and np.array_equal(p[3], g["pyr3"])
    s = _load("frontend_97x53.npz")
    assert np.array_equal(orc.gftt(s["img"], None, 60, 0.01, 6.0), s["corners"])
    q, st, err = orc.lk(s["img"], s["img2"], s["corners"], s["corners"])
    assert _beq(q, s["q"]) and np.array_equal(st, s["st"])
    assert _beq(orc.min_eig_map(s["img"]), s["eig"])


def test_oracle_reproduces_geometry_golden(orc):
    g = _load("geometry.npz")
    xyz, ok = orc.triangulate(cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R, g["uv_l"], g["uv_r"], g["T_wc"], 300.0)
    assert np.array_equal(ok, g["tri_ok"]) and np.allclose(xyz, g["tri_xyz"], rtol=1e-12, atol=1e-12)
    T, outl, ninl = orc.pose_only(cm.CAM, cm.EXT_L, g["po_P"], g["po_uv"])
    assert np.allclose(T, g["po_T"], atol=1e-10) and np.array_equal(outl, g["po_outl"]) and ninl == int(g["po_ninl"][0])
    pa, xa, ca, ia = orc.local_ba(cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R, g["ba_poses0"], g["ba_pts0"], g["ba_okf"],
                                  g["ba_olm"], g["ba_ori"], g["ba_ouv"], jac_mode=0)
    assert ia == int(g["ba_iters"][0])
    assert np.allclose(pa, g["ba_poses"], atol=1e-9) and np.allclose(xa, g["ba_pts"], atol=1e-9)


@pytest.mark.gpu
def test_hip_matches_frontend_golden(svs):
    g = _load("frontend_620x188.npz")
    c = svs.Context(cm.W, cm.H, max_slots=4, max_jobs=4, max_kf=0, max_lm=0, max_obs=0)
    c.pyramid([0, 1, 2], [g["l0"], g["r0"], g["l1"]])
    for lvl, key in ((1, "pyr1"), (2, "pyr2"), (3, "pyr3")):
        assert np.array_equal(c.pyramid_read(0, lvl), g[key])
    e = c.gftt_eigmap(0)
    assert int(np.bitwise_xor.reduce(e.view(np.uint32).ravel())) == int(g["eig_crc"][0])
    a, b = c.gftt([(0, None), (2, g["rect"])])
    assert np.array_equal(a, g["corners"]) and np.array_equal(b, g["corners_masked"])
    (qr, sr, er), (qt, stt, et) = c.lk([(0, 1, g["corners"], g["corners"]), (0, 2, g["corners"], g["guess"])])
    assert _beq(qr, g["q_r"]) and np.array_equal(sr, g["st_r"]) and _beq(er, g["err_r"])
    assert _beq(qt, g["q_t"]) and np.array_equal(stt, g["st_t"]) and _beq(et, g["err_t"])
    c.close()
    s = _load("frontend_97x53.npz")
    c = svs.Context(97, 53, max_slots=2, max_jobs=2, max_corners=60, max_kf=0, max_lm=0, max_obs=0)
    c.pyramid([0, 1], [s["img"], s["img2"]])
    assert np.array_equal(c.pyramid_read(0, 1), s["pyr1"])
    assert _beq(c.gftt_eigmap(0), s["eig"])
    (corners,) = c.gftt([(0, None)], 60, 0.01, 6.0)
    assert np.array_equal(corners, s["corners"])
    (q, st, err), = c.lk([(0, 1, s["corners"], s["corners"])])
    assert _beq(q, s["q"]) and np.array_equal(st, s["st"]) and _beq(err, s["err"])
    c.close()


@pytest.mark.gpu
def test_hip_matches_geometry_golden(svs):
    g = _load("geometry.npz")
    c = svs.Context(cm.W, cm.H, max_slots=1, max_jobs=2, max_kf=6, max_lm=256, max_obs=2048)
    (xyz, ok), = c.triangulate([(g["uv_l"], g["uv_r"], g["T_wc"], 300.0)], cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)
    assert np.array_equal(ok, g["tri_ok"]) and np.allclose(xyz, g["tri_xyz"], rtol=1e-9, atol=1e-9)
    (T, outl, ninl), = c.pose_only([(cm.EXT_L, g["po_P"], g["po_uv"])], cm.CAM)
    assert np.allclose(T[4:], g["po_T"][4:], atol=1e-6) and np.allclose(T[:4], g["po_T"][:4], atol=1e-7)
    assert np.array_equal(outl, g["po_outl"]) and ninl == int(g["po_ninl"][0])
    (pa, xa, ca, ia), = c.local_ba([(g["ba_poses0"], g["ba_pts0"], g["ba_okf"], g["ba_olm"], g["ba_ori"], g["ba_ouv"])],
                                   cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)
    assert ia == int(g["ba_iters"][0])
    assert np.allclose(pa[:, 4:], g["ba_poses"][:, 4:], atol=1e-6) and np.allclose(pa[:, :4], g["ba_poses"][:, :4], atol=1e-7)
    assert np.allclose(xa, g["ba_pts"], rtol=1e-6, atol=1e-6)
    c.close()
