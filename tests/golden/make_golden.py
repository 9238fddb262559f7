#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the CPU oracle.

The reference ships no golden vectors and cannot be built or imported here (SURVEY
§8c), so these fixtures pin the *oracle's* outputs on seeded inputs: the CPU tests
check that the oracle still reproduces them (guards against silent drift of the
checker), the GPU tests check the HIP path against them.  Re-run only when the
declared algorithm changes:   python tests/golden/make_golden.py
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import common as cm
import oracle_lib as orc

svs = importlib.import_module("stereovision-slam_amd")


def frontend():
    rng = np.random.default_rng(2024)
    l0, r0 = svs.synth_pair(42, 0)
    l1, _ = svs.synth_pair(42, 1)
    corners = orc.gftt(l0)
    rect = corners[:70] + rng.normal(0, 1.5, (70, 2)).astype(np.float32)
    corners_masked = orc.gftt(l1, rect)
    q_r, st_r, err_r = orc.lk(l0, r0, corners, corners)
    guess = corners + rng.normal(0, 2.0, corners.shape).astype(np.float32)
    q_t, st_t, err_t = orc.lk(l0, l1, corners, guess)
    pyr = orc.pyramid(l0)
    np.savez_compressed(os.path.join(HERE, "frontend_620x188.npz"), l0=l0, r0=r0, l1=l1, corners=corners, rect=rect,
                        corners_masked=corners_masked, q_r=q_r, st_r=st_r, err_r=err_r, guess=guess, q_t=q_t,
                        st_t=st_t, err_t=err_t, pyr1=pyr[1], pyr2=pyr[2], pyr3=pyr[3],
                        eig_crc=np.array([np.bitwise_xor.reduce(orc.min_eig_map(l0).view(np.uint32).ravel())]))
    # small odd-sized image: edge cases of reflect borders / partial tiles
    img = cm.textured(rng, 53, 97)
    img2 = np.roll(img, (1, 2), (0, 1))
    c = orc.gftt(img, None, 60, 0.01, 6.0)
    q, st, err = orc.lk(img, img2, c, c)
    np.savez_compressed(os.path.join(HERE, "frontend_97x53.npz"), img=img, img2=img2, corners=c, q=q, st=st, err=err,
                        eig=orc.min_eig_map(img), pyr1=orc.pyramid(img)[1])


def geometry():
    rng = np.random.default_rng(77)
    l0, r0 = svs.synth_pair(42, 0)
    corners = orc.gftt(l0)
    q, st, _ = orc.lk(l0, r0, corners, corners)
    m = st > 0
    T_wc = cm.random_pose(rng)
    xyz, ok = orc.triangulate(cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R, corners[m], q[m], T_wc, 300.0)
    n = 200
    P = np.stack([rng.uniform(-8, 8, n), rng.uniform(-3, 1.5, n), rng.uniform(5, 50, n)], 1)
    T_true = cm.random_pose(rng, 0.6, 0.03)
    uv, _ = cm.project(cm.CAM, T_true, cm.EXT_L, P)
    uv += rng.normal(0, 0.5, uv.shape)
    bad = rng.random(n) < 0.1
    uv[bad] += rng.normal(0, 30, (int(bad.sum()), 2))
    uv = uv.astype(np.float32)
    T_po, outl, ninl = orc.pose_only(cm.CAM, cm.EXT_L, P, uv)
    ba = cm.make_ba_problem(rng, 5, 120)
    pa, xa, ca, ia = orc.local_ba(cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R, ba["poses0"], ba["pts0"], ba["okf"], ba["olm"],
                                  ba["ori"], ba["ouv"], jac_mode=0)
    np.savez_compressed(os.path.join(HERE, "geometry.npz"), uv_l=corners[m], uv_r=q[m], T_wc=T_wc, tri_xyz=xyz, tri_ok=ok,
                        po_P=P, po_uv=uv, po_T=T_po, po_outl=outl, po_ninl=np.array([ninl]),
                        ba_poses0=ba["poses0"], ba_pts0=ba["pts0"], ba_okf=ba["okf"], ba_olm=ba["olm"], ba_ori=ba["ori"],
                        ba_ouv=ba["ouv"], ba_poses=pa, ba_pts=xa, ba_chi2=ca, ba_iters=np.array([ia]))


if __name__ == "__main__":
    frontend()
    geometry()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
