#!/usr/bin/env python3
"""Development: fingerprints (sha256) of the local-BA outputs — poses, positions, per-edge chi2, iterations — of the captured
pipeline problem and a few seeded problems, on the batch solver and on the low-latency solver.  For changes that must not move
a bit (tools/ab.sh "old base" -- python tools/dev/ba_bits.py; SVSLAM_BA_GENERIC_EXT=1 selects the general-extrinsic code)."""
import hashlib, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common as cm
svs = importlib.import_module("stereovision-slam_amd")


def jobs():
    d = np.load(os.path.join(ROOT, "tools", "ba_pipeline_problem.npz"))
    out = [(d["poses"], d["pts"], d["okf"], d["olm"], d["ori"], d["uv"])]
    rng = np.random.default_rng(11)
    for nkf, nlm in ((10, 700), (7, 300), (4, 120)):
        p = cm.make_ba_problem(rng, nkf, nlm)
        keep = rng.random(len(p["okf"])) < 0.35
        o = np.lexsort((p["okf"][keep], p["olm"][keep]))
        out.append((p["poses0"], p["pts0"], p["okf"][keep][o], p["olm"][keep][o], p["ori"][keep][o], p["ouv"][keep][o]))
    return out


def fp(res):
    h = hashlib.sha256()
    for r in res:
        for a in r[:3]:
            h.update(np.ascontiguousarray(a, np.float64).tobytes())
        h.update(bytes([int(r[3]) & 255]))
    return h.hexdigest()[:16]


J = jobs()
for name, ll in (("batch", 0), ("low-latency", 1)):
    c = svs.Context(cm.W, cm.H, max_slots=1, max_jobs=8, max_kf=11, max_lm=4096, max_obs=16384)
    if ll:
        c.low_latency(True)
    res = c.local_ba(J, cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)
    one = [c.local_ba([j], cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)[0] for j in J]
    print("%-12s all-in-one-call %s  one-by-one %s  iterations %s" % (name, fp(res), fp(one), [int(r[3]) for r in res]))
    c.close()
