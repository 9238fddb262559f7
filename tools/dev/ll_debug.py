import os, sys, importlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common as cm
import oracle_lib as orc
svs = importlib.import_module("stereovision-slam_amd")

def sj(job):
    poses, pts, okf, olm, ori, ouv = job
    o = np.lexsort((okf, olm))
    return poses, pts, okf[o], olm[o], ori[o], ouv[o]

def mk(w):
    os.environ["SVSLAM_LL_SHARDS"] = str(w)
    c = svs.Context(cm.W, cm.H, max_slots=1, max_jobs=16, max_kf=11, max_lm=4096, max_obs=16384)
    c.low_latency(True); c.lm_trace(True)
    return c
cb = svs.Context(cm.W, cm.H, max_slots=1, max_jobs=16, max_kf=11, max_lm=4096, max_obs=16384); cb.lm_trace(True)
rng = np.random.default_rng(5)
cm.make_ba_problem(rng, 4, 7, outlier_frac=0.0); cm.make_ba_problem(rng, 4, 7, noise=0.1, outlier_frac=0.0, pose_noise=0.002, pt_noise=0.005)
p = cm.make_ba_problem(rng, 6, 200)
okf, olm, ori, ouv = p["okf"], p["olm"], p["ori"], p["ouv"]
variants = {"full": np.ones(len(okf), bool), "no kf2": okf != 2, "no lm": ~np.isin(olm, (0, 77, 199)), "both": (okf != 2) & ~np.isin(olm, (0, 77, 199))}
for w in (8, 16, 4):
    c = mk(w)
    for name, keep in variants.items():
        job = sj((p["poses0"], p["pts0"], okf[keep], olm[keep], ori[keep], ouv[keep]))
        (pa, xa, ca, ia), = c.local_ba([job], cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)
        ta = c.lm_trace(job=0)
        sh = c.ll_shards(1)[0]
        (pb, xb, cb_, ib), = cb.local_ba([job], cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)
        tb = cb.lm_trace(job=0)
        pr, xr, cr, itr = orc.local_ba(cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R, *job)
        print("W=%d %-7s it %d/%d/%d  LL-batch dpose %.2e dpts %.2e | batch-oracle %.2e %.2e | LL-oracle %.2e %.2e | tiles %s" % (
            w, name, ia, ib, itr, np.abs(pa - pb).max(), np.abs(xa - xb).max(), np.abs(pb - pr).max(), np.abs(xb - xr).max(),
            np.abs(pa - pr).max(), np.abs(xa - xr).max(), sh[:, 3].tolist()))
        n = min(len(ta), len(tb))
        d = np.abs(ta[:n, 1:4] / tb[:n, 1:4] - 1).max(axis=1)
        print("   trace rel diff per trial:", " ".join("%.1e" % v for v in d), " accepted", ta[:n, 5].astype(int).tolist())
    c.close()
