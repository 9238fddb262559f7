"""pose-only results of a fixed set of jobs (tracking-shaped, rejecting, > 192 and 512 edges; both kernel shapes) into an .npz: run it
with two builds of the library and compare the files (bit-identity check of a kernel rewrite).  Development tool."""
import importlib, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common as cm, lm_cases as lc
svs = importlib.import_module("stereovision-slam_amd")
c = svs.Context(cm.W, cm.H, max_slots=1, max_jobs=64)
jobs = [lc.po_tracking_case(s) for s in range(12)] + [lc.po_case(s) for s in lc.PO_REJECT]
# a job with more than 192 edges (slots beyond the register ones) and one with > 256
rng = np.random.default_rng(5)
for n in (230, 400, 512):
    P = np.stack([rng.uniform(-8, 8, n), rng.uniform(-2, 1.5, n), rng.uniform(4.0, 40.0, n)], 1)
    T_true = cm.random_pose(rng, 0.9, 0.02)
    uv, _ = cm.project(cm.CAM, T_true, cm.EXT_L, P); uv += rng.normal(0, 0.5, uv.shape)
    b = rng.random(n) < 0.05; uv[b] += rng.normal(0, 30, (int(b.sum()), 2))
    jobs.append((cm.EXT_L.copy(), P, uv.astype(np.float32)))
out = {}
for ll in (0, 1):
    c.low_latency(bool(ll))
    res = c.pose_only(jobs, cm.CAM)
    for i, (T, o, n) in enumerate(res):
        out["T%d_%d" % (ll, i)] = T; out["o%d_%d" % (ll, i)] = o; out["n%d_%d" % (ll, i)] = np.array(n)
np.savez(sys.argv[1], **out)
