"""LM trajectory + timing of k_pose_only on a realistic tracking job (previous pose as the start, frame t -> t+1).  Development tool."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common as cm
svs = importlib.import_module("stereovision-slam_amd")

l0, r0 = svs.synth_pair(3, 0); l1, _ = svs.synth_pair(3, 1)
c = svs.Context(cm.W, cm.H, max_slots=3, max_jobs=4096, max_pts=256, max_corners=256, max_kf=0, max_lm=0, max_obs=0)
c.pyramid([0, 1, 2], [l0, r0, l1])
pts = c.gftt([(0, None)], max_corners=150, min_dist=8.0)[0]
q, st, _ = c.lk([(0, 1, pts, pts)])[0]
xyz, ok = c.triangulate([(pts, q, None, 0.0)], cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)[0]
q1, st1, _ = c.lk([(0, 2, pts, pts)])[0]
m = (st > 0) & (ok > 0) & (st1 > 0)
print("edges", m.sum())
c.lm_trace(True)
if len(sys.argv) > 1:
    c.low_latency(True)
res = c.pose_only([(cm.EXT_L, xyz[m], q1[m])], cm.CAM)
tr = c.lm_trace(True, job=0)
c.lm_trace(False)
prof = tr[tr[:, 0] < 0]; tr = tr[tr[:, 0] >= 0]
if len(prof) == 2:
    a = np.concatenate([prof[0, 1:5], prof[1, 1:5]]) / 100.0
    print("phase us: linearise %.1f, sum32 %.1f, ldlt %.1f, exp+mul %.1f, errors %.1f, rest %.1f, sum1 %.1f, rho/lambda %.1f  (total %.1f)" % (*a, a.sum()))
print("pose", res[0][0], "inliers", res[0][2])
print("trials", len(tr), "accepted", int(tr[:, 5].sum()))
for r in range(4):
    t = tr[(tr[:, 0] >= 16 * r) & (tr[:, 0] < 16 * r + 16)]
    print(" round %d: iterations %d trials %d accepted %d" % (r, len(np.unique(t[:, 0])), len(t), int(t[:, 5].sum())))
    for row in t:
        print("    it %2d lambda %.3e chi %.6f -> %.6f rho %.3e %s" % (int(row[0]) - 16 * r, row[1], row[2], row[3], row[4], "ok" if row[5] else "REJ"))
c.timing(True)
for nj in (1, 256, 1024, 2048, 4096):
    for rep in range(3):
        c.pose_only([(cm.EXT_L, xyz[m], q1[m])] * nj, cm.CAM)
    ms, n, _ = c.timing_get("pose_only")
    print("pose_only jobs=%d: %.1f us/launch" % (nj, 1e3 * ms / max(n, 1)))
    c.timing(True)
# The figures above belong to ONE job and move with its trial count (a last digit of its inputs changes that count).  For a
# figure that compares builds: 64 different tracking-shaped jobs (tests/lm_cases.py: po_tracking_case, numpy only), cycled to
# the launch size, with the trials they ran.
import lm_cases as lc
many = [lc.po_tracking_case(s) for s in range(64)]
c.lm_trace(True)
c.pose_only(many, cm.CAM)
ntr = [len(c.lm_trace(True, job=i)) for i in range(len(many))]
c.lm_trace(False)
print("64 tracking-shaped jobs: trials per job mean %.1f (min %d, max %d), edges per job mean %.0f" % (np.mean(ntr), min(ntr), max(ntr), np.mean([len(j[1]) for j in many])))
c.timing(True)
for nj in (2048, 4096):
    jobs = [many[i % 64] for i in range(nj)]
    for rep in range(3):
        c.pose_only(jobs, cm.CAM)
    ms, n, _ = c.timing_get("pose_only")
    print("pose_only jobs=%d (64 different, cycled): %.1f us/launch" % (nj, 1e3 * ms / max(n, 1)))
    c.timing(True)
c.close()
