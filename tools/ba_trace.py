"""LM trajectory of k_local_ba on the local-BA problem captured from the pipeline (tools/ba_pipeline_problem.npz).  Development tool."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common as cm
svs = importlib.import_module("stereovision-slam_amd")
d = np.load(os.path.join(ROOT, "tools", "ba_pipeline_problem.npz"))
c = svs.Context(cm.W, cm.H, max_slots=1, max_jobs=4, max_kf=11, max_lm=4096, max_obs=16384)
c.lm_trace(True)
c.local_ba([(d["poses"], d["pts"], d["okf"], d["olm"], d["ori"], d["uv"])], cm.CAM, cm.EXT_L, cm.CAM, cm.EXT_R)
tr = c.lm_trace(job=0)
for r in tr: print("it %2d lambda %.3e chi %.9f -> %.9f rho %.3e %s" % (r[0], r[1], r[2], r[3], r[4], "ok" if r[5] else "REJ"))
