import sys, time, importlib
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import common
svs = importlib.import_module("stereovision-slam_amd")
W, H = 620, 188
rng = np.random.default_rng(5)
pr = common.make_ba_problem(rng, nkf=8, nlm=400)
o = np.lexsort((pr["okf"], pr["olm"]))
job = (pr["poses0"], pr["pts0"], pr["okf"][o], pr["olm"][o], pr["ori"][o], pr["ouv"][o])
args = (common.CAM, common.EXT_L, common.CAM, common.EXT_R, 5.991, 10)
mk = lambda: svs.Context(W, H, max_slots=1, max_jobs=2, max_kf=11, max_lm=2048, max_obs=16384)
ll = mk(); ll.low_latency(True)
print("limits", ll.ll_limits())
ll.local_ba([job], *args); ll.host_counters()
holder = mk()
t0 = time.perf_counter(); holder.hold_cus(250, 30.0); t1 = time.perf_counter(); holder.sync(); t2 = time.perf_counter()
print("hold: enqueue %.3f ms, until done %.3f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t0)))
for held in (256, 250, 240, 200, 128):
    holder.hold_cus(held, 30.0); time.sleep(0.004)
    t0 = time.perf_counter(); ll.local_ba([job], *args); dt = time.perf_counter() - t0
    hc = ll.host_counters(); holder.sync()
    print("held %d: BA call %.3f ms, ll problems %d fallbacks %d" % (held, 1e3 * dt, hc[6], hc[7]))
