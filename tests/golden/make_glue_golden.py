"""tests/golden/glue_second_reading.npz: per-frame metadata of the reference's host glue as tests/ref_glue.py (the second,
independent reading of /root/reference's Frontend / Map / MapPoint / Backend glue) produces it over the oracle kernels with
analytic BA Jacobians, on two seeded synthetic streams.  Run from the repo root:  python tests/golden/make_glue_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, os.path.join(HERE, "..", ".."))
import common          # noqa: E402
import glue_scenarios as gs   # noqa: E402
import ref_glue        # noqa: E402

SEEDS, N = [7, 8], 60
out = {k: np.zeros((len(SEEDS), N), np.int64) for k in gs.KEYS + ("n_landmarks", "n_observations", "map_crc32")}
out["pose"] = np.zeros((len(SEEDS), N, 7))
out["window"] = -np.ones((len(SEEDS), N, 16), np.int64)
for si, seed in enumerate(SEEDS):
    vo = ref_glue.VisualOdometry(common.CAM, common.BASELINE, {"ba_jac_mode": 0})
    for f, (left, right) in enumerate(gs.frames(seed, "plain", N)):
        r = vo.step(left, right)
        for k in gs.KEYS:
            out[k][si, f] = r[k]
        out["pose"][si, f] = r["pose"]
        kf, nlm, nobs, crc = gs.map_digest(vo.snapshot())
        out["window"][si, f, :len(kf)] = kf
        out["n_landmarks"][si, f], out["n_observations"][si, f], out["map_crc32"][si, f] = nlm, nobs, crc
    print(hex(seed), vo.events, vo.status_seen)
np.savez_compressed(os.path.join(HERE, "glue_second_reading.npz"), seeds=np.array(SEEDS, np.int64), **out)
print("wrote glue_second_reading.npz")
