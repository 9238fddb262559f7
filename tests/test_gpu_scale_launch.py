"""GPU: the multi-GPU launch path on whatever this box has.  tools/scale.sh N is the driver's own launch
(torch.distributed.run, one rank per GPU, RCCL); here N = the number of visible devices — 1 on the one-GPU box, where the
job still forms an RCCL communicator, meets at its barriers and all-reduces over it.  The stream partition itself
(north_star: one stream set per GPU, no data-path collective) is covered on CPU by the world-2 gloo test in
tests/test_abi_and_host.py."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_scale_launch_at_the_visible_device_count(svs):
    n = svs.load().svslam_device_count()
    assert n >= 1
    args = ["--steps", "6", "--warmup", "2", "--streams", "768", "--no-cpu-baseline", "--spread-windows", "0",
            "--host-input-steps", "0", "--solo-steps", "0", "--full-res-streams", "0"]
    env = dict(os.environ, SCALE_PORT="29547")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "scale.sh"), str(n)] + args, capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == n and d["ranks_seen"] == n and d["rank_exchange"].startswith("RCCL")
    assert d["config"]["checks"]["duplicate_stream_bit_identical"] is True
    assert d["config"]["parallelism"].endswith("no collective")
    assert d["steps"] == 6 and d["warmup"] == 2
