#!/bin/bash
# tools/scale.sh N [bench.py arguments...] — exactly the launch the round driver uses for its scaling runs:
# one process per GPU under torch.distributed.run, RCCL (backend "nccl") over xGMI, streams sharded by rank with no
# data-path collective (stereovision-slam_amd/dist.py).  Prints bench.py's JSON line and checks that the job's
# communicator really spanned N ranks (ranks_seen = an all-reduce of ones) and that n_gpus says so.
# N = 1 goes through the same launcher, so the RCCL path runs on a one-GPU box too (tests/test_gpu_scale_launch.py).
set -e
N=${1:-1}; shift || true
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
LINE=$(python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "${SCALE_PORT:-29541}" \
       bench.py --gpus "$N" "$@" | tail -1)
echo "$LINE"
python - "$N" "$LINE" <<'PY'
import json, sys
n = int(sys.argv[1]); d = json.loads(sys.argv[2])
assert d["n_gpus"] == n, "n_gpus %s != %d" % (d["n_gpus"], n)
assert d["ranks_seen"] == n, "the communicator spanned %s ranks, not %d" % (d["ranks_seen"], n)
assert d["rank_exchange"].startswith("RCCL") or d["rank_exchange"].startswith("gloo"), d["rank_exchange"]
assert d["value"] > 0 and d["scaling"] == "weak"
print("scale.sh: %d rank(s) seen over %s; %.0f frames/s" % (n, d["rank_exchange"].split(":")[0], d["value"]), file=sys.stderr)
# one line per rank: which device it drove, the device's NUMA node, the CPUs the rank may use / pinned itself to, its streams
for r in d.get("ranks", []):
    print("scale.sh: rank %(rank)d device %(device)d numa_node %(numa_node)d cpus_allowed %(cpus_allowed)d cpus_pinned %(cpus_pinned)d "
          "streams %(streams)d from seed 0x%(first_stream_seed)X" % r, file=sys.stderr)
devs = [r["device"] for r in d.get("ranks", [])]
assert len(set(devs)) == len(devs), "two ranks drove the same device: %s" % devs
PY
