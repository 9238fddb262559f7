#!/bin/bash
# VALU wave-instructions per unit of work for EVERY kernel family of the step, STEADY STATE: one rocprofv3 PMC pass (SQ_INSTS_VALU,
# SQ_INSTS_SALU; --kernel-trace only) over a bench run with every further leg off, reduced over the launches of the run's timed window
# (tools/pmc_reduce.py) -> gpurun_out/pmc_valu_step.json.  bench.py prices the whole step with these constants (roofline_valu_step).
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/pmc_valu_step; rm -rf $O; mkdir -p $O
timeout ${PMC_TIMEOUT:-900} rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d "$O" -- python bench.py ${PMC_BENCH_ARGS:---streams 2048 --groups 2 --steps 20 --warmup 5} --no-cpu-baseline --spread-windows 0 --super-windows 0 --host-input-steps 0 --solo-steps 0 --predecimated-streams 0 > gpurun_out/pmc_valu_step_bench.json 2> gpurun_out/pmc_valu_step.err < /dev/null
python tools/pmc_reduce.py $O gpurun_out/pmc_valu_step_bench.json SQ_INSTS_VALU SQ_INSTS_SALU > gpurun_out/pmc_valu_step_raw.json
python - <<'PY'
import json
r = json.load(open("gpurun_out/pmc_valu_step_raw.json"))
line = json.loads(open("gpurun_out/pmc_valu_step_bench.json").read().strip().splitlines()[-1])
pub = {"_comment": "SQ_INSTS_VALU / SQ_INSTS_SALU wave-instructions per unit of work in the STEADY STATE: the launches of a bench run's timed window (its last "
                   "launches; every further leg off) per family, divided by the window's unit counts (tools/pmc_valu_step.sh, tools/pmc_reduce.py)",
       "build_info": line.get("library"),
       "operating_point": {k: line["config"].get(k) for k in ("streams_per_gpu", "host_threads_per_gpu", "frame", "frame_ring")}}
for fam, d in r.items():
    n = max(d["units"], 1)
    pub[fam] = {"unit": d["unit"], "valu_insts": round(d["SQ_INSTS_VALU"] / n, 1), "salu_insts": round(d["SQ_INSTS_SALU"] / n, 1), "units_in_window": d["units"],
                "launches": d["launches"], "dispatches_used": d["dispatches_used"]}
json.dump(pub, open("gpurun_out/pmc_valu_step.json", "w"), indent=1)
print(json.dumps({k: v for k, v in pub.items() if isinstance(v, dict) and "valu_insts" in v}, indent=1))
PY
rm -rf $O
