#!/bin/bash
# HBM-side traffic per unit of work, STEADY STATE: two PMC passes (FETCH_SIZE, WRITE_SIZE separately: they do not fit one pass)
# over a bench run with every further leg switched off, reduced over the launches of the run's timed window (tools/pmc_reduce.py)
# -> gpurun_out/pmc_traffic.json in the layout bench.py reads from profiles/pmc_traffic.json, stamped with svslam_build_info() of
# the library that was measured.  PMC_BENCH_ARGS chooses the operating point (round 6: "--steps 20 --warmup 5" = the headline one,
# 8192 streams x 4 groups, 1241x376 frames in HBM).
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for ctr in FETCH_SIZE WRITE_SIZE; do
  O=gpurun_out/pmc_$ctr; rm -rf "$O"; mkdir -p "$O"
  timeout ${PMC_TIMEOUT:-400} rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$O" -- python bench.py ${PMC_BENCH_ARGS:---streams 256 --groups 1 --host-threads 4 --steps 30 --warmup 5 --preroll 100} --no-cpu-baseline --spread-windows 0 --super-windows 0 --host-input-steps 0 --solo-steps 0 --predecimated-streams 0 > gpurun_out/pmc_${ctr}_bench.json 2> gpurun_out/pmc_${ctr}.err < /dev/null
  python tools/pmc_reduce.py "$O" gpurun_out/pmc_${ctr}_bench.json $ctr > gpurun_out/pmc_${ctr}_reduced.json
done
python - <<'PY'
import json
f = json.load(open("gpurun_out/pmc_FETCH_SIZE_reduced.json")); w = json.load(open("gpurun_out/pmc_WRITE_SIZE_reduced.json"))
line = json.loads(open("gpurun_out/pmc_FETCH_SIZE_bench.json").read().strip().splitlines()[-1])
pub = {"_comment": "HBM-side traffic per unit of work in the STEADY STATE from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, "
                   "--kernel-trace only) over one bench.py run: the launches of the run's timed window (its last launches; every further leg off), "
                   "per family, divided by the window's unit counts (units_timed_window; tools/pmc_reduce.py).  KB -> bytes x 1024.  fetch_bytes = 2 x "
                   "the raw counter: on gfx950 FETCH_SIZE reports half of the bytes read for every access width (tools/pmc_calib.sh, "
                   "profiles/r2_pmc_calibration.txt; WRITE_SIZE 1.000 x).  Under a PMC pass rocprofv3 serialises the kernels: every launch is measured "
                   "alone on the chip at the bench's own batch shapes.  Written by tools/pmc_traffic.sh.",
       "build_info": line.get("library"),
       "operating_point": {k: line["config"].get(k) for k in ("streams_per_gpu", "host_threads_per_gpu", "frame", "frame_ring")}}
for fam in f:
    n = max(f[fam]["units"], 1)
    fr = f[fam]["FETCH_SIZE"] * 1024 / n
    wb = w[fam]["WRITE_SIZE"] * 1024 / n
    pub[fam] = {"unit": f[fam]["unit"], "units_in_window": f[fam]["units"], "launches": f[fam]["launches"], "dispatches_used": f[fam]["dispatches_used"],
                "fetch_bytes_raw": round(fr), "fetch_bytes": round(2 * fr), "write_bytes": round(wb), "bytes": round(2 * fr + wb)}
json.dump(pub, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in pub.items() if isinstance(v, dict) and "bytes" in v}, indent=1))
PY
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
