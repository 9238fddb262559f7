#!/bin/bash
# HBM-side traffic per kernel from two PMC passes (FETCH_SIZE, WRITE_SIZE separately: they do not fit one
# pass) over a small bench run (256 streams, 1 group); aggregated per kernel name -> gpurun_out/pmc_traffic_raw.json
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for ctr in FETCH_SIZE WRITE_SIZE; do
  O=gpurun_out/pmc_$ctr; rm -rf "$O"; mkdir -p "$O"
  timeout 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$O" -- python bench.py --streams 256 --groups 1 --host-threads 4 --steps 30 --warmup 5 --preroll 100 --no-cpu-baseline > gpurun_out/pmc_${ctr}_bench.json 2> gpurun_out/pmc_${ctr}.err < /dev/null
done
python - <<'PY'
import csv, glob, json, collections, re
out = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % ctr, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr:
                continue
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            a = acc[name]; a[0] += 1; a[1] += float(r["Counter_Value"])
    out[ctr] = {k: {"launches": n, "total_KB": round(v, 1), "KB_per_launch": round(v / n, 2)} for k, (n, v) in sorted(acc.items())}
    try:
        out[ctr + "_bench"] = json.loads(open("gpurun_out/pmc_%s_bench.json" % ctr).read().strip().splitlines()[-1])["config"]
    except Exception as e:
        out[ctr + "_bench"] = str(e)
json.dump(out, open("gpurun_out/pmc_traffic_raw.json", "w"), indent=1)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    print(ctr, {k: v["KB_per_launch"] for k, v in out[ctr].items()})
PY
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
