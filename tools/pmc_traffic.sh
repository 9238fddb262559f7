#!/bin/bash
# HBM-side traffic per kernel from two PMC passes (FETCH_SIZE, WRITE_SIZE separately: they do not fit one
# pass) over a bench run; aggregated per kernel name -> gpurun_out/pmc_traffic_raw.json.
# PMC_BENCH_ARGS chooses the operating point: default the small one of round 2 (256 streams, 1 group); round 3 also
# runs the headline one: PMC_BENCH_ARGS="--steps 20 --warmup 5" (round 6: 8192 streams x 4 groups, 1241x376 frames in HBM).
# Also writes gpurun_out/pmc_traffic.json in the layout bench.py reads from profiles/pmc_traffic.json, stamped with
# svslam_build_info() of the library that was measured (bench.py compares the stamp with the library it loads).
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for ctr in FETCH_SIZE WRITE_SIZE; do
  O=gpurun_out/pmc_$ctr; rm -rf "$O"; mkdir -p "$O"
  timeout ${PMC_TIMEOUT:-400} rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$O" -- python bench.py ${PMC_BENCH_ARGS:---streams 256 --groups 1 --host-threads 4 --steps 30 --warmup 5 --preroll 100} --no-cpu-baseline --spread-windows 0 --super-windows 0 --host-input-steps 0 --solo-steps 0 --predecimated-streams 0 > gpurun_out/pmc_${ctr}_bench.json 2> gpurun_out/pmc_${ctr}.err < /dev/null
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
# per unit of work: a PMC pass sees the whole process, so divide by the unit counts of the whole process
# that bench.py reports (units_whole_process); KB -> bytes (x1024, the counter's documented unit)
try:
    u = json.loads(open("gpurun_out/pmc_FETCH_SIZE_bench.json").read().strip().splitlines()[-1])["units_whole_process"]
    fam = {"local_ba": (["k_local_ba_t<0", "k_ba_build", "k_dmap_ba_gather", "k_dmap_ba_scatter"], "job", u["ba_calls"]),
           "lk": (["k_lk"], "point", u["track_pts"] + u["right_pts"]),
           "pose_only": (["k_pose_only<"], "job", u["frames"]),
           "pyramid": (["k_pyr_fused<"], "image", u["pyr_left"] + u["pyr_right"]),
           "gftt": (["k_gftt_eig3<", "k_gftt_select2"], "image", u["gftt_calls"]),
           "triangulate": (["k_triangulate"], "point", u["tri_pts"])}
    per = {}
    def tot(ctr, prefixes):       # kernels are matched by name prefix (template arguments vary with the build)
        return sum(v.get("total_KB", 0.0) for k, v in out[ctr].items() if any(k.startswith(p_) for p_ in prefixes))
    for f, (ks, unit, n) in fam.items():
        # FETCH_SIZE counts half the bytes on gfx950 for every access width (tools/pmc_calib.sh: 0.500 x for byte,
        # dword, 8- and 16-byte reads of a 1 GiB buffer; WRITE_SIZE is exact): corrected here, raw value kept
        fb_raw = tot("FETCH_SIZE", ks) * 1024 / max(n, 1)
        fb = 2.0 * fb_raw
        wb = tot("WRITE_SIZE", ks) * 1024 / max(n, 1)
        per[f] = {"unit": unit, "units_in_run": n, "kernels": ks, "fetch_bytes_raw": round(fb_raw), "fetch_bytes": round(fb), "write_bytes": round(wb), "bytes": round(fb + wb)}
    out["per_unit"] = per
    out["units_whole_process"] = u
except Exception as e:
    out["per_unit"] = "unavailable: %r" % (e,)
json.dump(out, open("gpurun_out/pmc_traffic_raw.json", "w"), indent=1)
if isinstance(out.get("per_unit"), dict):
    line = json.loads(open("gpurun_out/pmc_FETCH_SIZE_bench.json").read().strip().splitlines()[-1])
    pub = {"_comment": "HBM-side traffic per unit of work from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, "
                       "--kernel-trace only) over one bench.py run, summed per kernel over the whole process and divided by the unit counts "
                       "of the whole process the bench line reports (units_whole_process).  KB -> bytes x 1024.  fetch_bytes = 2 x the raw "
                       "counter: on gfx950 FETCH_SIZE reports half of the bytes read for every access width (tools/pmc_calib.sh, "
                       "profiles/r2_pmc_calibration.txt; WRITE_SIZE 1.000 x).  Under a PMC pass rocprofv3 serialises the kernels: every launch "
                       "is measured alone on the chip at the bench's own batch shapes.  Written by tools/pmc_traffic.sh.",
           "build_info": line.get("library"),
           "operating_point": {k: line["config"].get(k) for k in ("streams_per_gpu", "host_threads_per_gpu", "frame", "frame_ring")}}
    pub.update(out["per_unit"])
    json.dump(pub, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    print(ctr, {k: v["KB_per_launch"] for k, v in out[ctr].items()})
print("per unit", out["per_unit"])
PY
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
