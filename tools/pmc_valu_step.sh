# VALU wave-instructions per unit of work for EVERY kernel family of the step (round 5): one rocprofv3 PMC pass
# (SQ_INSTS_VALU, SQ_INSTS_SALU; --kernel-trace only) over a small bench run, per-kernel totals divided by the unit counts
# bench.py reports for the whole process (units_whole_process) -> gpurun_out/pmc_valu_step.json.  bench.py prices the whole
# step with these constants (roofline_valu_step): instructions issued per second against the chip's VALU issue peak.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/pmc_valu_step; rm -rf $O; mkdir -p $O
timeout ${PMC_TIMEOUT:-900} rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d "$O" -- python bench.py ${PMC_BENCH_ARGS:---streams 1024 --groups 2 --steps 20 --warmup 5} --no-cpu-baseline --spread-windows 0 --super-windows 0 --host-input-steps 0 --solo-steps 0 --predecimated-streams 0 > gpurun_out/pmc_valu_step_bench.json 2> gpurun_out/pmc_valu_step.err < /dev/null
python - <<'PY'
import csv, glob, json, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob("gpurun_out/pmc_valu_step/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_VALU": calls[name] += 1
line = json.loads(open("gpurun_out/pmc_valu_step_bench.json").read().strip().splitlines()[-1])
u = line["units_whole_process"]
fam = {"local_ba": (["k_local_ba_t<0", "k_ba_build", "k_dmap_ba_gather", "k_dmap_ba_scatter"], "job", u["ba_calls"]),
       "lk": (["k_lk"], "point", u["track_pts"] + u["right_pts"]),
       "pose_only": (["k_pose_only<"], "job", u["frames"]),
       "pyramid": (["k_pyr_fused<"], "image", u["pyr_left"] + u["pyr_right"]),
       "gftt": (["k_gftt_eig3<", "k_gftt_select2"], "image", u["gftt_calls"]),
       "triangulate": (["k_triangulate"], "point", u["tri_pts"]),
       "map": (["k_dmap_begin", "k_dmap_commit", "k_dmap_refresh", "k_dmap_stereo"], "keyframe", u["keyframes"])}
out = {"_comment": "SQ_INSTS_VALU / SQ_INSTS_SALU per unit of work, whole process of a small bench run (tools/pmc_valu_step.sh); kernels matched by name prefix",
       "kernels": {k: {"launches": calls[k], **{c: v for c, v in d.items()}} for k, d in sorted(acc.items())}, "units_whole_process": u, "per_unit": {}}
for f, (ks, unit, n) in fam.items():
    tot = lambda c: sum(d.get(c, 0.0) for k, d in acc.items() if any(k.startswith(p) for p in ks))
    out["per_unit"][f] = {"unit": unit, "units_in_run": n, "valu_insts": round(tot("SQ_INSTS_VALU") / max(n, 1), 1), "salu_insts": round(tot("SQ_INSTS_SALU") / max(n, 1), 1)}
json.dump(out, open("gpurun_out/pmc_valu_step_raw.json", "w"), indent=1)
# the layout bench.py reads from profiles/pmc_valu_step.json (per-family constants at top level), stamped with the measured build
pub = {"_comment": out["_comment"] + "; valu_insts = SQ_INSTS_VALU wave-instructions per unit", "build_info": line.get("library"),
       "operating_point": {k: line["config"].get(k) for k in ("streams_per_gpu", "host_threads_per_gpu", "frame", "frame_ring")}}
for f, d in out["per_unit"].items():
    pub[f] = {"unit": d["unit"], "valu_insts": d["valu_insts"], "salu_insts": d["salu_insts"], "units_in_run": d["units_in_run"]}
json.dump(pub, open("gpurun_out/pmc_valu_step.json", "w"), indent=1)
print(json.dumps(out["per_unit"], indent=1))
PY
rm -rf $O
