# Where k_lk's VALU instructions go: SQ_INSTS_VALU per point at iteration caps 1, 2, 4, 8, 30 (tools/kbench.py lk,
# 512 jobs x 150 points, temporal pair).  With cap 1 every (point, level) runs its set-up and ONE iteration; the differences
# between caps, divided by the extra iterations the histogram says were run, give the cost of an iteration.
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/pmc_lk_split; rm -rf $O; mkdir -p $O
for mc in 1 2 4 8 30; do
  timeout 180 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/p$mc -- env LKBENCH_MC=$mc python tools/kbench.py lk > $O/p$mc.log 2>&1
  f=$(find $O/p$mc -name "*counter_collection.csv" | head -1)
  python - "$f" $mc <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if "k_lk" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
except Exception as e: print("no counters:", e)
# launches alternate temporal (first half of the process) / stereo (second half): report both halves
for k,v in sorted(acc.items()):
    h=len(v)//2
    print("max_count %2s  %-14s temporal per-launch %.5g  stereo per-launch %.5g  (n=%d)" % (sys.argv[2], k, sum(v[:h])/max(h,1), sum(v[h:])/max(len(v)-h,1), len(v)))
PY
  rm -rf $O/p$mc
done
