# PMC passes over k_lk at 512 jobs x 150 points (tools/kbench.py lk): issue / wait breakdown
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/pmc_lk; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAVES" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- env LKBENCH_MC=30 python tools/kbench.py lk > $O/p$i.log 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: [0,0.0])
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if "k_lk" in r["Kernel_Name"]:
            a=acc[r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
except Exception as e: print("no counters:", e)
for k,(n,v) in sorted(acc.items()): print("%-28s per-launch %.5g  (n=%d)"%(k, v/n, n))
PY
  rm -rf $O/p$i
done
