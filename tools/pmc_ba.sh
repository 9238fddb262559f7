# PMC passes over the BA kernel at scale (256 problems in one launch); summaries -> gpurun_out/pmc_ba
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/pmc_ba; rm -rf $O; mkdir -p $O
cat > /tmp/ba256.py <<'PY'
import sys; sys.argv=["kbench","none"]
sys.path.insert(0,"tools"); import kbench
kbench.ba(256, 0, 0, reps=2)
PY
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 90 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python /tmp/ba256.py > $O/p$i.log 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: [0,0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if "k_local_ba" in r["Kernel_Name"]:
        a=acc[r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
for k,(n,v) in sorted(acc.items()): print("%-32s per-launch %.4g  (n=%d)"%(k, v/n, n))
PY
  else echo "pass $i: no counters"; tail -3 $O/p$i.log; fi
  rm -rf $O/p$i
done 2>&1 | tee $O/summary.txt
