export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/pmc_lds; rm -rf $O; mkdir -p $O
cat > /tmp/ba256.py <<'PY'
import sys; sys.argv=["kbench","none"]
sys.path.insert(0,"tools"); import kbench
kbench.ba(256, 0, 0, reps=2)
PY
timeout 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/p -- python /tmp/ba256.py > $O/p.log 2>&1
f=$(find $O/p -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: [0,0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if "k_local_ba" in r["Kernel_Name"]:
        a=acc[r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
for k,(n,v) in sorted(acc.items()): print("%-32s per-launch %.4g  (n=%d)"%(k, v/n, n))
PY
rm -rf $O/p
