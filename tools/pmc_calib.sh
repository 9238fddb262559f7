# FETCH_SIZE / WRITE_SIZE calibration (tools/ubench_fetch.hip): KB counted per kernel for 1 GiB moved
export TMPDIR=/tmp; O=gpurun_out/pmc_calib; rm -rf $O; mkdir -p $O
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/$ctr -- tools/bin/ubench_fetch > $O/$ctr.log 2>&1 < /dev/null
  f=$(find $O/$ctr -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" $ctr <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"]==sys.argv[2]:
        print("%-11s %-60s %12.0f KB  = %.3f x the 1 GiB moved" % (sys.argv[2], r["Kernel_Name"][:60], float(r["Counter_Value"]), float(r["Counter_Value"])*1024/2**30))
PY
  rm -rf $O/$ctr
done
