#!/bin/bash
# round 6: PMC passes over k_pyr_fused<true> — cache-line look-ups of the vector L1 (TCP_TOTAL_CACHE_ACCESSES), VMEM / VALU instructions,
# wave cycles — for library variants under tools/bin/ab/ (old = five aligned dwords per 16 source bytes, 1024 threads; base = the product).
# A short bench run per pass (2048 streams, one group, 1241x376 frames in HBM); averages over every launch of the kernel in the run
# (the pipeline is bit-identical across the variants, so the launches are the same work).   tools/pmc_pyr.sh "old base"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/pmc_pyr; rm -rf $O; mkdir -p $O
L=stereovision-slam_amd/lib
mkdir -p tools/bin/ab/base && cp $L/*.so tools/bin/ab/base/
for v in ${1:-old base}; do
cp tools/bin/ab/$v/*.so $L/ || exit 1
echo "== variant $v"
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TA_TCP_STATE_READ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
           "GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python bench.py --streams 2048 --groups 1 --steps 10 --warmup 3 --preroll 30 --no-cpu-baseline --spread-windows 0 --super-windows 0 --host-input-steps 0 --solo-steps 0 --predecimated-streams 0 > $O/${v}_p$i.log 2>&1
  python - "$O/p$i" <<'PY'
import csv,sys,collections,glob
acc=collections.defaultdict(lambda: [0,0.0])
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_pyr_fused" in r["Kernel_Name"]:
            a=acc[r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
if not acc: print("  (no counters in this pass)")
for k,(n,v) in sorted(acc.items()): print("  %-36s per-launch %.5g  (n=%d)"%(k, v/n, n))
PY
  rm -rf $O/p$i
done
done
cp tools/bin/ab/base/*.so $L/
