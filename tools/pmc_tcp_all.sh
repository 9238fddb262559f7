#!/bin/bash
# round 6: cache-line look-ups of the vector L1 per CU and cycle, every kernel of the path (the figure that exposed the pyramid's fill: the L1
# answers one look-up per cycle).  Two PMC passes (TCP_TOTAL_CACHE_ACCESSES_sum; GRBM_GUI_ACTIVE) over a short bench run, per kernel:
# look-ups / 256 CUs / (GRBM_GUI_ACTIVE / 8 XCDs).  Under a PMC pass every launch runs alone on the chip.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/pmc_tcp; rm -rf $O; mkdir -p $O
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python bench.py --streams 2048 --groups 1 --steps 10 --warmup 3 --preroll 60 --no-cpu-baseline --spread-windows 0 --super-windows 0 --host-input-steps 0 --solo-steps 0 --predecimated-streams 0 > $O/p$i.log 2>&1
done
python - "$O" <<'PY'
import csv,sys,collections,glob,re
acc=collections.defaultdict(lambda: collections.defaultdict(lambda: [0,0.0]))
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","")
        a=acc[name][r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
rows=[]
for k,c in acc.items():
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in c and "GRBM_GUI_ACTIVE" in c:
        n,t=c["TCP_TOTAL_CACHE_ACCESSES_sum"]; n2,g=c["GRBM_GUI_ACTIVE"]
        rd=c.get("SQ_INSTS_VMEM_RD",[1,0])[1]; wr=c.get("SQ_INSTS_VMEM_WR",[1,0])[1]
        rows.append((g, k, n, t/n, g/n2/8, (t/n)/256/(g/n2/8), t/max(rd+wr,1)))
print("%-34s %7s %14s %14s %22s %18s" % ("kernel","launches","look-ups/launch","cycles/launch","look-ups per CU-cycle","per VMEM instr"))
for g,k,n,t,cy,u,pv in sorted(rows, reverse=True): print("%-34s %7d %14.4g %14.4g %22.3f %18.1f" % (k[:34],n,t,cy,u,pv))
PY
rm -rf $O/p1 $O/p2
