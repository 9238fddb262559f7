#!/bin/bash
# round 3: refresh the evidence under gpurun_out/prof3 (what is to be judged is copied into profiles/ afterwards):
#   the default bench line, the driver's 20/5 line, the same command under rocprofv3 --kernel-trace --stats,
#   full resolution, kernel benches (BA alone / 256 / throughput, LK, pose-only), few-stream latency,
#   the PMC traffic passes at the headline operating point
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/prof3; mkdir -p $O
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err < /dev/null ) 2> $O/bench_default.time
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err < /dev/null
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp -o rp -- python bench.py --no-cpu-baseline --spread-windows 0 --host-input-steps 0 > $O/bench_under_rocprof.json 2> $O/rocprof.err < /dev/null
f=$(find $O/rp -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv
rm -rf $O/rp
timeout 600 python bench.py --full-res --steps 20 --warmup 5 --no-cpu-baseline --spread-windows 2 --host-input-steps 0 --solo-steps 0 > $O/bench_full_res.json 2> $O/bench_full_res.err < /dev/null
timeout 600 python bench.py --host-map --steps 20 --warmup 5 --no-cpu-baseline --spread-windows 2 --host-input-steps 0 --solo-steps 0 > $O/bench_host_map.json 2> $O/bench_host_map.err < /dev/null
timeout 200 python tools/kbench.py ba1 > $O/kbench_ba1.txt 2>&1 < /dev/null
timeout 200 python tools/kbench.py ba2 > $O/kbench_ba2.txt 2>&1 < /dev/null
timeout 200 python tools/kbench.py lk > $O/kbench_lk.txt 2>&1 < /dev/null
timeout 120 python tools/kbench.py gftt > $O/kbench_gftt.txt 2>&1 < /dev/null
timeout 120 python tools/po_trace.py > $O/po_trace.txt 2>&1 < /dev/null
timeout 900 bash tools/lat3.sh > $O/latency_small_S.txt 2>&1 < /dev/null
PMC_TIMEOUT=900 PMC_BENCH_ARGS="--steps 20 --warmup 5" timeout 1900 bash tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1 < /dev/null
cp gpurun_out/pmc_traffic_raw.json $O/ 2>/dev/null
cat $O/bench_default.time | tail -3; tail -c 600 $O/bench_default.json; echo; head -14 $O/kernel_stats.csv | cut -c1-110; tail -2 $O/pmc_traffic.log | cut -c1-900
