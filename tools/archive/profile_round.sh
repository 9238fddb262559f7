#!/bin/bash
# refresh the per-round evidence under gpurun_out/prof (copy what is to be judged into profiles/):
#   default bench line, the driver's 20/5 line, the same command under rocprofv3 --kernel-trace --stats,
#   the full-resolution bench line, micro-benchmarks, the ATE distribution, the PMC traffic passes
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/prof; mkdir -p $O
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err < /dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err < /dev/null
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp -o rp -- python bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err < /dev/null
f=$(find $O/rp -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv
timeout 300 python tools/trace_busy.py $O/rp > $O/trace_busy.txt 2>&1 < /dev/null
rm -rf $O/rp
timeout 600 python bench.py --full-res --steps 60 --no-cpu-baseline > $O/bench_full_res.json 2> $O/bench_full_res.err < /dev/null
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_100_10.json 2> $O/bench_100_10.err < /dev/null
timeout 120 python tools/kbench.py gftt > $O/kbench_gftt.txt 2>&1 < /dev/null
SVSLAM_TIMING_SPLIT=1 timeout 120 python tools/kbench.py gftt > $O/kbench_gftt_split.txt 2>&1 < /dev/null
timeout 200 python tools/kbench.py lk > $O/kbench_lk.txt 2>&1 < /dev/null
timeout 300 python tools/kbench.py tput > $O/kbench_tput.txt 2>&1 < /dev/null
timeout 900 python tests/ate_distribution.py 384 320 --analytic-too > $O/ate_distribution.txt 2> $O/ate_distribution.err < /dev/null
timeout 900 bash tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1 < /dev/null
cp gpurun_out/pmc_traffic_raw.json $O/ 2>/dev/null
timeout 600 bash tools/lat.sh > $O/latency_small_S.txt 2>&1 < /dev/null
tail -c 400 $O/bench_default.json; echo; head -12 $O/kernel_stats.csv | cut -c1-120; head -8 $O/ate_distribution.txt; tail -3 $O/pmc_traffic.log | cut -c1-600
