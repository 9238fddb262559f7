#!/bin/bash
# round 4: refresh the evidence under gpurun_out/prof4 (what is to be judged is copied into profiles/ afterwards):
#   the default bench line, the driver's 20/5 line, the default command under rocprofv3 --kernel-trace --stats,
#   the script the driver's scaling run uses at N = 1, kernel benches (batch BA, low-latency BA, LK, GFTT, pose-only),
#   few-stream latency with and without the measurement's own HIP events, PMC passes (VALU side of BA and LK, traffic)
# usage: tools/profile_round4.sh [part ...]   parts: bench rocprof kbench latency pmc_valu pmc_traffic (default: all)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/prof4; mkdir -p $O
parts="${*:-bench rocprof kbench latency pmc_valu pmc_traffic}"
has() { case " $parts " in *" $1 "*) return 0;; esac; return 1; }
if has bench; then
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err < /dev/null ) 2> $O/bench_default.time
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err < /dev/null
timeout 900 bash tools/scale.sh 1 --steps 20 --warmup 5 > $O/scale_n1.json 2> $O/scale_n1.err < /dev/null
timeout 600 python bench.py --backend-mode 2 --steps 20 --warmup 5 --no-cpu-baseline --spread-windows 2 --host-input-steps 0 --solo-steps 0 --full-res-streams 0 > $O/bench_backend_mode2.json 2> $O/bench_backend_mode2.err < /dev/null
timeout 600 python bench.py --host-map --steps 20 --warmup 5 --no-cpu-baseline --spread-windows 2 --host-input-steps 0 --solo-steps 0 --full-res-streams 0 > $O/bench_host_map.json 2> $O/bench_host_map.err < /dev/null
fi
if has rocprof; then
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp -o rp -- python bench.py --no-cpu-baseline --spread-windows 0 --host-input-steps 0 --full-res-streams 0 > $O/bench_under_rocprof.json 2> $O/rocprof.err < /dev/null
f=$(find $O/rp -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv
rm -rf $O/rp
fi
if has kbench; then
timeout 200 python tools/kbench.py ba1 > $O/kbench_ba1.txt 2>&1 < /dev/null
timeout 300 python tools/kbench.py ball > $O/kbench_ball.txt 2>&1 < /dev/null
timeout 200 python tools/kbench.py lk > $O/kbench_lk.txt 2>&1 < /dev/null
timeout 120 python tools/kbench.py gftt > $O/kbench_gftt.txt 2>&1 < /dev/null
timeout 120 python tools/po_trace.py > $O/po_trace.txt 2>&1 < /dev/null
fi
if has latency; then
( for s in 1 8 64; do
for v in "--backend-mode 1" "--backend-mode 1 --host-map" "--backend-mode 2 --backend-lag 1" "--backend-mode 2 --backend-lag 6"; do
python bench.py --streams $s --groups 1 --host-threads 1 --steps 300 --warmup 20 --no-cpu-baseline --spread-windows 3 --host-input-steps 0 --solo-steps 0 --full-res-streams 0 --low-latency $v 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); h=d['host_ms_per_step']; k=d['kernel_ms']; sp=d.get('value_spread') or {}
print('S=$s $v --low-latency: fps %.0f (ms/step %.3f) with the per-family HIP events of the measurement; without them %.0f (three further windows %s)  in_abi %.3f  kernel ms/step: ' % (d['value'], d['ms_per_step'], sp.get('mean', 0), sp.get('windows'), h['in_abi_calls']) + ', '.join('%s %.3f' % (a, b/d['steps']) for a,b in k.items()), ' kf', d['config']['keyframes_in_timed_region'], 'ate', d['config']['checks'])"
done; done ) > $O/latency_small_S.txt 2>&1
fi
if has pmc_valu; then
timeout 900 bash tools/pmc_ba.sh > $O/pmc_ba.log 2>&1 < /dev/null; cp gpurun_out/pmc_ba/summary.txt $O/pmc_local_ba_256problems.txt 2>/dev/null
timeout 900 bash tools/pmc_lk.sh > $O/pmc_lk_512x150.txt 2>&1 < /dev/null
fi
if has pmc_traffic; then
PMC_TIMEOUT=900 PMC_BENCH_ARGS="--steps 20 --warmup 5 --full-res-streams 0" timeout 1900 bash tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1 < /dev/null
cp gpurun_out/pmc_traffic_raw.json $O/ 2>/dev/null
fi
ls -la $O | head -40
