#!/bin/bash
# round 6: the evidence run -> gpurun_out/prof6 (what is to be judged is copied into profiles/r6_* afterwards).
#   bench       the default command (= the driver's: 20 + 5 steps, 8192 streams, 1241x376 frames in HBM), the launcher at N = 1
#   rocprof     the default command under rocprofv3 --kernel-trace --stats
#   variants    stream count / ring sweeps of the full-resolution configuration, backend mode 2, host map
#   kbench      kernel benches (batch BA, low-latency BA, LK, GFTT, pose-only)
#   coop        cooperative launch of the low-latency BA solver, A/B
#   latency     few-stream latency table (pre-decimated frames like rounds 4-5, and 1241x376 frames)
#   pmc_valu    PMC passes over k_local_ba (256 problems) and k_lk (512 x 150) -> pmc_valu.json
#   pmc_step    SQ_INSTS_VALU per unit for every family at the headline configuration -> pmc_valu_step.json
#   pmc_traffic FETCH_SIZE / WRITE_SIZE passes at the headline configuration -> pmc_traffic.json
# usage: tools/profile_round6.sh [part ...]
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/prof6; mkdir -p $O
parts="${*:-bench rocprof kbench latency pmc_valu pmc_step pmc_traffic}"
has() { case " $parts " in *" $1 "*) return 0;; esac; return 1; }
QUIET="--no-cpu-baseline --host-input-steps 0 --solo-steps 0 --predecimated-streams 0"
if has bench; then
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err < /dev/null ) 2> $O/bench_default.time
timeout 900 bash tools/scale.sh 1 --steps 20 --warmup 5 > $O/scale_n1.json 2> $O/scale_n1.err < /dev/null
fi
if has rocprof; then
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp -o rp -- python bench.py --no-cpu-baseline --spread-windows 0 --super-windows 0 --host-input-steps 0 --predecimated-streams 0 > $O/bench_under_rocprof.json 2> $O/rocprof.err < /dev/null
f=$(find $O/rp -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv
rm -rf $O/rp
fi
if has variants; then
VARS=("--streams 8192 --groups 3" "--streams 8192 --groups 4" "--streams 8192 --groups 5" "--streams 8192 --groups 6" "--streams 8192 --groups 8" "--streams 12288 --ring-frames 16 --groups 4" "--streams 12288 --ring-frames 16 --groups 7" "--streams 10240 --ring-frames 20 --groups 6" "--streams 6144 --groups 4" "--backend-mode 2" "--host-map" "--pre-decimated --streams 12288")
for v in "${VARS[@]}"; do
  echo "== $v"; timeout 600 python bench.py $v $QUIET --spread-windows 3 --super-windows 0 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); w = d['value_windows']; k = d['kernel_ms']
print('value %.0f (min %.0f max %.0f) ms/step %.3f S %d G %d ring %s | kernel-ms ' % (d['value'], w['min'], w['max'], d['ms_per_step'], d['config']['streams_per_gpu'], d['config']['host_threads_per_gpu'], d['config']['frame_ring']) + ', '.join('%s %.0f' % (a, b) for a, b in k.items()) + ' | roofline %s frac %.4f' % (d['roofline']['interval'], d['roofline']['frac']))"
done > $O/variants.txt 2>&1
fi
if has kbench; then
timeout 200 python tools/kbench.py ba1 > $O/kbench_ba1.txt 2>&1 < /dev/null
timeout 300 python tools/kbench.py ball > $O/kbench_ball.txt 2>&1 < /dev/null
timeout 200 python tools/kbench.py lk > $O/kbench_lk.txt 2>&1 < /dev/null
timeout 120 python tools/kbench.py gftt > $O/kbench_gftt.txt 2>&1 < /dev/null
timeout 120 python tools/po_trace.py > $O/po_trace.txt 2>&1 < /dev/null
fi
if has latency; then
( for fr in "--pre-decimated" ""; do for s in 1 8 64; do
for v in "--backend-mode 1" "--backend-mode 2 --backend-lag 1" "--backend-mode 2 --backend-lag 6"; do
python bench.py $fr --streams $s --groups 1 --host-threads 1 --steps 300 --warmup 20 --ring-frames 160 $QUIET --spread-windows 3 --super-windows 0 --low-latency $v 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); h=d['host_ms_per_step']; k=d['kernel_ms']; sp=d.get('value_spread') or {}
print('S=$s $fr $v --low-latency: fps %.0f (ms/step %.3f) with the per-family HIP events of the measurement; without them %.0f (three further windows %s)  in_abi %.3f  kernel ms/step: ' % (d['value_windows']['first'], d['value_windows']['first_ms_per_step'], sp.get('mean', 0), sp.get('windows'), h['in_abi_calls']) + ', '.join('%s %.3f' % (a, b/d['steps']) for a,b in k.items()), ' kf', d['config']['keyframes_in_timed_region'], 'ate', d['config']['checks'])"
done; done; done ) > $O/latency_small_S.txt 2>&1
fi
if has coop; then
# low-latency BA through hipLaunchCooperativeKernel (SVSLAM_LL_COOP=1) against the ordinary launch: a lone camera's frame rate and the solver alone
( for c in 0 1 0 1; do
SVSLAM_LL_COOP=$c python bench.py --pre-decimated --streams 1 --groups 1 --host-threads 1 --steps 300 --warmup 20 $QUIET --spread-windows 3 --super-windows 0 --low-latency --backend-mode 1 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); sp=d.get('value_spread') or {}; k=d['kernel_ms']
print('SVSLAM_LL_COOP=$c S=1 mode 1: %.0f frames/s with HIP events, %.0f without (%s); local_ba %.3f ms/step, ba_solve %.3f' % (d['value_windows']['first'], sp.get('mean', 0), sp.get('windows'), k['local_ba']/d['steps'], k['ba_solve']/d['steps']))"
done
for c in 0 1; do echo "== SVSLAM_LL_COOP=$c kbench ball"; SVSLAM_LL_COOP=$c timeout 300 python tools/kbench.py ball 2>&1 | tail -6; done ) > $O/coop_ab.txt 2>&1
fi
if has pmc_valu; then
timeout 900 bash tools/pmc_ba.sh > $O/pmc_ba.log 2>&1 < /dev/null; cp gpurun_out/pmc_ba/summary.txt $O/pmc_local_ba_256problems.txt 2>/dev/null
timeout 900 bash tools/pmc_lk.sh > $O/pmc_lk_512x150.txt 2>&1 < /dev/null
python tools/pmc_publish.py $O/pmc_local_ba_256problems.txt $O/pmc_lk_512x150.txt "profiles/r6_pmc_local_ba_256problems.txt, profiles/r6_pmc_lk_512x150.txt" > $O/pmc_valu.json 2> $O/pmc_publish.err
fi
if has pmc_step; then
PMC_TIMEOUT=900 PMC_BENCH_ARGS="--streams 2048 --groups 2 --steps 20 --warmup 5" timeout 1000 bash tools/pmc_valu_step.sh > $O/pmc_valu_step.log 2>&1 < /dev/null
cp gpurun_out/pmc_valu_step.json gpurun_out/pmc_valu_step_raw.json $O/ 2>/dev/null
fi
if has pmc_traffic; then
PMC_TIMEOUT=900 PMC_BENCH_ARGS="--steps 20 --warmup 5" timeout 1900 bash tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1 < /dev/null
cp gpurun_out/pmc_traffic_raw.json gpurun_out/pmc_traffic.json $O/ 2>/dev/null
fi
ls -la $O | head -40
