#!/bin/bash
# development: the pose-only parameter tolerance (SVSLAM_PO_XTOL) at 0 / 1e-12 / 1e-9: tests, a lone camera's frame, the batch line
mkdir -p gpurun_out/r5
timeout 600 python -m pytest tests/test_gpu_pose_xtol.py -q -s --tb=short 2>&1 | tail -25 | cut -c1-400
for x in 0 1e-12 1e-9; do
echo "== SVSLAM_PO_XTOL=$x"
SVSLAM_PO_XTOL=$x timeout 600 python -m pytest tests/test_gpu_pipeline.py -q -x --tb=line 2>&1 | tail -3 | cut -c1-300
for v in "--backend-mode 1" "--backend-mode 2 --backend-lag 6"; do
SVSLAM_PO_XTOL=$x python bench.py --streams 1 --groups 1 --host-threads 1 --steps 300 --warmup 20 --no-cpu-baseline --spread-windows 3 --host-input-steps 0 --solo-steps 0 --full-res-streams 0 --low-latency $v 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; sp=d.get('value_spread') or {}
print('S=1 $v: fps %.0f (ms/step %.3f); without events %.0f  kernel ms/step: ' % (d['value'], d['ms_per_step'], sp.get('mean', 0)) + ', '.join('%s %.3f' % (a, b/d['steps']) for a,b in k.items()), 'ate', d['config']['checks'])"
done
SVSLAM_PO_XTOL=$x bash tools/dev/bench_line.sh
done
