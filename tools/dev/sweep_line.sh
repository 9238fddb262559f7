#!/bin/bash
# development: one bench line per argument set, e.g. tools/dev/sweep_line.sh "--groups 3" "--groups 5 --streams 7168"
cd "$(dirname "$0")/../.." || exit 1
for v in "$@"; do
echo "== $v"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --spread-windows 3 --super-windows 0 --host-input-steps 0 --solo-steps 0 --predecimated-streams 0 $v 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms']; w = d['value_windows']
print('value %.0f (windows min %.0f max %.0f) ms/step %.3f S %d G %d  kernel-ms: ' % (d['value'], w['min'], w['max'], d['ms_per_step'], d['config']['streams_per_gpu'], d['config']['host_threads_per_gpu']) + ', '.join('%s %.0f' % (a, b) for a, b in k.items()) + '  kf/step %.0f' % (d['units_timed_window']['keyframes'] / d['steps']))"
done
