#!/bin/bash
# development: one short bench run, the fields an A/B needs on one line:  tools/dev/bench_line.sh [bench args]
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --spread-windows 3 --host-input-steps 0 --solo-steps 0 --full-res-streams 0 "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms']; w = d['value_windows']
print('value %.0f (windows min %.0f max %.0f) ms/step %.3f groups %d  kernel-ms: ' % (d['value'], w['min'], w['max'], d['ms_per_step'], d['config']['host_threads_per_gpu']) + ', '.join('%s %.0f' % (a, b) for a, b in k.items()) + '  roofline.frac %.4f compute.frac %.4f' % (d['roofline']['frac'], d['roofline_compute']['local_ba']['frac']))"
