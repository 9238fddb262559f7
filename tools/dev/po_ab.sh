#!/bin/bash
# development: pose-only variants — the batch shape (tools/po_trace.py: 2048 / 4096 jobs) and a lone camera's frame (four-wave shape)
timeout 100 python tools/po_trace.py 2>&1 | tail -2
for v in "--backend-mode 1" "--backend-mode 2 --backend-lag 6"; do
python bench.py --streams 1 --groups 1 --host-threads 1 --steps 300 --warmup 20 --no-cpu-baseline --spread-windows 3 --host-input-steps 0 --solo-steps 0 --full-res-streams 0 --low-latency $v 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; sp=d.get('value_spread') or {}
print('S=1 $v: fps %.0f (ms/step %.3f); without events %.0f  kernel ms/step: ' % (d['value'], d['ms_per_step'], sp.get('mean', 0)) + ', '.join('%s %.3f' % (a, b/d['steps']) for a,b in k.items()))"
done
