# round 3: few-stream latency.  device-resident map (backend inside the frame) vs the reference's backend-thread shape
# (backend mode 2) with the BA result landing 1 or 6 frames late; low-latency kernel shapes
for s in 1 8 64; do
for v in "--backend-mode 1" "--backend-mode 1 --host-map" "--backend-mode 2 --backend-lag 1" "--backend-mode 2 --backend-lag 6"; do
python bench.py --streams $s --groups 1 --host-threads 1 --steps 300 --warmup 20 --no-cpu-baseline --spread-windows 0 --host-input-steps 0 --solo-steps 0 --low-latency $v 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); h=d['host_ms_per_step']; k=d['kernel_ms']
print('S=$s $v --low-latency: fps %.0f ms/step %.3f  in_abi %.3f  kernel ms/step: ' % (d['value'], d['ms_per_step'], h['in_abi_calls']) + ', '.join('%s %.3f' % (a, b/d['steps']) for a,b in k.items()), ' kf', d['config']['keyframes_in_timed_region'], 'ate', d['config']['checks'])"
done; done
