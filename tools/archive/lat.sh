# single-stream / few-stream latency: frames/s and per-family kernel time at S = 1, 8, 64;
# BA inside the frame (backend mode 1) or beside the next frame (mode 2, the reference's backend
# thread); default kernel shapes or --low-latency
for s in 1 8 64; do
for m in 1 2; do
for ll in "" "--low-latency"; do
python bench.py --streams $s --groups 1 --host-threads 1 --steps 300 --warmup 20 --no-cpu-baseline --backend-mode $m $ll 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); h=d['host_ms_per_step']; k=d['kernel_ms']
print('S=$s backend_mode=$m $ll fps %.0f ms/step %.3f  in_abi %.3f wait %.3f  kernel ms/step: ' % (d['value'], d['ms_per_step'], h['in_abi_calls'], h['stream_wait']) + ', '.join('%s %.3f' % (a, b/d['steps']) for a,b in k.items()), ' kf', d['config']['keyframes_in_timed_region'])"
done; done; done
