for ll in "" "--low-latency"; do
python bench.py --streams 1 --groups 1 --host-threads 1 --steps 300 --warmup 20 --no-cpu-baseline --backend-mode 1 $ll 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); h=d['host_ms_per_step']; k=d['kernel_ms']
print('S=1 $ll fps %.0f ms/step %.3f  kernel ms/step: ' % (d['value'], d['ms_per_step']) + ', '.join('%s %.3f' % (a, b/d['steps']) for a,b in k.items()))"
done
