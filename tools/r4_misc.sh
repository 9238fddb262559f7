cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4misc
timeout 1500 python -m pytest tests/test_gpu_device_map.py tests/test_gpu_pipeline.py tests/test_facade_kitti.py -x -q -m gpu > gpurun_out/r4misc/test.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4misc/test.log
tail -25 gpurun_out/r4misc/test.log
for v in "--backend-mode 2 --backend-lag 1" "--backend-mode 2 --backend-lag 6"; do
python bench.py --streams 1 --groups 1 --host-threads 1 --steps 300 --warmup 20 --no-cpu-baseline --spread-windows 0 --host-input-steps 0 --solo-steps 0 --full-res-streams 0 --low-latency $v 2>gpurun_out/r4misc/err.log | tail -1 > gpurun_out/r4misc/s1.json
python - "$v" <<'PY'
import sys,json
try:
    d=json.loads(open('gpurun_out/r4misc/s1.json').read()); k=d['kernel_ms']; h=d['host_ms_per_step']
    print('S=1 devmap', sys.argv[1], 'fps %.0f ms/step %.3f in_abi %.3f ' % (d['value'], d['ms_per_step'], h['in_abi_calls']), ', '.join('%s %.3f' % (a, b/d['steps']) for a,b in k.items()), 'kf', d['config']['keyframes_in_timed_region'], d['config']['checks'], d['config']['map'][:6])
except Exception as e:
    print('FAILED', e); print(open('gpurun_out/r4misc/err.log').read()[-1500:])
PY
done
python bench.py --steps 20 --warmup 5 --backend-mode 2 --no-cpu-baseline --spread-windows 0 --host-input-steps 0 --solo-steps 0 --full-res-streams 0 2>gpurun_out/r4misc/err2.log | tail -1 > gpurun_out/r4misc/m2.json
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r4misc/m2.json').read())
    print('12288 streams mode 2 devmap: value', d['value'], 'cpus_busy', d['host_ms_per_step']['cpus_busy'], d['config']['map'][:6], d['config']['checks'])
except Exception as e:
    print('FAILED', e); print(open('gpurun_out/r4misc/err2.log').read()[-1500:])
PY
