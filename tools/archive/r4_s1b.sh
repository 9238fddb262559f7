# S = 1 operating point with and without the per-family HIP events of the measurement (value = with, value_spread = without)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4s1
run() {
python bench.py --streams 1 --groups 1 --host-threads 1 --steps 300 --warmup 20 --no-cpu-baseline --spread-windows 3 --host-input-steps 0 --solo-steps 0 --full-res-streams 0 --low-latency $1 2>gpurun_out/r4s1/err.log | tail -1 > gpurun_out/r4s1/s1.json
python - "$1" <<'PY'
import sys,json
try:
    d=json.loads(open('gpurun_out/r4s1/s1.json').read()); k=d['kernel_ms']
    print('S=1', sys.argv[1], 'fps %.0f ms/step %.3f' % (d['value'], d['ms_per_step']), 'spread', d.get('value_spread'), 'kf', d['config']['keyframes_in_timed_region'])
except Exception as e:
    print('FAILED', sys.argv[1], e); print(open('gpurun_out/r4s1/err.log').read()[-1500:])
PY
}
run "--backend-mode 1"
run "--backend-mode 2 --backend-lag 1"
run "--backend-mode 2 --backend-lag 6"
