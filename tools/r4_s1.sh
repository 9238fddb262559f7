cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4s1
run() {
python bench.py --streams 1 --groups 1 --host-threads 1 --steps 300 --warmup 20 --no-cpu-baseline --spread-windows 0 --host-input-steps 0 --solo-steps 0 --low-latency $1 2>gpurun_out/r4s1/err.log | tail -1 > gpurun_out/r4s1/s1.json
python - "$ENVTAG $1" <<'PY'
import sys,json
try:
    d=json.loads(open('gpurun_out/r4s1/s1.json').read()); k=d['kernel_ms']; h=d['host_ms_per_step']
    print('S=1', sys.argv[1], 'fps %.0f ms/step %.3f in_abi %.3f ' % (d['value'], d['ms_per_step'], h['in_abi_calls']), ', '.join('%s %.3f' % (a, b/d['steps']) for a,b in k.items()), 'kf', d['config']['keyframes_in_timed_region'], d['config']['checks'])
except Exception as e:
    print('FAILED', sys.argv[1], e); print(open('gpurun_out/r4s1/err.log').read()[-1500:])
PY
}
for z in 0 1; do
export SVSLAM_ZERO_COPY=$z; ENVTAG="zc=$z"
run "--backend-mode 1"
done
run "--backend-mode 2 --backend-lag 1"
run "--backend-mode 2 --backend-lag 6"
run "--backend-mode 1 --host-map"
