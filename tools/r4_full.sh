cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4full
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r4full/test.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4full/test.log
tail -8 gpurun_out/r4full/test.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4full/bench_20_5.json 2> gpurun_out/r4full/bench_err.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4full/bench_20_5.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline'])
PY
