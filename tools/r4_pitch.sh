cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4pitch
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_edge_cases.py tests/test_gpu_pipeline.py -x -q -m gpu > gpurun_out/r4pitch/test.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4pitch/test.log
tail -5 gpurun_out/r4pitch/test.log
python tools/kbench.py lk 2>&1 | grep "max_count=30" 
python tools/kbench.py gftt 2>&1 | tail -4
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --spread-windows 2 --host-input-steps 0 --solo-steps 0 --full-res-streams 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('value', d['value'], 'spread', d['value_spread']['windows'], {k:v['avg_launch_us'] for k,v in d['roofline_by_family'].items()})"
