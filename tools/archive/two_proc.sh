# is one process the limit?  two bench processes on the same GPU, half the streams each, vs one
run() { python bench.py --no-cpu-baseline --streams $1 --groups $2 --host-threads 4 --steps 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); h=d['host_ms_per_step']; print('$3', d['value'], d['ms_per_step'], h['cpus_busy'])"; }
run 6144 8 single
( run 3072 4 procA ) & ( run 3072 4 procB ) & wait
run 6144 8 single
( run 3072 4 procA ) & ( run 3072 4 procB ) & wait
