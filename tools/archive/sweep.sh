# host layout sweep: streams, groups (host threads driving disjoint stream groups), bookkeeping threads per group, steps
for cfg in "8192 12 3 100" "8192 16 2 100" "8192 8 4 100" "6144 8 4 100" "6144 12 3 100" "12288 16 2 20" "12288 12 3 20" "8192 12 3 20" "0 0 0 100" "0 0 0 20"; do set -- $cfg
timeout 500 python bench.py --streams $1 --groups $2 --host-threads $3 --steps $4 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); h=d['host_ms_per_step']; c=d['config']; print('$cfg ->', c['streams_per_gpu'], c['host_threads_per_gpu'], c['bookkeeping_threads_per_group'], ':', d['value'], d['ms_per_step'], 'in_step', h['in_step'], 'abi', h['in_abi_calls'], 'cpus', h['cpus_busy'])" 2>&1 | tail -1
done
