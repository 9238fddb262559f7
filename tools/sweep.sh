# host layout sweep: streams, groups (host threads driving disjoint stream groups), bookkeeping threads per group
for cfg in "6144 8 4" "6144 12 3" "6144 16 2" "8192 8 4" "4096 8 4" "6144 6 5" "6144 10 3" "6144 8 4"; do set -- $cfg
timeout 400 python bench.py --streams $1 --groups $2 --host-threads $3 --steps 60 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); h=d['host_ms_per_step']; print('$cfg', d['value'], d['ms_per_step'], 'in_step', h['in_step'], 'abi', h['in_abi_calls'], 'wait', h['stream_wait'], 'cpus', h['cpus_busy'], 'ba_prep', h['ba_host_prep'])" 2>&1 | tail -1
done
