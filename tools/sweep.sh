# host layout sweep: streams, groups (host threads driving disjoint stream groups), bookkeeping threads per group
for cfg in "6144 8 4" "6144 12 3" "6144 16 2" "6144 12 2" "6144 16 3" "8192 16 2" "6144 8 4" "6144 12 3" "6144 16 2"; do set -- $cfg
timeout 400 python bench.py --streams $1 --groups $2 --host-threads $3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); h=d['host_ms_per_step']; print('$cfg', d['value'], d['ms_per_step'], h['in_step'], h['stream_wait'], h['cpus_busy'])" 2>&1 | tail -1
done
