for cfg in "6144 12 3 100" "6144 8 4 100" "6144 12 3 100" "6144 8 4 100" "6144 12 3 100" "6144 8 4 100"; do set -- $cfg
timeout 400 python bench.py --streams $1 --groups $2 --host-threads $3 --steps $4 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); h=d['host_ms_per_step']; print('$cfg', d['value'], d['ms_per_step'], h['in_step'], h['stream_wait'], h['cpus_busy'])" 2>&1 | tail -1
done
