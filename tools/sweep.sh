for cfg in "8192 8 4 50" "8192 16 2 50" "6144 8 4 60" "8192 32 1 50"; do set -- $cfg
timeout 400 python bench.py --streams $1 --groups $2 --host-threads $3 --steps $4 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], d['host_ms_per_step'])" 2>&1 | tail -1
done
