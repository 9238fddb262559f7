for cfg in "8192 8 4 60" "6144 8 4 80" "4096 8 4 100" "8192 8 4 60"; do set -- $cfg
timeout 400 python bench.py --streams $1 --groups $2 --host-threads $3 --steps $4 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], d['host_ms_per_step'])" 2>&1 | tail -1
done
