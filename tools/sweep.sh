for cfg in "4096 8 4" "4096 8 2" "4096 16 1" "6144 12 2" "2048 8 4" "4096 4 8"; do set -- $cfg
python bench.py --streams $1 --groups $2 --host-threads $3 --steps 150 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], d['host_ms_per_step'], {k:int(v) for k,v in d['kernel_ms'].items()})"
done
