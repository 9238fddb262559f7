for cfg in "4096 8 4 1" "4096 8 4 2" "4096 8 2 1" "4096 8 2 2" "4096 8 4 1" "4096 8 4 2" "3072 6 4 2" "3072 6 4 1"; do set -- $cfg
python bench.py --streams $1 --groups $2 --host-threads $3 --backend-mode $4 --steps 150 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], d['host_ms_per_step'])"
done
