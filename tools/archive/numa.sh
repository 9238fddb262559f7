# NUMA placement A/B: bench with its host threads pinned to the GPU's NUMA node (default) vs --no-pin
run() { python bench.py --no-cpu-baseline $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); h=d['host_ms_per_step']; print('$1', d['value'], h['in_step'], h['stream_wait'], h['cpus_busy'], h['pinned_to_gpu_numa_cpus'])"; }
for rep in 1 2 3; do run pinned; run unpinned --no-pin; done
