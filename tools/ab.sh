for i in 1 2 3; do
for lib in A B; do
if [ $lib = A ]; then export SVS_PIPELINE_LIB=$PWD/stereovision-slam_amd/lib/libsvslam_pipeline_A.so; else unset SVS_PIPELINE_LIB; fi
python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); h=d['host_ms_per_step']; print('$lib', d['value'], h['in_step'], h['stream_wait'], h['cpus_busy'])"
done; done
