# A/B of two builds of libsvslam_hip.so (lib/libsvslam_hip_A.so, _B.so) on the same box, alternating
L=stereovision-slam_amd/lib
for i in 1 2 3; do
for v in A B; do
cp $L/libsvslam_hip_$v.so $L/libsvslam_hip.so
python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); h=d['host_ms_per_step']; k=d['kernel_ms']; print('$v', d['value'], h['in_step'], h['stream_wait'], h['cpus_busy'], 'ba %.0f lk %.0f' % (k['local_ba'], k['lk']))"
done; done
cp $L/libsvslam_hip_A.so $L/libsvslam_hip.so
