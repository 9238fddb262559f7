# A/B of two complete library sets (stereovision-slam_amd/lib_A/*.so vs lib_B/*.so: HIP kernels + host pipeline) on the
# same box, alternating; leaves lib_B in place.   usage: tools/ab_dirs.sh [rounds] [bench args...]
L=stereovision-slam_amd
R=${1:-3}; shift
for i in $(seq $R); do
for v in A B; do
cp $L/lib_$v/*.so $L/lib/
python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); h=d['host_ms_per_step']; k=d['kernel_ms']; print('$v', d['value'], 'ms/step', d['ms_per_step'], 'in_step', h['in_step'], 'wait', h['stream_wait'], 'cpus', h['cpus_busy'], 'rss/frame', h['rss_growth_bytes_per_frame'], 'ba %.0f lk %.0f pyr %.0f po %.0f gftt %.0f' % (k['local_ba'], k['lk'], k['pyramid'], k['pose_only'], k['gftt']))"
done; done
cp $L/lib_B/*.so $L/lib/
