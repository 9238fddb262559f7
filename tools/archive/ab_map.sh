# host-resident map vs device-resident map, same library, same box, alternating.  usage: tools/ab_map.sh [rounds] [bench args...]
R=${1:-2}; shift
for i in $(seq $R); do
for v in "--host-map" ""; do
python bench.py --no-cpu-baseline --spread-windows 2 --host-input-steps 0 --solo-steps 0 $v "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); h=d['host_ms_per_step']; k=d['kernel_ms']; print('${v:-device-map}', d['value'], 'spread', d['value_spread']['windows'], 'ms/step', d['ms_per_step'], 'in_step', h['in_step'], 'wait', h['stream_wait'], 'cpus', h['cpus_busy'], h['cpus_busy_by_thread_name'], 'rss/frame', h['rss_growth_bytes_per_frame'], 'rss_gb', h['rss_gb'], 'ba %.0f lk %.0f pyr %.0f po %.0f gftt %.0f tri %.0f' % (k['local_ba'], k['lk'], k['pyramid'], k['pose_only'], k['gftt'], k['triangulate']), d['config']['checks'], d['config']['ba_problem_mean'])"
done; done
