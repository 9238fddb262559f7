# layout sweep with the device-resident map: streams per GPU x groups.  usage: tools/sweep_devmap.sh "S1 S2 .." "G1 G2 .." [bench args]
SS=${1:-"12288"}; GG=${2:-"4 6 8 12 16"}; shift; shift
for S in $SS; do for G in $GG; do
python bench.py --no-cpu-baseline --spread-windows 1 --host-input-steps 0 --solo-steps 0 --streams $S --groups $G "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); h=d['host_ms_per_step']; print('S $S G $G', d['value'], d['value_spread']['windows'], 'ms/step', d['ms_per_step'], 'cpus', h['cpus_busy'], 'rss_gb', h['rss_gb'], {k: round(v['avg_launch_us']) for k, v in d['roofline_by_family'].items()})"
done; done
