#!/bin/bash
# development (round 6): the pyramid fill by unaligned 16-byte loads / workgroup size / LDS budget, A/B on one box.
#   tools/dev/pyr_ab.sh "old ua ua512" -> per variant: the pyramid parity tests, then one bench line (value, kernel ms, solo pyramid)
cd "$(dirname "$0")/../.." || exit 1
L=stereovision-slam_amd/lib
mkdir -p tools/bin/ab/base && cp $L/*.so tools/bin/ab/base/
line() {
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --spread-windows 3 --super-windows 0 --host-input-steps 0 --predecimated-streams 0 "$@" 2>/dev/null | tail -1 | tee -a gpurun_out/s3/pyr_ab_raw.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms']; w = d['value_windows']; s = d.get('roofline_solo') or {}
print('value %.0f (windows min %.0f max %.0f) ms/step %.3f  kernel-ms: ' % (d['value'], w['min'], w['max'], d['ms_per_step']) + ', '.join('%s %.0f' % (a, b) for a, b in k.items()) + '  solo us: ' + ', '.join('%s %.0f' % (a, b['avg_launch_us']) for a, b in s.items() if isinstance(b, dict) and 'avg_launch_us' in b))"
}
for spec in $1; do
v=${spec%%:*}; kb=${spec#*:}; [ "$kb" = "$spec" ] && kb=""
cp tools/bin/ab/$v/*.so $L/ || exit 1
echo "== variant $v  SVSLAM_PYR_LDS_KB=${kb:-default}"
[ -n "$kb" ] && export SVSLAM_PYR_LDS_KB=$kb || unset SVSLAM_PYR_LDS_KB
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "pyramid" 2>&1 | tail -2
line
done
unset SVSLAM_PYR_LDS_KB
cp tools/bin/ab/base/*.so $L/
