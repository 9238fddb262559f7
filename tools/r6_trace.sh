#!/bin/bash
# A lone camera's frame on the time line: rocprofv3 kernel trace of bench.py --streams 1 --low-latency, split into
# tracked frames and keyframes; per kernel of the chain its mean duration and the mean gap in front of it (= dependent-launch
# latency + host work between the calls).  The sum is the frame's floor at this kernel set.   -> gpurun_out/r6trace/budget_*.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6trace; mkdir -p $O
for tag in predecimated fullres; do
  fr=""; [ $tag = predecimated ] && fr="--pre-decimated"
  for mode in "1" "2 --backend-lag 6"; do
    m=${mode%% *}
    rm -rf $O/prof_$tag$m
    rocprofv3 --kernel-trace -d $O/prof_$tag$m -o s1 --output-format csv -- python bench.py $fr --streams 1 --groups 1 --host-threads 1 --steps 300 --warmup 20 --ring-frames 160 --no-cpu-baseline --spread-windows 0 --super-windows 0 --host-input-steps 0 --solo-steps 0 --predecimated-streams 0 --low-latency --backend-mode $mode > $O/s1_$tag$m.log 2>&1
    python - $O/prof_$tag$m "$tag, backend mode $mode" > $O/budget_$tag$m.txt <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/s1_kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'synth' not in r['Kernel_Name'] and 'rocclr' not in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
name = lambda r: r['Kernel_Name'].split('(')[0].replace('void ', '')[:34]
# a frame starts at k_pyr_fused whose follower is k_lk (the tracking launch); keep the last 300 frames (the timed steps)
starts = [i for i, r in enumerate(rows[:-1]) if name(r).startswith('k_pyr_fused') and name(rows[i + 1]).startswith('k_lk')]
starts = starts[-301:]
frames = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
kinds = {'tracked frame': [fr for fr in frames if not any('k_dmap_begin' in name(r) for r in fr)],
         'keyframe': [fr for fr in frames if any('k_dmap_begin' in name(r) for r in fr)]}
print('S = 1, %s: %d frames of the timed region (%d keyframes)' % (sys.argv[2], len(frames), len(kinds['keyframe'])))
period = [(int(b[0]['Start_Timestamp']) - int(a[0]['Start_Timestamp'])) / 1e3 for a, b in zip(frames[:-1], frames[1:])]
print('mean frame period %.1f us = %.0f frames/s' % (sum(period) / len(period), 1e6 * len(period) / sum(period)))
for kind, frs in kinds.items():
    if not frs: continue
    acc = collections.OrderedDict()
    span = 0.0
    for fr in frs:
        prev_end = None
        for j, r in enumerate(fr):
            s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            k = '%02d %s' % (j, name(r))
            a = acc.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3; a[2] += ((s - prev_end) / 1e3 if prev_end is not None else 0.0)
            prev_end = e if prev_end is None else max(prev_end, e)
        span += (prev_end - int(fr[0]['Start_Timestamp'])) / 1e3
    n = len(frs)
    print('\n%s (%d): first kernel start -> last kernel end %.1f us; per kernel of the chain: mean duration, mean gap in front of it' % (kind, n, span / n))
    td = tg = 0.0
    for k, (c, d, g) in acc.items():
        if c < 0.5 * n: continue
        print('  %-38s dur %7.1f us   gap %6.1f us   (in %d of %d)' % (k, d / c, g / c, c, n)); td += d / c; tg += g / c
    print('  %-38s dur %7.1f us   gap %6.1f us' % ('sum', td, tg))
PY
    rm -rf $O/prof_$tag$m
  done
done
cat $O/budget_*.txt
