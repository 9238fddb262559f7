cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4trace
rocprofv3 --kernel-trace --stats -d gpurun_out/r4trace/prof_s1 -o s1 --output-format csv -- python bench.py --streams 1 --groups 1 --host-threads 1 --steps 100 --warmup 20 --no-cpu-baseline --spread-windows 0 --host-input-steps 0 --solo-steps 0 --full-res-streams 0 --low-latency --backend-mode 1 > gpurun_out/r4trace/s1_prof.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r4trace/prof_s1/**/s1_kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find a keyframe chain near the end: locate last k_dmap_begin
idx=[i for i,r in enumerate(rows) if 'k_dmap_begin' in r['Kernel_Name']]
i0=idx[-3]-12
t0=int(rows[i0]['Start_Timestamp']); prev=None
for r in rows[i0:i0+48]:
    s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
    gap=(s-prev) if prev is not None else 0
    print('%9.1f us  dur %7.1f  gap %6.1f  %s' % (s/1e3,(e-s)/1e3,gap/1e3,r['Kernel_Name'][:46]))
    prev=e
PY
