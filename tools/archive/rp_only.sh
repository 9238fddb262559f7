cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/prof; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp -o rp -- python bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err
f=$(find $O/rp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv
python tools/trace_busy.py $O/rp > $O/trace_busy.txt 2>&1
rm -rf $O/rp
head -8 $O/kernel_stats.csv; head -12 $O/trace_busy.txt
