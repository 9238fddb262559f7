# refresh the per-round evidence under gpurun_out/ (copy what is to be judged into profiles/):
#   default bench line, the same command under rocprofv3 --kernel-trace --stats, the full-resolution
#   bench line, the small-S latency table
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/prof; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp -o rp -- python bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err
f=$(find $O/rp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv
python tools/trace_busy.py $O/rp > $O/trace_busy.txt 2>&1
rm -rf $O/rp
python bench.py --full-res --steps 60 --no-cpu-baseline > $O/bench_full_res.json 2> $O/bench_full_res.err
bash tools/lat.sh > $O/latency_small_S.txt 2>&1
tail -c 600 $O/bench_default.json; head -8 $O/kernel_stats.csv; cat $O/trace_busy.txt | head -20
