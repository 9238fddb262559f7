cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4base
for v in "--backend-mode 1" "--backend-mode 2 --backend-lag 1"; do
python bench.py --streams 1 --groups 1 --host-threads 1 --steps 300 --warmup 20 --no-cpu-baseline --spread-windows 0 --host-input-steps 0 --solo-steps 0 --low-latency $v 2>/dev/null | tail -1 > gpurun_out/r4base/s1_$(echo $v | tr -d ' -').json
done
rocprofv3 --kernel-trace --stats -d gpurun_out/r4base/prof_s1 -o s1 --output-format csv -- python bench.py --streams 1 --groups 1 --host-threads 1 --steps 100 --warmup 20 --no-cpu-baseline --spread-windows 0 --host-input-steps 0 --solo-steps 0 --low-latency --backend-mode 1 > gpurun_out/r4base/s1_prof.log 2>&1
python tools/kbench.py ba1 > gpurun_out/r4base/kbench_ba1.txt 2>&1
ls -R gpurun_out/r4base | head -30
