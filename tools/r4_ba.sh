cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4ba
timeout 1200 python -m pytest tests/test_gpu_ll_ba.py tests/test_gpu_lm_parity.py tests/test_gpu_parity.py tests/test_shared_map_ba.py -x -q -m gpu > gpurun_out/r4ba/test.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4ba/test.log
tail -15 gpurun_out/r4ba/test.log
timeout 300 python tools/kbench.py ball > gpurun_out/r4ba/kbench_ball.txt 2>&1
grep -A4 "16 shards" gpurun_out/r4ba/kbench_ball.txt | cut -c1-330
tail -3 gpurun_out/r4ba/kbench_ball.txt | cut -c1-330
timeout 300 python tools/kbench.py ba1 > gpurun_out/r4ba/kbench_ba1.txt 2>&1
grep "rep 1" gpurun_out/r4ba/kbench_ba1.txt | cut -c1-330
