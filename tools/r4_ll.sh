cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4ll
timeout 900 python -m pytest tests/test_gpu_ll_ba.py -x -q -s -m gpu > gpurun_out/r4ll/test_ll.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4ll/test_ll.log
tail -30 gpurun_out/r4ll/test_ll.log
timeout 300 python tools/kbench.py ball > gpurun_out/r4ll/kbench_ball.txt 2>&1
cat gpurun_out/r4ll/kbench_ball.txt
