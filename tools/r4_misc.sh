cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4misc
timeout 1500 python -m pytest tests/test_facade_kitti.py tests/test_gpu_scale_launch.py tests/test_gpu_device_map.py tests/test_gpu_pipeline.py -x -q -m gpu > gpurun_out/r4misc/test.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4misc/test.log
tail -25 gpurun_out/r4misc/test.log
