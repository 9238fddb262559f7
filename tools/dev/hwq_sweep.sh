cd /root/repo; mkdir -p gpurun_out/s3
( for q in 8; do export GPU_MAX_HW_QUEUES=$q; echo "#### GPU_MAX_HW_QUEUES=$q"; bash tools/dev/sweep_line.sh "--groups 4" "--groups 5" "--groups 6" "--groups 8"; done
export GPU_MAX_HW_QUEUES=16; echo "#### GPU_MAX_HW_QUEUES=16"; bash tools/dev/sweep_line.sh "--groups 8" "--groups 12"
unset GPU_MAX_HW_QUEUES; echo "#### default"; bash tools/dev/sweep_line.sh "--groups 4" ) > gpurun_out/s3/sweep_hwq.txt 2>&1
cat gpurun_out/s3/sweep_hwq.txt
