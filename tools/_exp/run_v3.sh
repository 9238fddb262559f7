# swap in the experimental library for this run only
cp stereovision-slam_amd/lib/libsvslam_hip.so /tmp/orig.so
cp tools/_exp/libsvslam_hip_v3.so stereovision-slam_amd/lib/libsvslam_hip.so
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "local_ba" 2>&1 | tail -2
bash tools/tput.sh
cp /tmp/orig.so stereovision-slam_amd/lib/libsvslam_hip.so
