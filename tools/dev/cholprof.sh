cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dev
cp stereovision-slam_amd/lib/libsvslam_hip.so /tmp/keep.so
cp stereovision-slam_amd/lib_P/libsvslam_hip.so stereovision-slam_amd/lib/libsvslam_hip.so
SVSLAM_BA_PROF_EXTRA=1 python tools/kbench.py ball > gpurun_out/dev/cholprof.txt 2>&1
SVSLAM_BA_PROF_EXTRA=1 python tools/kbench.py ba1 >> gpurun_out/dev/cholprof.txt 2>&1
cp /tmp/keep.so stereovision-slam_amd/lib/libsvslam_hip.so
grep -B1 "chol wave" gpurun_out/dev/cholprof.txt | cut -c1-250 | tail -40
