cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dev
python "$@" > gpurun_out/dev/out.txt 2>&1; echo rc $? >> gpurun_out/dev/out.txt; tail -60 gpurun_out/dev/out.txt
