#!/bin/bash
# per-kernel time of one command under rocprofv3 (kernel trace + stats); prints the stats CSV head
#   tools/rpstats.sh <tag> <command...>     -> gpurun_out/<tag>_kernel_stats.csv
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
tag=$1; shift
O=gpurun_out/rp_$tag
mkdir -p gpurun_out "$O"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o rp -- "$@" > gpurun_out/${tag}_cmd.log 2>&1 < /dev/null
f=$(find "$O" -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" gpurun_out/${tag}_kernel_stats.csv; cut -c1-160 "$f" | head -${RP_HEAD:-14}; else echo "no kernel stats produced"; tail -5 gpurun_out/${tag}_cmd.log; fi
rm -rf "$O"
