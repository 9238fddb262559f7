#!/bin/bash
# A/B of library variants on ONE box, alternating: tools/ab.sh "A B C" [rounds] -- <command...>
# Variants are built in the build container into tools/bin/ab/<X>/ (git-ignored, but shipped to the GPU box):
#   SVS_LIB_DIR=tools/bin/ab/B SVS_EXTRA_HIP_FLAGS="-DBA_NUM_VGPR=96" python stereovision-slam_amd/build.py --force
# Each run copies the variant's libraries over stereovision-slam_amd/lib/ (the box's copy of the tree is scratch); "base" is
# the committed build, saved first and restored at the end.  Never leaves a variant inside the product package in the repo.
cd "$(dirname "$0")/.." || exit 1
L=stereovision-slam_amd/lib
SETS=$1; shift
R=1; if [ "$1" != "--" ]; then R=$1; shift; fi
shift
mkdir -p tools/bin/ab/base && cp $L/*.so tools/bin/ab/base/
for i in $(seq $R); do
for v in $SETS; do
cp tools/bin/ab/$v/*.so $L/ || exit 1
echo "== variant $v (round $i)"
"$@"
done; done
cp tools/bin/ab/base/*.so $L/
