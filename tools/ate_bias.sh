#!/bin/bash
# VERDICT r5 item 3 — the HIP - twin ATE difference by cause: tests/ate_bias.py over two seed sets; the twin's side once per set.
# Library variants are built in the build container into tools/bin/ab/{ieee,nocontract,all3} (tools/README.md: ab.sh), copied
# over the box's scratch copy of stereovision-slam_amd/lib per run and restored at the end.
#   tools/ate_bias.sh [streams] [frames]     -> gpurun_out/ate_bias/{*.npz,report.txt}
cd "$(dirname "$0")/.." || exit 1
N=${1:-6144}; F=${2:-320}
O=gpurun_out/ate_bias; mkdir -p $O tools/bin/ab/base
L=stereovision-slam_amd/lib
cp $L/*.so tools/bin/ab/base/
hip() {   # tag variant-dir seed0 [env]
  cp tools/bin/ab/$2/*.so $L/ || exit 1
  env $4 python tests/ate_bias.py hip $3 $N $F $O/hip_$1_$3.npz 2>> $O/log.txt | tail -1
  cp tools/bin/ab/base/*.so $L/
}
for S in 0x5EED1000 0x5EED9000; do
  ( time python tests/ate_bias.py twin $S $N $F $O/twin_$S.npz 2>> $O/log.txt | tail -1 ) 2>&1 | grep -v "^$" | head -3
  hip base base $S ""
  hip all3 all3 $S "SVSLAM_PO_XTOL=0"
  if [ $S = 0x5EED1000 ]; then
    hip ieee ieee $S ""
    hip nocontract nocontract $S ""
    hip xtol0 base $S "SVSLAM_PO_XTOL=0"
    hip base_again base $S ""
  fi
done
{
python tests/ate_bias.py report $O/twin_0x5EED1000.npz base=$O/hip_base_0x5EED1000.npz base_again=$O/hip_base_again_0x5EED1000.npz ieee=$O/hip_ieee_0x5EED1000.npz nocontract=$O/hip_nocontract_0x5EED1000.npz xtol0=$O/hip_xtol0_0x5EED1000.npz all3=$O/hip_all3_0x5EED1000.npz
echo
python tests/ate_bias.py report $O/twin_0x5EED9000.npz base=$O/hip_base_0x5EED9000.npz all3=$O/hip_all3_0x5EED9000.npz
} > $O/report.txt 2>&1
cat $O/report.txt
