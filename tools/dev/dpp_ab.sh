#!/bin/bash
# development: A/B of a library change over the kernels that use the DPP helpers (bit fingerprints of the BA, kernel benches)
mkdir -p gpurun_out/r5
bash tools/ab.sh "c9 base" 1 -- python tools/dev/ba_bits.py
bash tools/ab.sh "c9 base" 2 -- bash -c 'timeout 200 python tools/kbench.py ba1 2>&1 | grep "rep 1" | cut -c1-120; timeout 100 python tools/kbench.py gftt 2>&1 | grep "rects= 80"; timeout 100 python tools/po_trace.py 2>&1 | tail -2; timeout 200 python tools/kbench.py ball 2>&1 | tail -3 | cut -c1-160'
