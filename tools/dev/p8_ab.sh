#!/bin/bash
# development (round 6): whole-pair Schur task (BA_SCHUR_PAIR8) against the three row-group tasks: fingerprints, batch and low-latency solver
cd "$(dirname "$0")/../.." || exit 1
bash tools/ab.sh "p8off p8" 1 -- python tools/dev/ba_bits.py
bash tools/ab.sh "p8off p8 p8off p8" 1 -- python tools/kbench.py ba1
bash tools/ab.sh "p8off p8" 1 -- python tools/kbench.py ball
