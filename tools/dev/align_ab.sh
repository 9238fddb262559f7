#!/bin/bash
# development: does code alignment move the kernels? (kernel benches of one build)
timeout 100 python tools/po_trace.py 2>&1 | tail -3
timeout 200 python tools/kbench.py fe 2>&1 | tail -1 | cut -c1-200
timeout 200 python tools/kbench.py lk 2>&1 | grep "max_count=30"
timeout 100 python tools/kbench.py gftt 2>&1 | grep "rects= 80"
timeout 200 python tools/kbench.py ba1 2>&1 | grep "rep 1" | cut -c1-40
