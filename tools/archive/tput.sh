python tools/kbench.py tput 2>&1 | grep -E "^BA|rep 1" | cut -c1-400
