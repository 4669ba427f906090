#!/bin/bash
O=gpurun_out/exp41; mkdir -p $O
timeout -k 10 600 python tools/schedule_probe.py 2>&1 | grep -E "dragon|synthetic|MISMATCH"
timeout -k 10 300 python tools/perf_probe.py dragon --tree --occ 0,6 --n 20 2>&1 | grep -E "tree      hbm"
