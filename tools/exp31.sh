#!/bin/bash
O=gpurun_out/exp31; mkdir -p $O
timeout -k 10 300 python tools/perf_probe.py dragon --tree --chunks 1,2,4,8,16,64,256 --n 20 > $O/dragon.txt 2>&1 || { tail -5 $O/dragon.txt; exit 1; }
grep -E "^tree" $O/dragon.txt
timeout -k 10 300 python tools/perf_probe.py --synthetic 1000000 --chunks 1,4,16,64 --n 10 > $O/syn6.txt 2>&1 || { tail -5 $O/syn6.txt; exit 1; }
grep -E "^wavefront hbm/lane" $O/syn6.txt
