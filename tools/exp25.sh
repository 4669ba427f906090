#!/bin/bash
# dragon / synthetic timings on every schedule (no test suite)
O=gpurun_out/exp25; mkdir -p $O
timeout -k 10 300 python tools/perf_probe.py dragon --occ 0,6 --n 20 > $O/dragon.txt 2>&1 || { tail -5 $O/dragon.txt; exit 1; }
grep -E "hbm/lane" $O/dragon.txt
timeout -k 10 300 python tools/perf_probe.py --synthetic 1000000 --n 10 > $O/syn6.txt 2>&1 || { tail -5 $O/syn6.txt; exit 1; }
grep -E "hbm/lane" $O/syn6.txt
if grep -q "Memory access fault" $O/*.txt; then echo "GPU FAULT"; exit 99; fi
