#!/bin/bash
# GPU tests, then dragon / synthetic timings on every schedule
O=gpurun_out/exp24; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?
tail -3 $O/pytest.txt
if grep -q "Memory access fault" $O/pytest.txt; then echo "GPU FAULT"; exit 99; fi
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/perf_probe.py dragon --occ 0,5 --n 20 > $O/dragon.txt 2>&1 || { tail -5 $O/dragon.txt; exit 1; }
grep -E "hbm/lane" $O/dragon.txt
timeout -k 10 300 python tools/perf_probe.py dragon --depth 6 --n 10 > $O/dragon6.txt 2>&1 || { tail -5 $O/dragon6.txt; exit 1; }
echo depth 6; grep -E "hbm/lane" $O/dragon6.txt
if grep -q "Memory access fault" $O/*.txt; then echo "GPU FAULT"; exit 99; fi
