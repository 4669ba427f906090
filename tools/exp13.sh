#!/bin/bash
set -o pipefail
O=gpurun_out/exp13; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc $rc" >> $O/pytest.txt
grep -v "^  File\|^Extension" $O/pytest.txt | tail -n 12
if grep -q "Memory access fault" $O/pytest.txt; then exit 9; fi
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/perf_probe.py mount_low 1920 1080 --n 100 > $O/probe_c2.txt 2>&1 && grep -h "lds/lane" $O/probe_c2.txt
timeout -k 10 300 python tools/perf_probe.py dragon 1920 1080 --n 10 --occ 0,5,6 > $O/probe_c3.txt 2>&1 && grep -h "hbm/lane" $O/probe_c3.txt
timeout -k 10 300 python tools/perf_probe.py --synthetic 1000000 --n 10 --occ 0,5,6 > $O/probe_syn.txt 2>&1 && grep -h "hbm/lane" $O/probe_syn.txt
timeout -k 10 300 python tools/config4.py > $O/c4.txt 2>&1 && tail -n 3 $O/c4.txt
