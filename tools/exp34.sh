#!/bin/bash
O=gpurun_out/exp34; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?
tail -3 $O/pytest.txt
if grep -q "Memory access fault" $O/pytest.txt; then echo "GPU FAULT"; exit 99; fi
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" $O/pytest.txt | head -20; exit $rc; }
echo "--- pair mode off"; P3D_NO_PAIR_MODE=1 python3 tools/perf_probe.py mount_low --n 300 2>&1 | grep -E "wavefront lds/lane"
echo "--- pair mode on"; python3 tools/perf_probe.py mount_low --n 300 2>&1 | grep -E "wavefront lds/lane"
P3D_NO_PAIR_MODE=1 timeout -k 10 300 python tools/shard_probe.py wavefront 2>&1 | grep -E " 4 in flight"
timeout -k 10 300 python tools/shard_probe.py wavefront 2>&1 | grep -E " 4 in flight"
python3 tools/perf_probe.py --synthetic 1000000 --n 10 2>&1 | grep -E "wavefront hbm/lane"
P3D_NO_PAIR_MODE=1 python3 tools/perf_probe.py --synthetic 1000000 --n 10 2>&1 | grep -E "wavefront hbm/lane"
