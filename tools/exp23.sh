#!/bin/bash
# GPU tests + config 2 / config 3 / synthetic timings.  Exit non-zero on a GPU fault or a failure.
O=gpurun_out/exp23; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?
tail -3 $O/pytest.txt
if grep -q "Memory access fault" $O/pytest.txt; then echo "GPU FAULT"; exit 99; fi
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/perf_probe.py mount_low --n 200 > $O/c2.txt 2>&1 || exit 1
grep -E "wavefront lds/lane|tile      lds/lane" $O/c2.txt
timeout -k 10 600 python tools/stream_probe.py > $O/probe.txt 2>&1; rc=$?
grep -E "dragon|synthetic|MISMATCH" $O/probe.txt | grep -v stream
if grep -q "Memory access fault" $O/probe.txt $O/c2.txt; then echo "GPU FAULT"; exit 99; fi
exit $rc
