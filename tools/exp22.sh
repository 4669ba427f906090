#!/bin/bash
# GPU test suite + schedule probe.  Exit non-zero on a GPU fault or a failure.
O=gpurun_out/exp22; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?
tail -5 $O/pytest.txt
if grep -q "Memory access fault" $O/pytest.txt; then echo "GPU FAULT"; exit 99; fi
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python tools/stream_probe.py > $O/probe.txt 2>&1; rc=$?
grep -E "dragon|synthetic|MISMATCH" $O/probe.txt
if grep -q "Memory access fault" $O/probe.txt; then echo "GPU FAULT"; exit 99; fi
exit $rc
