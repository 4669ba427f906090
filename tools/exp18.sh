#!/bin/bash
# stream schedule: first light.  Exit non-zero on a GPU fault or a mismatch.
O=gpurun_out/exp18; mkdir -p $O
timeout -k 10 600 python tools/stream_probe.py "$@" > $O/probe.txt 2>&1; rc=$?
tail -40 $O/probe.txt
if grep -q "Memory access fault" $O/probe.txt; then echo "GPU FAULT"; exit 99; fi
exit $rc
