#!/bin/bash
# round 3, experiment 9: after taking the register-hungry experiments out of the kernels: full GPU suite, config 2 back to its round-2 rate?
set -e
O=gpurun_out/r3_09; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for rep in 1 2; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print({k:d[k] for k in ('value','ms_per_frame','ms_per_frame_latency')}, d['roofline']['kernel_ms_live'], d['config']['frame_checksum'])" >> $O/config2.txt
done
cat $O/config2.txt
timeout -k 10 300 python tools/schedule_probe.py 2>&1 | grep -E "mount|balls|dragon|synthetic|MISMATCH" > $O/probe.txt; cat $O/probe.txt
timeout -k 10 300 python tools/config4.py 2>&1 | grep -E "device" > $O/config4.txt; cat $O/config4.txt
for a in "mount_low wavefront 1" "mount_low wavefront 2"; do timeout -k 10 120 python tools/wave_timeline.py $a 2>&1 | grep -E "frame|span|peak|lifetime|batches|first batch" >> $O/timelines.txt; done; cat $O/timelines.txt
