#!/bin/bash
# round 3, experiment 25: frames in flight against the schedule: config 3 (dragon) and 1e5 / 1e6 primitives, bench.py --schedule X
for w in "config3" "synthetic --prims 100000" "synthetic --prims 1000000"; do for s in default wavefront tree tile; do
  timeout -k 10 300 python bench.py --workload $w --schedule $s --no-cpu-baseline 2>/dev/null > gpurun_out/r3_25.json || { echo "$w $s failed"; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/r3_25.json')); print('$w', '$s', round(d['value'],1), 'Mrays/s', round(d['ms_per_frame'],4), 'ms/frame in flight', d['config'].get('schedule'), d['config'].get('frames_in_flight'))"
done; done
