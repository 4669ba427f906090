#!/bin/bash
# round 3, experiment 28: frames in flight for the scenes read from HBM (tuned schedule choice)
for w in "config3" "synthetic --prims 1000000"; do for f in 2 4 6 8 12; do
  timeout -k 10 300 python bench.py --workload $w --frames-per-step 12 --frames-in-flight $f --steps 10 --no-cpu-baseline 2> gpurun_out/r3_28.err > gpurun_out/r3_28.json || { echo "$w $f failed"; tail -5 gpurun_out/r3_28.err; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/r3_28.json')); print('$w', 'in flight $f', round(d['value'],1), 'Mrays/s', round(d['ms_per_frame'],4), 'ms/frame |', d['config'].get('schedule').split('(')[0], (d['config'].get('schedule_tuning') or {}).get('best'))"
done; done
