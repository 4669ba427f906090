#!/bin/bash
# round 3, experiment 35: register budget of the kernels of scenes read from HBM, with the frames in flight (tuned schedule)
for w in "config3" "synthetic --prims 1000000"; do for occ in 6 5 0; do
  P3D_OCC=$occ timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline 2>/dev/null > gpurun_out/r3_35.json || { echo "$w $occ failed"; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/r3_35.json')); print('$w', 'P3D_OCC=$occ', round(d['value'],1), 'Mrays/s', round(d['ms_per_frame'],4), 'ms/frame in flight |', (d['config'].get('schedule_tuning') or {}).get('best'))"
done; done
