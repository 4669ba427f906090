#!/bin/bash
# round 3, experiment 10: register budgets of the LDS kernels (config 2 wavefront, config 4 tile)
set -e
O=gpurun_out/r3_10; mkdir -p $O
for occ in 0 5 6; do
  echo "=== P3D_OCC=$occ config2" >> $O/occ.txt
  for rep in 1 2; do
  P3D_OCC=$occ timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print({k:d[k] for k in ('value','ms_per_frame','ms_per_frame_latency')}, d['config']['frame_checksum'])" >> $O/occ.txt
  done
  echo "=== P3D_OCC=$occ config4" >> $O/occ.txt
  P3D_OCC=$occ P3D_VERBOSE=1 timeout -k 10 300 python tools/config4.py 2>&1 | grep -E "p3d: tile|device" | sort | uniq >> $O/occ.txt
done
cat $O/occ.txt
