#!/bin/bash
# round 3, experiment 8: persistent grids larger than the occupancy query says (it counts 256 VGPRs per SIMD lane; gfx950 has 512)
set -e
O=gpurun_out/r3_08; mkdir -p $O
for sc in 1.0 1.6 2.0; do
  echo "=== P3D_OCC_SCALE=$sc config2" >> $O/occ.txt
  for rep in 1 2; do
  P3D_OCC_SCALE=$sc timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print({k:d[k] for k in ('value','ms_per_frame','ms_per_frame_latency')}, d['config']['frame_checksum'])" >> $O/occ.txt
  done
  P3D_OCC_SCALE=$sc timeout -k 10 120 python tools/wave_timeline.py mount_low wavefront 2 2>&1 | grep -E "frame|span|peak|lifetime|batches|last-start" >> $O/occ.txt
  echo "=== P3D_OCC_SCALE=$sc config4" >> $O/occ.txt
  P3D_OCC_SCALE=$sc P3D_VERBOSE=1 timeout -k 10 300 python tools/config4.py 2>&1 | grep -E "p3d: tile|device" | sort | uniq >> $O/occ.txt
done
cat $O/occ.txt
