#!/bin/bash
# round 3, experiment 32: heaviest-tile-first for the tile kernel of scenes served from LDS too (config 4); P3D_TILE_LPT=0 = natural order
for lpt in 0 1 0 1; do for a in "--frames-per-step 1 --frames-in-flight 1" "--frames-per-step 2 --frames-in-flight 2"; do
  P3D_TILE_LPT=$lpt timeout -k 10 400 python bench.py --workload config4 --steps 6 --warmup 2 --no-cpu-baseline $a 2> gpurun_out/r3_32.err > gpurun_out/r3_32.json || { echo "$a failed"; tail -5 gpurun_out/r3_32.err; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/r3_32.json')); print('LPT=$lpt', '$a', round(d['value'],1), 'Mrays/s', round(d['ms_per_frame'],4), 'ms/frame', round(d['ms_per_frame_latency'],4), 'alone')"
done; done
