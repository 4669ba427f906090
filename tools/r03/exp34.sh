#!/bin/bash
# round 3, experiment 34: how finely to cut a strip's frames (path tracer, 1080p x 256 samples)
for c in "16 16" "32 8" "64 4" "128 2" "256 1" "8 32"; do set -- $c
  P3D_PT_CHUNKS=$1 P3D_PT_MIN_RUN=$2 timeout -k 10 300 python bench.py --workload pathtracer --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null > gpurun_out/r3_34.json
  python -c "
import json; d=json.load(open('gpurun_out/r3_34.json')); print('runs <= $1 of >= $2 frames:', round(d['ms_per_step'],2), 'ms', round(d['value'],1), 'Msamples/s')"
done
