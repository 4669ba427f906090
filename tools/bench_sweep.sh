#!/bin/bash
# bench.py over graph on/off x frames in flight (GPU box); prints value, ms/frame, graph used, checksum
for g in off on; do for f in 3 6; do
  timeout -k 10 200 python bench.py --graph $g --frames-in-flight $f --no-cpu-baseline --steps 50 2>gpurun_out/sweep.err | \
  python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('graph $g F $f', round(d['value']), round(d['ms_per_frame'], 4), d['config']['hip_graph'], d['config']['frame_checksum'])
"
  tail -n 2 gpurun_out/sweep.err
done; done
