#!/bin/bash
# round 3, experiment 31: config 4 with two frames in flight
for a in "--frames-per-step 1 --frames-in-flight 1" "--frames-per-step 2 --frames-in-flight 2" "--frames-per-step 4 --frames-in-flight 4" "--frames-per-step 2 --frames-in-flight 2 --schedule wavefront" "--frames-per-step 1 --frames-in-flight 1 --schedule wavefront"; do
  timeout -k 10 400 python bench.py --workload config4 --steps 6 --warmup 2 --no-cpu-baseline $a 2> gpurun_out/r3_31.err > gpurun_out/r3_31.json || { echo "$a failed"; tail -5 gpurun_out/r3_31.err; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/r3_31.json')); print('$a', round(d['value'],1), 'Mrays/s', round(d['ms_per_frame'],4), 'ms/frame', d['config']['schedule'])"
done
