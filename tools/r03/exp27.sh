#!/bin/bash
# round 3, experiment 27: p3d_tune_schedule -- the schedule choice measured with the bench's frames in flight
set -e
timeout -k 10 600 python -m pytest tests/test_gpu_large_scenes.py -x -q > gpurun_out/r3_27_pytest.log 2>&1 || { tail -40 gpurun_out/r3_27_pytest.log; exit 1; }
tail -1 gpurun_out/r3_27_pytest.log
for w in "config3" "synthetic --prims 100000" "synthetic --prims 1000000"; do for t in off on on; do
  P3D_VERBOSE=1 timeout -k 10 300 python bench.py --workload $w --tune $t --no-cpu-baseline 2> gpurun_out/r3_27.err > gpurun_out/r3_27.json || { echo "$w $t failed"; tail -5 gpurun_out/r3_27.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/r3_27.json')); print('$w', 'tune $t', round(d['value'],1), 'Mrays/s', round(d['ms_per_frame'],4), 'ms/frame in flight |', d['config'].get('schedule'), d['config'].get('schedule_tuning'))"
done; done
