#!/bin/bash
# round 3, experiment 22: with the libm powf restatement in, config 4 per schedule under the three register budgets; 1e5 / 1e6 primitives against v0
set -e
O=gpurun_out/r3_22; mkdir -p $O
L=$PWD/u_4a_2s_p3d_raytracer_template2_amd
for occ in 0 5 6; do
  echo "=== P3D_OCC=$occ config4" >> $O/config4.txt
  P3D_OCC=$occ timeout -k 10 300 python tools/config4.py 2>&1 | grep -E "device [0-9.]+ ms" >> $O/config4.txt
done
echo "=== v0 (device-library powf), default budgets" >> $O/config4.txt
P3D_LIB=$L/libp3d_hip_v0.so timeout -k 10 300 python tools/config4.py 2>&1 | grep -E "device [0-9.]+ ms" >> $O/config4.txt
cat $O/config4.txt
for n in 100000 1000000; do for v in _v0 ""; do
  P3D_LIB=$L/libp3d_hip$v.so timeout -k 10 300 python bench.py --workload synthetic --prims $n --no-cpu-baseline 2>/dev/null > $O/s${n}$v.json
  python -c "
import json; d=json.load(open('$O/s${n}$v.json')); print('$n v$v', round(d['value'],1), d['ms_per_step'], d['config'].get('schedule'))"
done; done
