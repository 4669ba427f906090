#!/bin/bash
# round 3, experiment 23: HBM-scene kernels stage powf's tables in LDS too: parity, soak, cost against v0 (device-library powf)
set -e
O=gpurun_out/r3_23; mkdir -p $O
L=$PWD/u_4a_2s_p3d_raytracer_template2_amd
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
timeout -k 10 300 python tools/r03/rgb_delta.py 2>&1 | grep -E "TOTAL" 
timeout -k 10 600 python tools/soak_shared.py 200 30 2>&1 | grep -v "^width\|^$\|amdgpu" | tail -2
for round in 1 2; do for v in _v0 ""; do
  P3D_LIB=$L/libp3d_hip$v.so timeout -k 10 300 python bench.py --workload config3 --no-cpu-baseline 2>/dev/null > $O/c3$v.json
  python -c "
import json; d=json.load(open('$O/c3$v.json')); print('config3 v$v', round(d['value'],1), d['ms_per_step'])"
done; done
for n in 100000 1000000; do for v in _v0 ""; do
  P3D_LIB=$L/libp3d_hip$v.so timeout -k 10 300 python bench.py --workload synthetic --prims $n --no-cpu-baseline 2>/dev/null > $O/s${n}$v.json
  python -c "
import json; d=json.load(open('$O/s${n}$v.json')); print('$n v$v', round(d['value'],1), d['ms_per_step'], d['config'].get('schedule'))"
done; done
for v in _v0 ""; do
  P3D_LIB=$L/libp3d_hip$v.so timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null > $O/c2$v.json
  python -c "
import json; d=json.load(open('$O/c2$v.json')); print('config2 v$v', round(d['value'],1))"
done
