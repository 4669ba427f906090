#!/bin/bash
# round 3, experiment 18: powf tables at the head of the scene blob, range check behind a wave-level branch, whole-wave skip when every base is +0
set -e
O=gpurun_out/r3_18; mkdir -p $O
L=u_4a_2s_p3d_raytracer_template2_amd
timeout -k 10 300 python -m pytest tests/test_gpu_powf.py -x -q > $O/powf.log 2>&1 || { tail -30 $O/powf.log; exit 1; }
tail -1 $O/powf.log
timeout -k 10 300 python tools/r03/rgb_delta.py 2>&1 | grep -E "TOTAL|max" > $O/rgb_delta.txt; tail -1 $O/rgb_delta.txt
for round in 1 2; do for v in _v0 ""; do
  P3D_LIB=$PWD/$L/libp3d_hip$v.so timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null > $O/c2_v${v}_$round.json || true
  python -c "
import json; d=json.load(open('$O/c2_v${v}_$round.json')); print('config2 v$v', round(d['value'],1))"
done; done
for v in _v0 ""; do
  P3D_LIB=$PWD/$L/libp3d_hip$v.so timeout -k 10 300 python bench.py --workload config4 --no-cpu-baseline 2>/dev/null > $O/c4_v$v.json || true
  python -c "
import json; d=json.load(open('$O/c4_v$v.json')); print('config4 v$v', round(d['value'],1), d['ms_per_step'])"
  P3D_LIB=$PWD/$L/libp3d_hip$v.so timeout -k 10 300 python bench.py --workload config3 --no-cpu-baseline 2>/dev/null > $O/c3_v$v.json || true
  python -c "
import json; d=json.load(open('$O/c3_v$v.json')); print('config3 v$v', round(d['value'],1), d['ms_per_step'])"
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
