#!/bin/bash
# round 3, experiment 17: what the libm powf restatement costs, variant against variant on ONE box
#   v0 device library powf | v1 restatement, tables in constant memory | v2 tables from LDS (timing only: garbage tables)
#   v3 restatement with every case inline and branchy
set -e
O=gpurun_out/r3_17; mkdir -p $O
L=u_4a_2s_p3d_raytracer_template2_amd
for round in 1 2; do for v in 0 1 2 3; do
  P3D_LIB=$PWD/$L/libp3d_hip_v$v.so timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null > $O/c2_v${v}_$round.json || true
  python -c "
import json; d=json.load(open('$O/c2_v${v}_$round.json')); print('config2 v$v', round(d['value'],1), d.get('frame_matches_reference'))"
done; done
for v in 0 1 2 3; do
  P3D_LIB=$PWD/$L/libp3d_hip_v$v.so timeout -k 10 300 python bench.py --workload config4 --no-cpu-baseline 2>/dev/null > $O/c4_v$v.json || true
  python -c "
import json; d=json.load(open('$O/c4_v$v.json')); print('config4 v$v', round(d['value'],1), d['ms_per_step'])"
  P3D_LIB=$PWD/$L/libp3d_hip_v$v.so timeout -k 10 300 python bench.py --workload config3 --no-cpu-baseline 2>/dev/null > $O/c3_v$v.json || true
  python -c "
import json; d=json.load(open('$O/c3_v$v.json')); print('config3 v$v', round(d['value'],1), d['ms_per_step'])"
done
