#!/bin/bash
# round 3, experiment 29: where kernels that read the scene from HBM get powf's tables from: LDS staged per workgroup (ldst) or vector loads from the blob (glob)
L=$PWD/u_4a_2s_p3d_raytracer_template2_amd
for w in "config3" "synthetic --prims 100000" "synthetic --prims 1000000"; do for v in ldst glob ldst glob; do
  P3D_LIB=$L/libp3d_hip_$v.so timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline 2> gpurun_out/r3_29.err > gpurun_out/r3_29.json || { echo "$w $v failed"; tail -5 gpurun_out/r3_29.err; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/r3_29.json')); print('$w', '$v', round(d['value'],1), 'Mrays/s', round(d['ms_per_frame'],4), 'ms/frame in flight', round(d['ms_per_frame_latency'],4), 'alone |', (d['config'].get('schedule_tuning') or {}).get('best'))"
done; done
