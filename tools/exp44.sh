#!/bin/bash
# scenes read from HBM, flat walk loop: 4-byte reference-only stack slots (libp3d_hip.so) vs 6-byte slots with pop-time pruning (libp3d_hip_x.so)
for lib in libp3d_hip.so libp3d_hip_x.so; do
  export P3D_LIB=$PWD/u_4a_2s_p3d_raytracer_template2_amd/$lib
  a=$(python3 tools/perf_probe.py dragon --tree --n 10 2>&1 | grep -E "tree      hbm/lane" | awk '{print $7}')
  b=$(python3 tools/perf_probe.py --synthetic 1000000 --n 5 2>&1 | grep -E "wavefront hbm/lane" | awk '{print $7}')
  echo "$lib: dragon tree $a ms   1e6 wavefront $b ms"
done
