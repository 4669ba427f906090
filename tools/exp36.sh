#!/bin/bash
# compiler-flag lottery for the ray kernels: config 2 frame time, dragon, parity of the golden frames
for lib in libp3d_hip.so libp3d_hip_v1.so libp3d_hip_v2.so libp3d_hip_v3.so libp3d_hip_v4.so; do
  export P3D_LIB=$PWD/u_4a_2s_p3d_raytracer_template2_amd/$lib
  a=$(python3 tools/perf_probe.py mount_low --n 300 2>&1 | grep -E "wavefront lds/lane" | awk '{print $7}')
  b=$(python3 tools/perf_probe.py dragon --tree --n 10 2>&1 | grep -E "tree      hbm/lane" | awk '{print $7}')
  c=$(python3 tools/config4.py 2>&1 | grep "^tile" | awk '{print $3}')
  echo "$lib config2 $a ms  dragon $b ms  config4 $c ms"
done
