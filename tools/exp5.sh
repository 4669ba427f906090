#!/bin/bash
# ubench (SALU / compare-select / exec / LDS) + occupancy-floor A/B on config 2 and config 4
set -o pipefail
O=gpurun_out/exp5; mkdir -p $O
PKG=$PWD/u_4a_2s_p3d_raytracer_template2_amd
tools/ubench/valu_rate > $O/valu_rate.txt 2>&1; grep -E "s_add|cmp|readfirst|ds_read|cndmask" $O/valu_rate.txt
for v in "" _occ7 _occ8; do
  export P3D_LIB=$PKG/libp3d_hip$v.so
  echo "== lib$v"
  timeout -k 10 300 python tools/perf_probe.py mount_low 1920 1080 --n 100 > $O/probe_c2$v.txt 2>&1; grep -h "wavefront lds\|tile      lds" $O/probe_c2$v.txt
  timeout -k 10 300 python tools/config4.py > $O/c4$v.txt 2>&1; tail -n 3 $O/c4$v.txt | head -2
done
