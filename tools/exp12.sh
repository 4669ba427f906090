#!/bin/bash
O=gpurun_out/exp12; mkdir -p $O
PKG=$PWD/u_4a_2s_p3d_raytracer_template2_amd
for v in "" _refstack; do
  export P3D_LIB=$PKG/libp3d_hip$v.so
  echo "== lib$v"
  timeout -k 10 300 python tools/perf_probe.py dragon 1920 1080 --n 10 > $O/c3$v.txt 2>&1; grep -h "hbm/lane" $O/c3$v.txt
  timeout -k 10 300 python tools/perf_probe.py --synthetic 1000000 --n 10 > $O/syn$v.txt 2>&1; grep -h "hbm/lane" $O/syn$v.txt
done
