#!/bin/bash
# round-2 experiment 1: VALU micro-benchmark, GPU tests, SLP on/off A/B on config 2 / dragon / config 4
set -o pipefail
O=gpurun_out/exp1; mkdir -p $O
PKG=u_4a_2s_p3d_raytracer_template2_amd
tools/ubench/valu_rate > $O/valu_rate.txt 2>&1
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
for v in base noslp; do
  export P3D_LIB=$PWD/$PKG/libp3d_hip_$v.so
  timeout -k 10 300 python tools/perf_probe.py mount_low 1920 1080 --n 100 > $O/probe_c2_$v.txt 2>&1
  timeout -k 10 300 python tools/perf_probe.py dragon 1920 1080 --n 10 > $O/probe_c3_$v.txt 2>&1
  timeout -k 10 300 python bench.py > $O/bench_$v.txt 2>&1
  timeout -k 10 300 python tools/config4.py > $O/c4_$v.txt 2>&1
done
tail -n 30 $O/valu_rate.txt; tail -n 3 $O/pytest.txt; grep -h "wavefront lds/packet" $O/probe_c2_*.txt; tail -n 2 $O/c4_*.txt
