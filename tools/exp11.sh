#!/bin/bash
O=gpurun_out/exp11; mkdir -p $O
PKG=$PWD/u_4a_2s_p3d_raytracer_template2_amd
R=$PWD
for v in "" _occ7 _occ8; do
  export P3D_LIB=$PKG/libp3d_hip$v.so
  echo "== lib$v"
  timeout -k 10 300 python tools/perf_probe.py mount_low 1920 1080 --n 100 > $O/probe_c2$v.txt 2>&1; grep -h "wavefront lds" $O/probe_c2$v.txt
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d $O/kt$v --output-format csv -- python3 $R/tools/render_frames.py mount_low wavefront 8 > $O/kt$v.log 2>&1)
  python3 tools/kt_summary.py $O/kt$v wf_primary
done
