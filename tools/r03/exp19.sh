#!/bin/bash
# round 3, experiment 19: per-kernel durations of config 2 (one frame in flight), device-library powf against the restatement
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_19; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in _v0 ""; do
  export P3D_LIB=$R/u_4a_2s_p3d_raytracer_template2_amd/libp3d_hip$v.so
  rocprofv3 --kernel-trace --stats -d $O/kt$v --output-format csv -- python3 $R/bench.py --no-cpu-baseline --frames-in-flight 1 --steps 10 --warmup 3 > $O/kt$v.log 2>&1
  cp $(ls $O/kt$v/*/*kernel_stats.csv | head -1) $O/config2${v}_kernel_stats.csv
  rm -rf $O/kt$v
  echo "== lib$v"; cut -d, -f1-4 $O/config2${v}_kernel_stats.csv | head -8
done
