#!/bin/bash
O=gpurun_out/exp16; mkdir -p $O
for fs in 1 2 3 4; do
  export P3D_FRAME_STREAMS=$fs
  echo "== frame streams $fs"
  timeout -k 10 300 python tools/perf_probe.py mount_low 1920 1080 --n 100 > $O/c2_$fs.txt 2>&1; grep -h "wavefront lds/lane" $O/c2_$fs.txt
done
