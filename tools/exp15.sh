#!/bin/bash
O=gpurun_out/exp15; mkdir -p $O
for rb in 4 8 16 32; do
  export P3D_RESOLVE_BLOCKS=$rb
  echo "== resolve blocks per shard $rb"
  timeout -k 10 300 python tools/perf_probe.py mount_low 1920 1080 --n 100 > $O/c2_$rb.txt 2>&1; grep -h "wavefront lds/lane" $O/c2_$rb.txt
done
export P3D_RESOLVE_BLOCKS=16
timeout -k 10 300 python tools/config4.py > $O/c4.txt 2>&1; tail -n 3 $O/c4.txt | head -2
