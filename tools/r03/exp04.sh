#!/bin/bash
# round 3, experiment 4: level-1 tiles per workgroup (config 2); per-tile durations of the tile schedule + LPT bound
set -e
O=gpurun_out/r3_04; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for t in 1 2 3 4; do
  echo "=== P3D_PRIMARY_TILES=$t" >> $O/tiles.txt
  P3D_PRIMARY_TILES=$t timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print({k:d[k] for k in ('value','ms_per_frame','ms_per_frame_latency')}, d['roofline']['kernel_ms_live'], d['config']['frame_checksum'])" >> $O/tiles.txt
  P3D_PRIMARY_TILES=$t timeout -k 10 120 python tools/wave_timeline.py mount_low wavefront 1 2>&1 | grep -E "frame|span|peak|lifetime" >> $O/tiles.txt
done
cat $O/tiles.txt
for v in "P3D_SHARE_MIN_IDLE=0" "P3D_SHARE_MIN_IDLE=16"; do
  for sc in dragon synthetic:1000000; do
    echo "=== $v $sc" >> $O/tile_timeline.txt
    env $v timeout -k 10 200 python tools/tile_timeline.py $sc 2>&1 | grep -v "^width\|^$\|amdgpu.ids" >> $O/tile_timeline.txt || echo FAILED >> $O/tile_timeline.txt
  done
done
cat $O/tile_timeline.txt
for a in "dragon wavefront 1" "dragon wavefront 2" "dragon wavefront 3"; do
  echo "=== share16 $a" >> $O/timelines.txt
  timeout -k 10 120 python tools/wave_timeline.py $a 2>&1 | grep -v "^width\|^$\|amdgpu.ids" >> $O/timelines.txt || echo FAILED >> $O/timelines.txt
done
cat $O/timelines.txt
