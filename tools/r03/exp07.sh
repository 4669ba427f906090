#!/bin/bash
# round 3, experiment 7: tile height of the tile schedule (scenes read from HBM), never-stealing shared kernels, timelines of the 1e6 scene
set -e
O=gpurun_out/r3_07; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_scenes.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
P3D_TILE_H=8 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_scenes.py tests/test_gpu_random_scenes.py -m gpu -x -q > $O/pytest_h8.log 2>&1 || { tail -40 $O/pytest_h8.log; exit 1; }
tail -2 $O/pytest_h8.log
for v in "P3D_TILE_H=16" "P3D_TILE_H=8" "P3D_TILE_H=4" "P3D_SHARE_MIN_IDLE=64" "P3D_SHARE_MIN_IDLE=48"; do
  echo "=== $v" >> $O/probe.txt
  env $v timeout -k 10 300 python tools/schedule_probe.py 2>&1 | grep -E "dragon|synthetic|MISMATCH" >> $O/probe.txt
done
cat $O/probe.txt
for a in "synthetic:1000000 wavefront 1" "synthetic:1000000 wavefront 2" "synthetic:1000000 wavefront 3" "synthetic:1000000 wavefront 4"; do
  echo "=== private, LPT: $a" >> $O/timelines.txt
  P3D_SHARE_MIN_IDLE=0 timeout -k 10 120 python tools/wave_timeline.py $a 2>&1 | grep -v "^width\|^$\|amdgpu.ids" >> $O/timelines.txt || echo FAILED >> $O/timelines.txt
done
cat $O/timelines.txt
echo "=== tile timeline dragon h8" >> $O/tile_timeline.txt
P3D_TILE_H=8 timeout -k 10 200 python tools/tile_timeline.py dragon 2>&1 | grep -v "^width\|^$\|amdgpu.ids" >> $O/tile_timeline.txt
cat $O/tile_timeline.txt
