#!/bin/bash
# round 3, experiment 5: heaviest-tile-first order of the tile schedule, fused last level (config 2), register budgets with sharing
set -e
O=gpurun_out/r3_05; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
P3D_FUSE_LAST=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_scenes.py tests/test_gpu_distribution.py -m gpu -x -q > $O/pytest_fuse.log 2>&1 || { tail -40 $O/pytest_fuse.log; exit 1; }
tail -2 $O/pytest_fuse.log
for v in "P3D_TILE_LPT=1" "P3D_TILE_LPT=0" "P3D_TILE_LPT=1 P3D_SHARE_MIN_IDLE=0" "P3D_OCC=5" "P3D_OCC=5 P3D_SHARE_MIN_IDLE=0" "P3D_OCC=0 P3D_SHARE_MIN_IDLE=16"; do
  echo "=== $v" >> $O/probe.txt
  env $v timeout -k 10 300 python tools/schedule_probe.py 2>&1 | grep -E "dragon|synthetic|MISMATCH" >> $O/probe.txt
done
cat $O/probe.txt
for f in 0 1; do
  echo "=== P3D_FUSE_LAST=$f" >> $O/fuse.txt
  for rep in 1 2; do
  P3D_FUSE_LAST=$f timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print({k:d[k] for k in ('value','ms_per_frame','ms_per_frame_latency')}, d['roofline']['kernel_ms_live'], d['config']['frame_checksum'])" >> $O/fuse.txt
  done
done
for a in "mount_low wavefront 3"; do
  P3D_FUSE_LAST=1 timeout -k 10 120 python tools/wave_timeline.py $a 2>&1 | grep -v "^width\|^$\|amdgpu.ids" >> $O/fuse.txt
done
cat $O/fuse.txt
echo "=== tile timeline dragon, LPT" >> $O/tile_timeline.txt
timeout -k 10 200 python tools/tile_timeline.py dragon 2>&1 | grep -v "^width\|^$\|amdgpu.ids" >> $O/tile_timeline.txt
cat $O/tile_timeline.txt
