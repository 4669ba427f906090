#!/bin/bash
# round 3, experiment 6: measured choice over schedule x shared/private walks; heaviest-first order for tree / wavefront level 1
set -e
O=gpurun_out/r3_06; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for v in "P3D_TILE_LPT=1" "P3D_TILE_LPT=0" "P3D_TILE_LPT=1 P3D_SHARE_MIN_IDLE=0"; do
  echo "=== $v" >> $O/probe.txt
  env $v timeout -k 10 300 python tools/schedule_probe.py 2>&1 | grep -E "dragon|synthetic|MISMATCH" >> $O/probe.txt
done
cat $O/probe.txt
for sc in dragon 100000 1000000; do
  echo "=== default choice $sc" >> $O/pick.txt
  P3D_VERBOSE=1 timeout -k 10 200 python tools/render_frames.py $sc default 16 2>&1 | grep -E "measured choice|^wavefront|^tree|^tile" >> $O/pick.txt
done
cat $O/pick.txt
