#!/bin/bash
# round 3, experiment 3: work-sharing walk (scenes read from HBM): parity suite, then A/B against private walks
set -e
O=gpurun_out/r3_03; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for mi in 0 8 16 32; do
  echo "=== P3D_SHARE_MIN_IDLE=$mi" >> $O/probe.txt
  P3D_SHARE_MIN_IDLE=$mi timeout -k 10 300 python tools/schedule_probe.py 2>&1 | grep -E "dragon|synthetic|MISMATCH" >> $O/probe.txt
done
cat $O/probe.txt
for a in "dragon tree 1" "synthetic:1000000 wavefront 1" "synthetic:1000000 wavefront 2"; do
  echo "=== $a" >> $O/timelines.txt
  timeout -k 10 120 python tools/wave_timeline.py $a 2>&1 | grep -v "^width\|^$\|amdgpu.ids" >> $O/timelines.txt || echo FAILED >> $O/timelines.txt
done
cat $O/timelines.txt
# level-1 workgroup size of LDS scenes (config 2)
for w in 4 8 16; do
  echo "=== P3D_PRIMARY_WG_WAVES=$w" >> $O/wg.txt
  P3D_PRIMARY_WG_WAVES=$w timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print({k:d[k] for k in ('value','ms_per_frame','ms_per_frame_latency')}, d['roofline']['kernel_ms_live'], d['config']['frame_checksum'])" >> $O/wg.txt
  P3D_PRIMARY_WG_WAVES=$w timeout -k 10 120 python tools/wave_timeline.py mount_low wavefront 1 2>&1 | grep -E "frame|span|peak|lifetime" >> $O/wg.txt
done
cat $O/wg.txt
P3D_VERBOSE=1 timeout -k 10 300 python tools/config4.py 2>&1 | grep -E "p3d:|device" | sort | uniq -c | head -20
for v in "P3D_TRI_STRIDE=64" "P3D_NODE_ORDER=treelet" "P3D_NODE_ORDER=treelet8" "P3D_TRI_STRIDE=64 P3D_SHARE_MIN_IDLE=0" "P3D_NODE_ORDER=treelet P3D_SHARE_MIN_IDLE=0"; do
  echo "=== $v" >> $O/levers.txt
  env $v timeout -k 10 300 python tools/schedule_probe.py 2>&1 | grep -E "dragon|synthetic|MISMATCH" >> $O/levers.txt
done
cat $O/levers.txt
