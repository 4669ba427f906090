#!/bin/bash
# round 3, experiment 1: sanity + baselines of this box + per-wave timelines (stamps) of the launches named in VERDICT r02
set -e
O=gpurun_out/r3_01; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python tools/schedule_probe.py > $O/schedule_probe.txt 2>&1 || true
tail -20 $O/schedule_probe.txt
for a in "dragon tree 1" "dragon wavefront 1" "dragon wavefront 2" "mount_low wavefront 1" "mount_low wavefront 2" "mount_low wavefront 3" "mount_low wavefront 4" "synthetic:1000000 wavefront 1" "synthetic:1000000 wavefront 2"; do
  echo "=== $a" >> $O/timelines.txt
  python tools/wave_timeline.py $a >> $O/timelines.txt 2>&1 || echo FAILED >> $O/timelines.txt
done
cat $O/timelines.txt
