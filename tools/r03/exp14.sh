#!/bin/bash
# round 3, experiment 14: tile schedule spreads a level's remainder over its four waves (shared walks)
set -e
O=gpurun_out/r3_14; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 600 python tools/soak_shared.py 300 30 2>&1 | grep -v "^width\|^$\|amdgpu" | tail -2
timeout -k 10 300 python tools/schedule_probe.py 2>&1 | grep -E "dragon|synthetic|MISMATCH" > $O/probe.txt; cat $O/probe.txt
timeout -k 10 200 python tools/tile_timeline.py dragon 2>&1 | grep -v "^width\|^$\|amdgpu.ids" > $O/tile_timeline.txt; cat $O/tile_timeline.txt
for sc in dragon 100000; do P3D_VERBOSE=1 timeout -k 10 200 python tools/render_frames.py $sc default 16 2>&1 | grep -E "measured choice" >> $O/pick.txt; done; cat $O/pick.txt
