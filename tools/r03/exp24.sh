#!/bin/bash
# round 3, experiment 24: the six candidates' measured times at 1e5 / 1e6 primitives and the dragon, v0 against the current library
L=$PWD/u_4a_2s_p3d_raytracer_template2_amd
for sc in 100000 1000000 dragon; do for v in _v0 ""; do
  echo "== $sc lib$v"
  P3D_LIB=$L/libp3d_hip$v.so P3D_VERBOSE=1 timeout -k 10 200 python tools/render_frames.py $sc default 24 2>&1 | grep -E "measured choice|candidate|ms" | head -12
done; done
