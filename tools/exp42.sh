#!/bin/bash
# leaf size sweep with the flat walk loop
for leaf in 1 2 4 8; do
  a=$(timeout -k 10 300 python tools/perf_probe.py dragon --tree --leaf $leaf --n 10 2>&1 | grep -E "tree      hbm/lane" | awk '{print $7}')
  b=$(timeout -k 10 300 python tools/perf_probe.py --synthetic 1000000 --leaf $leaf --n 5 2>&1 | grep -E "wavefront hbm/lane" | awk '{print $7}')
  echo "leaf_max $leaf: dragon tree $a ms   1e6 wavefront $b ms"
done
