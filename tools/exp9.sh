#!/bin/bash
O=gpurun_out/exp9; mkdir -p $O
for leaf in 2 4 8; do
  echo "== leaf $leaf"
  timeout -k 10 300 python tools/perf_probe.py dragon 1920 1080 --n 10 --leaf $leaf > $O/c3_leaf$leaf.txt 2>&1; grep -h "stats\|tile      hbm/lane\|tree      hbm/lane\|wavefront hbm/lane" $O/c3_leaf$leaf.txt | cut -c1-200
  timeout -k 10 300 python tools/perf_probe.py --synthetic 1000000 --n 10 --leaf $leaf > $O/syn_leaf$leaf.txt 2>&1; grep -h "stats\|wavefront hbm/lane" $O/syn_leaf$leaf.txt | cut -c1-200
done
