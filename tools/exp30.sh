#!/bin/bash
O=gpurun_out/exp30; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?
tail -3 $O/pytest.txt
if grep -q "Memory access fault" $O/pytest.txt; then echo "GPU FAULT"; exit 99; fi
[ $rc -ne 0 ] && exit $rc
echo "--- fused resolve off"; P3D_FUSED_RESOLVE_PX=0 timeout -k 10 300 python tools/shard_probe.py wavefront 2>&1 | grep -E " 4 in flight|12 in flight"
echo "--- fused resolve for shards <= 8192 px"; timeout -k 10 300 python tools/shard_probe.py wavefront 2>&1 | grep -E " 4 in flight|12 in flight"
echo "--- fused resolve always"; P3D_FUSED_RESOLVE_PX=100000000 timeout -k 10 300 python tools/shard_probe.py wavefront 2>&1 | grep -E " 4 in flight|12 in flight"
