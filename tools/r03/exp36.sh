#!/bin/bash
# round 3, experiment 36: 3-instruction exact reciprocal in normalize() and the triangle test
set -e
O=gpurun_out/r3_36; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for i in 1 2; do timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null > $O/c2.json; python -c "
import json; d=json.load(open('$O/c2.json')); print('config2', round(d['value'],1), d['ms_per_frame'], d['ms_per_frame_latency'])"; done
timeout -k 10 300 python bench.py --workload config4 --no-cpu-baseline 2>/dev/null > $O/c4.json; python -c "
import json; d=json.load(open('$O/c4.json')); print('config4', round(d['value'],1), d['ms_per_frame'])"
timeout -k 10 300 python bench.py --workload config3 --no-cpu-baseline 2>/dev/null > $O/c3.json; python -c "
import json; d=json.load(open('$O/c3.json')); print('config3', round(d['value'],1), d['ms_per_frame'], d['ms_per_frame_latency'])"
timeout -k 10 300 python bench.py --workload synthetic --prims 1000000 --no-cpu-baseline 2>/dev/null > $O/s6.json; python -c "
import json; d=json.load(open('$O/s6.json')); print('1e6', round(d['value'],1), d['ms_per_frame'], d['ms_per_frame_latency'])"
