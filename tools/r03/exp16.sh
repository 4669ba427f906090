#!/bin/bash
# round 3, experiment 16: straight-line powf for the shading's argument domain: parity again, cost
set -e
O=gpurun_out/r3_16; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_powf.py -x -q > $O/powf.log 2>&1 || { tail -30 $O/powf.log; exit 1; }
tail -1 $O/powf.log
timeout -k 10 300 python tools/r03/rgb_delta.py 2>&1 | grep -E "TOTAL|max" > $O/rgb_delta.txt; tail -1 $O/rgb_delta.txt
for i in 1 2 3; do timeout -k 10 300 python bench.py 2>/dev/null > $O/bench_config2_$i.json; python -c "
import json,sys; d=json.load(open('$O/bench_config2_$i.json')); print('config2', d['value'], d['ms_per_step'], d.get('frame_matches_reference'))"; done
timeout -k 10 400 python bench.py --workload config3 2>/dev/null > $O/bench_config3.json; python -c "
import json,sys; d=json.load(open('$O/bench_config3.json')); print('config3', d['value'], d['ms_per_step'], d.get('frame_matches_reference'))"
timeout -k 10 400 python bench.py --workload config4 2>/dev/null > $O/bench_config4.json; python -c "
import json,sys; d=json.load(open('$O/bench_config4.json')); print('config4', d['value'], d['ms_per_step'], d.get('frame_matches_reference'))"
timeout -k 10 400 python bench.py --workload synthetic --prims 100000 2>/dev/null > $O/bench_s5.json; python -c "
import json,sys; d=json.load(open('$O/bench_s5.json')); print('1e5', d['value'], d['ms_per_step'], d.get('frame_matches_reference'))"
timeout -k 10 400 python bench.py --workload synthetic --prims 1000000 2>/dev/null > $O/bench_s6.json; python -c "
import json,sys; d=json.load(open('$O/bench_s6.json')); print('1e6', d['value'], d['ms_per_step'], d.get('frame_matches_reference'))"
