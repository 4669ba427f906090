#!/bin/bash
# round 3, experiment 15: the host libm's powf restated on the device (csrc/p3d_powf.h): bit parity of rgb32f, cost
set -e
O=gpurun_out/r3_15; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_powf.py -x -q > $O/powf.log 2>&1 || { tail -30 $O/powf.log; exit 1; }
tail -2 $O/powf.log
timeout -k 10 300 python tools/r03/rgb_delta.py 2>&1 | grep -v "^width\|^$\|amdgpu.ids\|^from\|^at\|^up\|^angle\|^hither\|^res" > $O/rgb_delta.txt; tail -45 $O/rgb_delta.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for i in 1 2; do timeout -k 10 300 python bench.py 2>/dev/null > $O/bench_config2_$i.json; python -c "
import json,sys; d=json.load(open('$O/bench_config2_$i.json')); print('config2', d['value'], d['ms_per_step'], d.get('frame_matches_reference'), d.get('frame_check'))"; done
timeout -k 10 400 python bench.py --workload config3 2>/dev/null > $O/bench_config3.json; python -c "
import json,sys; d=json.load(open('$O/bench_config3.json')); print('config3', d['value'], d['ms_per_step'], d.get('frame_matches_reference'), d.get('frame_check'))"
timeout -k 10 400 python bench.py --workload config4 2>/dev/null > $O/bench_config4.json; python -c "
import json,sys; d=json.load(open('$O/bench_config4.json')); print('config4', d['value'], d['ms_per_step'], d.get('frame_matches_reference'), d.get('frame_check'))"
