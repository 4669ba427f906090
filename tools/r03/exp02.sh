#!/bin/bash
# round 3, experiment 2: the new GPU tests (BASELINE sizes) + the self-checking bench lines of config 2 and config 3
set -e
O=gpurun_out/r3_02; mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -14 $O/pytest.log
python bench.py > $O/bench_config2.json 2> $O/bench_config2.err || { tail -20 $O/bench_config2.err; exit 1; }
python -c "import json;d=json.load(open('$O/bench_config2.json'));print({k:d[k] for k in ('value','ms_per_frame','ms_per_frame_latency','frame_matches_reference')});print(d['frame_check'])"
python bench.py --workload config3 --steps 20 --warmup 3 > $O/bench_config3.json 2> $O/bench_config3.err || { tail -20 $O/bench_config3.err; exit 1; }
python -c "import json;d=json.load(open('$O/bench_config3.json'));print({k:d[k] for k in ('value','ms_per_frame','ms_per_frame_latency','frame_matches_reference')});print(d['frame_check']);print(d['config']);print(d['cpu_baseline'])"
