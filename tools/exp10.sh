#!/bin/bash
set -o pipefail
O=gpurun_out/exp10; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc $rc" >> $O/pytest.txt
grep -v "^  File\|^Extension" $O/pytest.txt | tail -n 12
if grep -q "Memory access fault" $O/pytest.txt; then exit 9; fi
if [ $rc -ne 0 ]; then exit $rc; fi
for s in default tile; do
  python bench.py --no-cpu-baseline --schedule $s --steps 30 > $O/bench_$s.json 2>$O/bench_$s.err; python3 -c "
import json;d=json.loads(open('$O/bench_$s.json').read().strip().splitlines()[-1]);print('$s', d['value'], d['ms_per_frame'], d['ms_per_frame_latency'], d['config']['schedule'])"
done
python bench.py --no-cpu-baseline --frames-in-flight 4 --steps 30 > $O/bench_f4.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('$O/bench_f4.json').read().strip().splitlines()[-1]);print('f4', d['value'], d['ms_per_frame'])"
python bench.py --no-cpu-baseline --frames-in-flight 2 --steps 30 > $O/bench_f2.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('$O/bench_f2.json').read().strip().splitlines()[-1]);print('f2', d['value'], d['ms_per_frame'])"
