#!/bin/bash
# End-of-round evidence run (GPU box): GPU test suite, profiles (tools/r02_profiles.sh), microbenchmarks, schedule probe,
# synthetic bench lines.  Everything under gpurun_out/r02/.
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?
tail -3 $O/pytest_gpu.txt
if grep -q "Memory access fault" $O/pytest_gpu.txt; then echo "GPU FAULT"; exit 99; fi
[ $rc -ne 0 ] && exit $rc
P3D_PMC_PASSES="1 2 3 4 5 6" bash tools/r02_profiles.sh > $O/profiles.log 2>&1 || { tail -20 $O/profiles.log; exit 1; }
tail -3 $O/profiles.log
timeout -k 10 200 tools/ubench/gather_rate > $O/gather_rate_ubench.txt 2>&1
timeout -k 10 600 python tools/schedule_probe.py > $O/schedule_probe.txt 2>&1; grep -E "dragon|synthetic|MISMATCH" $O/schedule_probe.txt
python bench.py --workload synthetic --prims 1000000 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_syn6.json 2> $O/bench_syn6.err; tail -c 300 $O/bench_syn6.json; echo
python bench.py --workload synthetic --prims 10000000 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_syn7.json 2> $O/bench_syn7.err; tail -c 300 $O/bench_syn7.json; echo
if grep -q "Memory access fault" $O/*.txt $O/*.err $O/*.log; then echo "GPU FAULT"; exit 99; fi
echo final done
