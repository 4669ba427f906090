#!/bin/bash
# Round-end evidence on the GPU box: smoke, default bench line, rocprofv3 kernel stats of the same
# command (and with one frame in flight), PMC passes.  Results under gpurun_out/final/.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.json; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.err
rocprofv3 --kernel-trace --stats -d $O/kt_f1 --output-format csv -- python3 $R/bench.py --no-cpu-baseline --frames-in-flight 1 > $O/bench_under_rocprof_f1.json 2> $O/kt_f1.err
cd $R
P3D_PMC_PASSES="1 2 3 4" tools/pmc_collect.sh final --steps 2 --warmup 1 --frames-in-flight 1 --no-cpu-baseline > $O/pmc.log 2>&1
cp $R/gpurun_out/pmc_final/summary.json $O/pmc.json
echo done
