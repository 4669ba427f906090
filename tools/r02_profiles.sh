#!/bin/bash
# Round-2 evidence on the GPU box (run through gpurun from the repo root): bench line, rocprofv3 kernel stats and
# PMC summaries of config 2 (bench.py, one frame in flight), config 3 (dragon 1080p) and config 4 (4096^2 d6 spp2).
# Results under gpurun_out/r02/; the summaries worth judging are copied to profiles/ by hand afterwards.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
python bench.py > $O/bench_config2.json 2> $O/bench_config2.err; tail -c 400 $O/bench_config2.json; echo
python bench.py --workload config4 --steps 10 --warmup 2 > $O/bench_config4.json 2> $O/bench_config4.err; tail -c 300 $O/bench_config4.json; echo
cd /tmp && export TMPDIR=/tmp
kt() {   # kt TAG program args...
  local tag=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/kt_$tag --output-format csv -- python3 "$@" > $O/kt_$tag.log 2>&1
  cp $(ls $O/kt_$tag/*/*kernel_stats.csv | head -1) $O/${tag}_kernel_stats.csv
}
pmc() {  # pmc TAG program args...
  local tag=$1; shift
  local script=$1; shift
  (cd $R && P3D_PMC_SCRIPT=$script P3D_PMC_PASSES="${P3D_PMC_PASSES:-1 2 3 4}" tools/pmc_collect.sh r02_$tag "$@" > $O/pmc_$tag.log 2>&1)
  cp $R/gpurun_out/pmc_r02_$tag/summary.json $O/${tag}_pmc.json
}
kt config2 $R/bench.py --no-cpu-baseline --frames-in-flight 1 --steps 10 --warmup 3
kt config2_default $R/bench.py --no-cpu-baseline
pmc config2 bench.py --no-cpu-baseline --frames-in-flight 1 --steps 2 --warmup 2
kt config3 $R/tools/render_frames.py dragon default 14
pmc config3 tools/render_frames.py dragon default 10
kt config4 $R/tools/config4.py
pmc config4 tools/config4.py
ls -la $O | head -40
echo done
