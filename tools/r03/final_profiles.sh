#!/bin/bash
# Round-3 evidence on the GPU box (run through gpurun from the repo root): self-checking bench lines, rocprofv3 kernel
# stats and PMC summaries of config 2 (bench.py, one frame in flight), config 3 (dragon 1080p), config 4 (4096^2 d6 spp2)
# and the 1e6-primitive scaling scene.  Results under gpurun_out/r03/; the summaries worth judging are copied to profiles/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
python bench.py > $O/bench_config2.json 2> $O/bench_config2.err; tail -c 300 $O/bench_config2.json; echo
python bench.py --workload config3 --steps 20 --warmup 3 > $O/bench_config3.json 2> $O/bench_config3.err; tail -c 300 $O/bench_config3.json; echo
python bench.py --workload config4 --steps 10 --warmup 2 > $O/bench_config4.json 2> $O/bench_config4.err; tail -c 300 $O/bench_config4.json; echo
python bench.py --workload synthetic --prims 1000000 --steps 10 --warmup 2 > $O/bench_synthetic_1e6.json 2> $O/bench_synthetic_1e6.err; tail -c 300 $O/bench_synthetic_1e6.json; echo
python bench.py --workload pathtracer --steps 3 --warmup 1 > $O/bench_pathtracer.json 2> $O/bench_pathtracer.err; tail -c 300 $O/bench_pathtracer.json; echo
cd /tmp && export TMPDIR=/tmp
kt() {   # kt TAG program args...
  local tag=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/kt_$tag --output-format csv -- python3 "$@" > $O/kt_$tag.log 2>&1
  cp $(ls $O/kt_$tag/*/*kernel_stats.csv | head -1) $O/${tag}_kernel_stats.csv
}
pmc() {  # pmc TAG script args...
  local tag=$1; shift
  local script=$1; shift
  (cd $R && P3D_PMC_SCRIPT=$script P3D_PMC_PASSES="${P3D_PMC_PASSES:-1 2 3 4 5 6}" tools/pmc_collect.sh r03_$tag "$@" > $O/pmc_$tag.log 2>&1)
  cp $R/gpurun_out/pmc_r03_$tag/summary.json $O/${tag}_pmc.json
}
kt config2 $R/bench.py --no-cpu-baseline --frames-in-flight 1 --steps 10 --warmup 3
kt config2_default $R/bench.py --no-cpu-baseline
pmc config2 bench.py --no-cpu-baseline --frames-in-flight 1 --steps 2 --warmup 2
# the profile of a scene read from HBM shows the schedule (and walk) the bench line of this run adopted (p3d_tune_schedule)
sched_of() { python3 -c "
import json,sys
c=json.load(open(sys.argv[1]))['config']; t=c.get('schedule_tuning')
s=c['schedule'].split()[0]
print(s + ('_private' if t and 'private' in t['best'] else '') if t else 'default')" $1; }
S3=$(sched_of $O/bench_config3.json); echo "config3 profiled as: $S3"
kt config3 $R/tools/render_frames.py dragon $S3 40
pmc config3 tools/render_frames.py dragon $S3 30
kt config4 $R/tools/config4.py
pmc config4 tools/config4.py
S6=$(sched_of $O/bench_synthetic_1e6.json); echo "1e6 primitives profiled as: $S6"
kt synthetic_1000000 $R/tools/render_frames.py 1000000 $S6 30
pmc synthetic_1000000 tools/render_frames.py 1000000 $S6 24
kt pathtracer $R/bench.py --workload pathtracer --no-cpu-baseline --steps 2 --warmup 1
pmc pathtracer bench.py --workload pathtracer --no-cpu-baseline --steps 1 --warmup 1
ls -la $O | head -60
echo done
# frames in flight of config 2 (the default is 4)
for f in 2 3 4 5 6 8; do
  echo "frames_in_flight $f: $(python $R/bench.py --no-cpu-baseline --frames-in-flight $f --steps 30 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['value']),'Mrays/s', round(d['ms_per_frame'],5),'ms/frame')")" >> $O/frames_in_flight.txt
done
cat $O/frames_in_flight.txt
