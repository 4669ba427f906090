#!/bin/bash
# the bench lines of the round, after the profiles they cite are in profiles/ (GPU box)
O=gpurun_out/r02b; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python bench.py > $O/bench_config2.json 2> $O/bench_config2.err || { tail -5 $O/bench_config2.err; exit 1; }
python bench.py --workload config4 --steps 10 --warmup 2 > $O/bench_config4.json 2> $O/bench_config4.err || { tail -5 $O/bench_config4.err; exit 1; }
python bench.py --workload synthetic --prims 1000000 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_syn6.json 2> $O/bench_syn6.err
python bench.py --workload synthetic --prims 10000000 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_syn7.json 2> $O/bench_syn7.err
python bench.py --workload pathtracer --steps 3 --warmup 1 > $O/bench_pt.json 2> $O/bench_pt.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt_default --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/kt_default.log 2>&1
cd $GRAFT_REPO_ROOT; cp $(ls $O/kt_default/*/*kernel_stats.csv | head -1) $O/config2_default_kernel_stats.csv
python - <<PY
import json
for f in ("bench_config2","bench_config4","bench_syn6","bench_syn7","bench_pt"):
    t=open("$O/%s.json"%f).read(); d=json.loads(t[t.index('{"metric"'):])
    r=d.get("roofline") or {}
    print(f, "%.0f %s" % (d["value"], d["unit"]), "ms/frame", d.get("ms_per_frame"), "latency", d.get("ms_per_frame_latency"), "F", d["config"].get("frames_in_flight"), "| roofline", r.get("kernel"), "frac", r.get("frac"), "stale", r.get("stale"), "| cpu", d["cpu_baseline"]["value"] if d.get("cpu_baseline") else None)
PY
if grep -q "Memory access fault" $O/*.txt $O/*.err; then echo "GPU FAULT"; exit 99; fi
