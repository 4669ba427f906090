#!/bin/bash
# the bench lines of the round, after the profiles they cite are in profiles/ (GPU box)
O=gpurun_out/r02b; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python bench.py > $O/bench_config2.json 2> $O/bench_config2.err || { tail -5 $O/bench_config2.err; exit 1; }
python bench.py --workload config4 --steps 10 --warmup 2 > $O/bench_config4.json 2> $O/bench_config4.err || { tail -5 $O/bench_config4.err; exit 1; }
python - <<PY
import json
for f in ("bench_config2","bench_config4"):
    t=open("$O/%s.json"%f).read(); d=json.loads(t[t.index('{"metric"'):])
    r=d["roofline"]
    print(f, "%.0f %s" % (d["value"], d["unit"]), "ms/frame", round(d["ms_per_frame"],4), "latency", round(d["ms_per_frame_latency"],4), "| roofline", r["kernel"], r["bound"], "frac", r["frac"], "stale", r["stale"], "| cpu", d["cpu_baseline"]["value"] if d.get("cpu_baseline") else None)
PY
if grep -q "Memory access fault" $O/*.txt $O/*.err; then echo "GPU FAULT"; exit 99; fi
