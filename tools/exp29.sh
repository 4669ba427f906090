#!/bin/bash
# bench.py: frames in flight 3 vs 4 (two runs each, alternating)
O=gpurun_out/exp29; mkdir -p $O
for r in 1 2; do for f in 3 4; do
python bench.py --no-cpu-baseline --frames-in-flight $f > $O/f${f}_$r.json 2> $O/f${f}_$r.err || exit 1
python - <<PY
import json
t=open("$O/f${f}_$r.json").read(); d=json.loads(t[t.index('{"metric"'):])
print("F=$f run $r: %.0f Mrays/s  %.4f ms/frame" % (d["value"], d["ms_per_frame"]))
PY
done; done
python bench.py --no-cpu-baseline --graph on > $O/graph.json 2> $O/graph.err && python - <<PY
import json
t=open("$O/graph.json").read(); d=json.loads(t[t.index('{"metric"'):])
print("graph on: %.0f Mrays/s  %.4f ms/frame hip_graph %s" % (d["value"], d["ms_per_frame"], d["config"]["hip_graph"]))
PY
P3D_BENCH_REHEARSAL=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > $O/n2.json 2> $O/n2.err; echo rehearsal rc $?
