#!/bin/bash
# N = 4 rehearsal of bench.py on one GPU (4 ranks on cuda:0, gather through host memory): exercises the world = 4 code path
O=gpurun_out/exp37; mkdir -p $O
P3D_BENCH_REHEARSAL=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 4 --steps 5 --warmup 2 > $O/n4.json 2> $O/n4.err; echo rc $?
python - <<PY
import json
t=open("$O/n4.json").read(); d=json.loads(t[t.index('{"metric"'):])
print(d["value"], d["unit"], d["n_gpus"], d["config"]["frames_in_flight"], d["config"]["gather"], d["config"]["frame_checksum"])
PY
tail -3 $O/n4.err
if grep -q "Memory access fault" $O/*; then echo "GPU FAULT"; exit 99; fi
