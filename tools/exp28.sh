#!/bin/bash
# N = 2 rehearsal of bench.py on one GPU (host-memory gather path) + the plain N = 1 line
O=gpurun_out/exp28; mkdir -p $O
P3D_BENCH_REHEARSAL=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > $O/n2.json 2> $O/n2.err; echo rc $?
tail -c 600 $O/n2.json; echo
if grep -q "Memory access fault" $O/*; then echo "GPU FAULT"; exit 99; fi
