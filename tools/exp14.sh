#!/bin/bash
O=gpurun_out/exp14; mkdir -p $O
run() { python bench.py --no-cpu-baseline --steps 30 "$@" > $O/b.json 2>/dev/null; python3 -c "
import json,sys;d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]);print('$*', round(d['value']), round(d['ms_per_frame'],4), d['config']['frames_in_flight'], d['config']['frames_per_step'])"; }
run
run --frames-per-step 24
run --frames-per-step 48
run --frames-per-step 24 --frames-in-flight 2
run --frames-per-step 24 --frames-in-flight 4
run --frames-per-step 24 --frames-in-flight 6
run --frames-per-step 24 --graph on
P3D_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/n2.json 2> $O/n2.err; echo rehearsal rc $?; tail -c 300 $O/n2.json
