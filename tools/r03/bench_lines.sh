#!/bin/bash
# the round's bench lines (profiles registered in profiles/current.json) + a rehearsal of the N = 2 code path on one GPU
set -o pipefail
O=gpurun_out/r03_lines; mkdir -p $O
python bench.py > $O/r03_bench_config2.json 2> $O/bench_config2.err; tail -c 200 $O/r03_bench_config2.json; echo
python bench.py --workload config3 --steps 20 --warmup 3 > $O/r03_bench_config3.json 2> $O/bench_config3.err; tail -c 200 $O/r03_bench_config3.json; echo
python bench.py --workload config4 --steps 10 --warmup 2 > $O/r03_bench_config4.json 2> $O/bench_config4.err; tail -c 200 $O/r03_bench_config4.json; echo
python bench.py --workload synthetic --prims 1000000 --steps 10 --warmup 2 > $O/r03_bench_synthetic_1e6.json 2> $O/bench_synthetic_1e6.err; tail -c 200 $O/r03_bench_synthetic_1e6.json; echo
python bench.py --workload pathtracer --steps 3 --warmup 1 > $O/r03_bench_pathtracer.json 2> $O/bench_pathtracer.err; tail -c 200 $O/r03_bench_pathtracer.json; echo
P3D_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 > $O/r03_bench_rehearsal_n2.json 2> $O/rehearsal.err || { tail -20 $O/rehearsal.err; }
tail -c 600 $O/r03_bench_rehearsal_n2.json; echo
P3D_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --workload config3 --steps 3 --warmup 1 > $O/r03_bench_rehearsal_n2_config3.json 2> $O/rehearsal3.err || { tail -20 $O/rehearsal3.err; }
tail -c 400 $O/r03_bench_rehearsal_n2_config3.json; echo
