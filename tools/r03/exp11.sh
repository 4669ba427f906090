#!/bin/bash
# round 3, experiment 11: scalar divisions out of the level kernels (2-D level-1 grid, constant shard count): SALU per wave + timing
set -e
O=gpurun_out/r3_11; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for rep in 1 2 3; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print({k:d[k] for k in ('value','ms_per_frame','ms_per_frame_latency')}, d['roofline']['kernel_ms_live'], d['config']['frame_checksum'])" >> $O/config2.txt
done
cat $O/config2.txt
P3D_PMC_PASSES="1 2" timeout -k 10 500 tools/pmc_collect.sh r3_11_config2 --no-cpu-baseline --frames-in-flight 1 --steps 2 --warmup 2 > $O/pmc.log 2>&1 || { tail -20 $O/pmc.log; exit 1; }
cp gpurun_out/pmc_r3_11_config2/summary.json $O/config2_pmc.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_11/config2_pmc.json'))
for k,v in d['kernels'].items():
    if v.get('SQ_WAVES',0)>1000 and '<false' in k or 'resolve' in k:
        print(k[:70], 'VALU/wave %.0f  SALU/wave %.0f'%(v['SQ_INSTS_VALU']/v['SQ_WAVES'], v['SQ_INSTS_SALU']/v['SQ_WAVES']))
PY
# tile kernel with the first 256 queued rays of a level in LDS: config 4 timing + traffic (FETCH_SIZE / WRITE_SIZE passes)
P3D_VERBOSE=1 timeout -k 10 300 python tools/config4.py 2>&1 | grep -E "p3d: tile|device" | sort | uniq > $O/config4.txt; cat $O/config4.txt
P3D_PMC_SCRIPT=tools/config4.py P3D_PMC_PASSES="3 4" timeout -k 10 500 tools/pmc_collect.sh r3_11_config4 > $O/pmc4.log 2>&1 || { tail -20 $O/pmc4.log; exit 1; }
cp gpurun_out/pmc_r3_11_config4/summary.json $O/config4_pmc.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_11/config4_pmc.json'))
for k,v in d['kernels'].items():
    if 'tile_kernel<false' in k: print(k[:70], 'HBM-side bytes per launch %.3g (FETCH %.0f KB x2 + WRITE %.0f KB)'%(v['hbm_bytes_per_launch_corrected'], v['FETCH_SIZE'], v['WRITE_SIZE']))
PY
