#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/exp7; mkdir -p $O
P3D_PMC_PASSES="1 2" tools/pmc_collect.sh exp7 --no-cpu-baseline --frames-in-flight 1 --steps 2 --warmup 2 > $O/pmc.log 2>&1
cp $R/gpurun_out/pmc_exp7/summary.json $O/pmc.json
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/exp7/pmc.json'))
for k,e in d['kernels'].items():
    if '<true' in k: continue
    w=e.get('SQ_WAVES',1)
    print(k[:70], 'waves',w, 'valu/w %.0f salu/w %.0f lds/w %.1f vmem/w %.1f smem/w %.1f' % (e.get('SQ_INSTS_VALU',0)/w, e.get('SQ_INSTS_SALU',0)/w, e.get('SQ_INSTS_LDS',0)/w, e.get('SQ_INSTS_VMEM',0)/w, e.get('SQ_INSTS_SMEM',0)/w),
          'wavecyc/w %.0f' % (e.get('SQ_WAVE_CYCLES',0)*4/w), 'wait_any %.2f wait_inst %.2f active %.2f' % (e.get('SQ_WAIT_ANY',0)/max(e.get('SQ_WAVE_CYCLES',1),1), e.get('SQ_WAIT_INST_ANY',0)/max(e.get('SQ_WAVE_CYCLES',1),1), e.get('SQ_ACTIVE_INST_ANY',0)/max(e.get('SQ_WAVE_CYCLES',1),1)), 'lane_util %.2f' % e.get('valu_lane_utilisation',0), 'gui %.0f' % (e.get('GRBM_GUI_ACTIVE',0)/8))
PY
