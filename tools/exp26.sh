#!/bin/bash
# level-1 kernel with / without the per-primitive eye constants: PMC instruction counts, kernel time, frame time
R=$PWD; O=$R/gpurun_out/exp26; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for eye in 0 1; do
  if [ $eye = 0 ]; then export P3D_NO_EYE_CONSTANTS=1; else unset P3D_NO_EYE_CONSTANTS; fi
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/p$eye --output-format csv -- python3 $R/tools/render_frames.py mount_low wavefront 6 > $O/p$eye.log 2>&1 || exit 1
  python3 $R/tools/pmc_summary.py $O/s$eye.json $O/p$eye --kernels wf_primary > /dev/null
  python3 - <<PY
import json
d=json.load(open('$O/s$eye.json'))
for k,e in d['kernels'].items():
    w=e['SQ_WAVES']; print('eye $eye', k[:60], 'valu/w %.0f salu/w %.0f lds/w %.1f smem/w %.1f cyc/w %.0f gui %.0f' % (e['SQ_INSTS_VALU']/w, e['SQ_INSTS_SALU']/w, e['SQ_INSTS_LDS']/w, e['SQ_INSTS_SMEM']/w, e['SQ_WAVE_CYCLES']*4/w, e['GRBM_GUI_ACTIVE']/8))
PY
  rocprofv3 --kernel-trace --stats -d $O/t$eye --output-format csv -- python3 $R/tools/render_frames.py mount_low wavefront 200 > $O/t$eye.log 2>&1 || exit 1
  f=$(ls $O/t$eye/*/*kernel_stats.csv | head -1); grep -E "wf_primary|wf_secondary|wf_resolve" $f | cut -d, -f1-4 | cut -c1-120
  cd $R && python3 tools/perf_probe.py mount_low --n 300 2>&1 | grep -E "wavefront lds/lane"; cd /tmp
done
if grep -q "Memory access fault" $O/*.log; then echo "GPU FAULT"; exit 99; fi
