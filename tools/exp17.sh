#!/bin/bash
# where the level-1 kernel's instructions go: PMC instruction counts with shading / shadow queries switched off (diagnostic build)
R=$PWD; O=$R/gpurun_out/exp17; mkdir -p $O
export P3D_LIB=$R/u_4a_2s_p3d_raytracer_template2_amd/libp3d_hip_dbg.so
cd /tmp && export TMPDIR=/tmp
for skip in 0 1 2; do
  export P3D_DEBUG_SKIP=$skip
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/p$skip --output-format csv -- python3 $R/tools/render_frames.py mount_low wavefront 6 > $O/p$skip.log 2>&1
  python3 $R/tools/pmc_summary.py $O/s$skip.json $O/p$skip --kernels wf_primary > /dev/null
  python3 - <<PY
import json
d=json.load(open('$O/s$skip.json'))
for k,e in d['kernels'].items():
    w=e['SQ_WAVES']; print('skip $skip', k[:50], 'valu/w %.0f salu/w %.0f lds/w %.1f smem/w %.1f cyc/w %.0f gui %.0f' % (e['SQ_INSTS_VALU']/w, e['SQ_INSTS_SALU']/w, e['SQ_INSTS_LDS']/w, e['SQ_INSTS_SMEM']/w, e['SQ_WAVE_CYCLES']*4/w, e['GRBM_GUI_ACTIVE']/8))
PY
done
