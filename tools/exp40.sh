#!/bin/bash
# while-while (libp3d_hip.so) vs if-if (libp3d_hip_x.so) walk loops
R=$PWD; O=$R/gpurun_out/exp40; mkdir -p $O
for lib in libp3d_hip.so libp3d_hip_x.so; do
  export P3D_LIB=$R/u_4a_2s_p3d_raytracer_template2_amd/$lib
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $O/p_$lib --output-format csv -- python3 $R/tools/render_frames.py mount_low wavefront 6 > $O/p_$lib.log 2>&1 || exit 1
  python3 $R/tools/pmc_summary.py $O/s_$lib.json $O/p_$lib --kernels wf_primary,wf_secondary > /dev/null
  python3 - <<PY
import json
d=json.load(open('$O/s_$lib.json'))
for k,e in d['kernels'].items():
    w=e['SQ_WAVES']; print('$lib', k[:48], 'valu/w %.0f salu/w %.0f lds/w %.1f gui %.0f' % (e['SQ_INSTS_VALU']/w, e['SQ_INSTS_SALU']/w, e['SQ_INSTS_LDS']/w, e['GRBM_GUI_ACTIVE']/8))
PY
  cd $R
  python3 tools/perf_probe.py mount_low --n 300 2>&1 | grep -E "wavefront lds/lane"
  python3 tools/config4.py 2>&1 | grep "^tile"
  python3 tools/perf_probe.py dragon --tree --n 10 2>&1 | grep -E "tree      hbm/lane"
  python3 tools/perf_probe.py --synthetic 1000000 --n 10 2>&1 | grep -E "wavefront hbm/lane"
done
P3D_LIB=$R/u_4a_2s_p3d_raytracer_template2_amd/libp3d_hip_x.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_scenes.py -m gpu -x -q 2>&1 | tail -2
