#!/bin/bash
# kernel change check: GPU tests, level-1 / deeper-level instruction counts (PMC), config 2 / 4 / dragon / synthetic timings
R=$PWD; O=$R/gpurun_out/exp39; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?
tail -2 $O/pytest.txt
if grep -q "Memory access fault" $O/pytest.txt; then echo "GPU FAULT"; exit 99; fi
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $O/p --output-format csv -- python3 $R/tools/render_frames.py mount_low wavefront 6 > $O/p.log 2>&1 || exit 1
python3 $R/tools/pmc_summary.py $O/s.json $O/p --kernels wf_primary,wf_secondary > /dev/null
python3 - <<PY
import json
d=json.load(open('$O/s.json'))
for k,e in d['kernels'].items():
    w=e['SQ_WAVES']; print(k[:60], 'valu/w %.0f salu/w %.0f lds/w %.1f gui %.0f' % (e['SQ_INSTS_VALU']/w, e['SQ_INSTS_SALU']/w, e['SQ_INSTS_LDS']/w, e['GRBM_GUI_ACTIVE']/8))
PY
cd $R
python3 tools/perf_probe.py mount_low --n 300 2>&1 | grep -E "wavefront lds/lane"
python3 tools/config4.py 2>&1 | grep "^tile"
timeout -k 10 600 python tools/schedule_probe.py --quick 2>&1 | grep -E "dragon .*tree|MISMATCH"
python3 tools/perf_probe.py --synthetic 1000000 --n 10 2>&1 | grep -E "wavefront hbm/lane"
