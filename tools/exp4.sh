#!/bin/bash
# kernel traces of one config-2 frame per schedule
set -o pipefail
R=$PWD; O=$R/gpurun_out/exp4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for s in wavefront wavefront_packet tile; do
  rocprofv3 --kernel-trace --stats -d $O/kt_$s --output-format csv -- python3 $R/tools/render_frames.py mount_low $s 8 > $O/kt_$s.log 2>&1
  python3 $R/tools/kt_summary.py $O/kt_$s ${1:-wf_primary} > $O/kt_$s.txt 2>&1 || python3 $R/tools/kt_summary.py $O/kt_$s wf_tile > $O/kt_$s.txt 2>&1
  echo "== $s"; cat $O/kt_$s.txt
done
