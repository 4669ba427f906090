#!/bin/bash
# Device-only assembly of p3d_kernels.hip and a static instruction mix of one kernel (tool, CPU only).
# usage: tools/isa_stats.sh OUT.s 'KERNEL_MANGLED_PREFIX' [extra hipcc flags...]
set -e
OUT=$1; KER=$2; shift 2
CS=/root/repo/u_4a_2s_p3d_raytracer_template2_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc -I/root/repo/include -I$CS "$@" \
  --cuda-device-only -S $CS/p3d_kernels.hip -o $OUT 2>/dev/null
awk -v k="^$KER" '$0 ~ k":" {p=1} p {print} p && /^\.Lfunc_end/ {exit}' $OUT > $OUT.kernel
echo "lines $(wc -l < $OUT.kernel) valu $(grep -cE '^\s+v_' $OUT.kernel) salu $(grep -cE '^\s+s_' $OUT.kernel) ds $(grep -cE '^\s+ds_' $OUT.kernel) vmem $(grep -cE '^\s+(global|buffer|flat|scratch)_' $OUT.kernel) v_mov $(grep -cE '^\s+v_mov' $OUT.kernel) v_pk $(grep -cE '^\s+v_pk' $OUT.kernel) lanes_rw $(grep -cE '^\s+v_(read|write)lane' $OUT.kernel)"
awk -v k="$KER" '$0 ~ "\\.name:.*"k {p=1} p && /sgpr_count|sgpr_spill|vgpr_count|vgpr_spill|group_segment_fixed/ {printf "%s ", $0} p && /wavefront_size/ {print ""; exit}' $OUT
