#!/bin/bash
# PMC passes for one bench command on the GPU box (run through gpurun from the repo root):
#   tools/pmc_collect.sh TAG [bench.py arguments...]        (P3D_PMC_SCRIPT=tools/x.py profiles that script instead;
#                                                             P3D_PMC_PASSES="1 2" limits the passes)
# Counters that do not fit one pass go in separate passes (MI355X_MICROARCH.md); --pmc is never
# combined with tracing options.  Results: gpurun_out/pmc_TAG/passN/, summary JSON from tools/pmc_summary.py.
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE"
 "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_SMEM"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"
)
SCRIPT=$R/${P3D_PMC_SCRIPT:-bench.py}
WANT=${P3D_PMC_PASSES:-1 2 3 4 5 6}
DIRS=""
for i in $WANT; do
  p=${PASSES[$((i-1))]}
  rocprofv3 --pmc $p -d "$OUT/pass$i" --output-format csv -- python3 "$SCRIPT" "$@" > "$OUT/pass$i.log" 2>&1
  echo "pass $i done: $p"
  DIRS="$DIRS $OUT/pass$i"
done
python3 "$R/tools/pmc_summary.py" "$OUT/summary.json" $DIRS --kernels p3d --note "rocprofv3 --pmc, one pass per counter group, of: ${P3D_PMC_SCRIPT:-bench.py} $*"
