#!/bin/bash
# round 3, experiment 21: SQ counters of config 4's tile kernel, device-library powf (v0) against the restatement
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in _v0 ""; do
  export P3D_LIB=$R/u_4a_2s_p3d_raytracer_template2_amd/libp3d_hip$v.so
  P3D_PMC_SCRIPT=tools/config4.py P3D_PMC_PASSES="1 2" tools/pmc_collect.sh r3_21$v > gpurun_out/pmc_r3_21$v.log 2>&1
  python3 - <<PY
import json
d=json.load(open("$R/gpurun_out/pmc_r3_21$v/summary.json"))["kernels"]
for k,v in d.items():
    if "wf_tile_kernel" in k or "wf_primary" in k or "wf_secondary" in k:
        w=v["SQ_WAVES"]
        print("lib$v", k[5:42], "n", v["_n"], "waves", w, "per launch: VALU %.0f SALU %.0f SMEM %.0f LDS %.0f VMEM %.0f | ACTIVE_VALU %.0f  WAVE_CYCLES %.0f BUSY %.0f WAIT_INST_ANY %.0f" % (
            v["SQ_INSTS_VALU"], v["SQ_INSTS_SALU"], v["SQ_INSTS_SMEM"], v["SQ_INSTS_LDS"], v["SQ_INSTS_VMEM"], v["SQ_ACTIVE_INST_VALU"], v["SQ_WAVE_CYCLES"], v["SQ_BUSY_CYCLES"], v["SQ_WAIT_INST_ANY"]))
PY
done
