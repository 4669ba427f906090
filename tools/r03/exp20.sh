#!/bin/bash
# round 3, experiment 20: SQ counters of config 2's level-1 kernel, device-library powf (v0) against the restatement
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in _v0 ""; do
  export P3D_LIB=$R/u_4a_2s_p3d_raytracer_template2_amd/libp3d_hip$v.so
  P3D_PMC_PASSES="1 2" tools/pmc_collect.sh r3_20$v --no-cpu-baseline --frames-in-flight 1 --steps 2 --warmup 2 > gpurun_out/pmc_r3_20$v.log 2>&1
  python3 - <<PY
import json
d=json.load(open("$R/gpurun_out/pmc_r3_20$v/summary.json"))["kernels"]
for k,v in d.items():
    if "wf_primary_kernel<false" in k or "wf_secondary_kernel<false" in k:
        w=v["SQ_WAVES"]
        print("lib$v", k[5:45], "waves", w, "per wave: VALU %.1f SALU %.1f SMEM %.1f LDS %.1f | ACTIVE_VALU cyc %.0f  WAVE_CYCLES %.0f BUSY %.0f WAIT_INST_ANY %.0f thread_cycles_valu/inst %.1f" % (
            v["SQ_INSTS_VALU"]/w, v["SQ_INSTS_SALU"]/w, v["SQ_INSTS_SMEM"]/w, v["SQ_INSTS_LDS"]/w, v["SQ_ACTIVE_INST_VALU"]/w, v["SQ_WAVE_CYCLES"]/w, v["SQ_BUSY_CYCLES"], v["SQ_WAIT_INST_ANY"]/w, v["SQ_THREAD_CYCLES_VALU"]/v["SQ_INSTS_VALU"]))
PY
done
