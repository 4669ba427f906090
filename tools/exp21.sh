#!/bin/bash
# stream schedule on the dragon: PMC per kernel
R=$PWD; O=$R/gpurun_out/exp21; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/p1 --output-format csv -- python3 $R/tools/render_frames.py dragon stream 3 > $O/p1.log 2>&1 || exit 1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU -d $O/p2 --output-format csv -- python3 $R/tools/render_frames.py dragon stream 3 > $O/p2.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("$O/p1","$O/p2"):
    for path in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if "p3d::wf_" not in r["Kernel_Name"]: continue
            acc[(r["Kernel_Name"][10:45], r["Dispatch_Id"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
# group dispatches by kernel in order; print the last frame's launches
keys=sorted(acc.keys(), key=lambda k:int(k[1]))
for k in keys[-19:]:
    e={c:sum(v)/len(v) for c,v in acc[k].items()}
    print(k[0], " ".join("%s=%.3g" % (c.replace("SQ_","").replace("_sum",""), v) for c,v in sorted(e.items())))
PY
if grep -q "Memory access fault" $O/*.log; then echo "GPU FAULT"; exit 99; fi
