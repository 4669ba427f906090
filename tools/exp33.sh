#!/bin/bash
R=$PWD; O=$R/gpurun_out/exp33; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt --output-format csv -- python3 $R/tools/render_frames.py 100000 tree 4 > $O/kt.log 2>&1 || { tail $O/kt.log; exit 1; }
f=$(ls $O/kt/*/*kernel_stats.csv | head -1)
python3 - <<PY
import csv
for r in csv.DictReader(open("$f")):
    print("%-90s calls %5s avg %12.1f us  %s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
