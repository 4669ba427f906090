#!/bin/bash
# per-kernel times of the stream schedule
R=$PWD; O=$R/gpurun_out/exp19; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for sc in ${SCENES:-dragon}; do
rocprofv3 --kernel-trace --stats -d $O/$sc --output-format csv -- python3 $R/tools/render_frames.py $sc stream 6 > $O/$sc.log 2>&1 || exit 1
f=$(ls $O/$sc/*/*kernel_stats.csv | head -1)
python3 - <<PY
import csv
for r in csv.DictReader(open("$f")):
    print("%-90s calls %5s avg %10.1f us total %10.1f us  %s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3, r["Percentage"]))
PY
t=$(ls $O/$sc/*/*kernel_trace.csv | head -1)
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$t")) if "p3d::" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
last=[r for r in rows][-19:]
t0=int(last[0]["Start_Timestamp"])
for r in last:
    print("%9.1f us +%8.1f us  %s grid %s" % ((int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r["Kernel_Name"][10:60], r["Grid_Size"]))
PY
done
if grep -q "Memory access fault" $O/*.log; then echo "GPU FAULT"; exit 99; fi
