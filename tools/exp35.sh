#!/bin/bash
# per-launch timeline of one config-2 frame (wavefront schedule)
R=$PWD; O=$R/gpurun_out/exp35; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/kt --output-format csv -- python3 $R/tools/render_frames.py mount_low wavefront 40 > $O/kt.log 2>&1 || { tail $O/kt.log; exit 1; }
t=$(ls $O/kt/*/*kernel_trace.csv | head -1)
python3 - <<PY
import csv, collections
rows=[r for r in csv.DictReader(open("$t")) if "p3d::wf_" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
n=6
frames=[rows[i:i+n] for i in range(0,len(rows)-n+1,n)][5:]
acc=collections.defaultdict(list)
for fr in frames:
    t0=int(fr[0]["Start_Timestamp"])
    for k,r in enumerate(fr):
        acc[k].append(((int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r["Kernel_Name"][10:32]))
for k in range(n):
    st=sorted(a[0] for a in acc[k]); du=sorted(a[1] for a in acc[k])
    print("launch %d %-22s start +%6.1f us  duration %5.1f us (median over %d frames)" % (k, acc[k][0][2], st[len(st)//2], du[len(du)//2], len(st)))
tot=sorted((int(fr[-1]["End_Timestamp"])-int(fr[0]["Start_Timestamp"]))/1e3 for fr in frames)
print("frame first start -> last end: median %.1f us" % tot[len(tot)//2])
PY
