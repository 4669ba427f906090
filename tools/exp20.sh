#!/bin/bash
# stream schedule: knob sweep on the dragon (kernel trace)
R=$PWD; O=$R/gpurun_out/exp20; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {
tag=$1
rocprofv3 --kernel-trace -d $O/$tag --output-format csv -- python3 $R/tools/render_frames.py dragon stream 4 > $O/$tag.log 2>&1 || exit 1
t=$(ls $O/$tag/*/*kernel_trace.csv | head -1)
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$t")) if "p3d::" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
last=rows[-19:]
t0=int(last[0]["Start_Timestamp"])
print("$tag: frame %.1f us; " % ((int(last[-1]["End_Timestamp"])-t0)/1e3) + " ".join("%s%.0f" % ("X" if "extend" in r["Kernel_Name"] else ("S" if "shade" in r["Kernel_Name"] else "R"), (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3) for r in last))
PY
}
export P3D_STREAM_REFILL=64
P3D_STREAM_MIN_BLOCKS=1 run base_mb1
P3D_STREAM_MIN_BLOCKS=1 P3D_STREAM_STATIC=1 run static_mb1
P3D_STREAM_MIN_BLOCKS=1 P3D_STREAM_WAVES_PCT=200 run mb1_w200
P3D_STREAM_MIN_BLOCKS=1 P3D_STREAM_WAVES_PCT=400 P3D_STREAM_STATIC=1 run static_mb1_w400
P3D_STREAM_MIN_BLOCKS=1 P3D_STREAM_WAVES_PCT=50 run mb1_w50
if grep -q "Memory access fault" $O/*.log; then echo "GPU FAULT"; exit 99; fi
