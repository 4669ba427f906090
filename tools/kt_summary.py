#!/usr/bin/env python3
"""Per-kernel durations of the LAST frame in a rocprofv3 --kernel-trace csv (diagnostic).
usage: kt_summary.py DIR [first_kernel_substr]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
first = sys.argv[2] if len(sys.argv) > 2 else "raygen"
rows = [r for r in csv.DictReader(open(f))]
seq = [(r["Kernel_Name"].split("(")[0][-48:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
        int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["VGPR_Count"], r["Grid_Size_X"]) for r in rows]
idx = [i for i, s in enumerate(seq) if first in s[0]]
i = idx[-1]
t0 = seq[i][2]
for s in seq[i:]:
    print("%-50s %9.1f us  start %9.1f  vgpr %s grid %s" % (s[0], s[1], (s[2] - t0) / 1e3, s[4], s[5]))
print("span %.1f us" % ((seq[-1][3] - t0) / 1e3))
