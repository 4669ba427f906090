#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel launch over one or more pass directories.

usage: python tools/pmc_summary.py OUT.json DIR [DIR ...] [--kernels substr,substr] [--note TEXT]

Each DIR is the -d directory of one `rocprofv3 --pmc ... --output-format csv` pass (counters that do
not fit one pass go in separate passes, as MI355X_MICROARCH.md prescribes).  Per kernel name the
counters are averaged over its launches; `hbm_bytes_per_launch_corrected` applies the guide's gfx950
rule: FETCH_SIZE and WRITE_SIZE are in KB and FETCH_SIZE counts 128-B requests at 64 B, so
HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    # keep the template arguments: the counting builds (<true, ...>) are different, much slower kernels
    name = re.sub(r"\(p3d::LaunchParams\)$", "", name)
    return re.sub(r"^void ", "", name)


def main():
    args = sys.argv[1:]
    out, dirs, want, note = args[0], [], None, ""
    i = 1
    while i < len(args):
        if args[i] == "--kernels":
            want = args[i + 1].split(","); i += 2
        elif args[i] == "--note":
            note = args[i + 1]; i += 2
        else:
            dirs.append(args[i]); i += 1
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    launch = {}
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                k = short(row["Kernel_Name"])
                if want and not any(w in k for w in want):
                    continue
                a = acc[k][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"]); a[1] += 1
                launch[k] = {"grid": row["Grid_Size"], "wg": row["Workgroup_Size"], "lds": row["LDS_Block_Size"],
                             "vgpr": row["VGPR_Count"], "accum_vgpr": row["Accum_VGPR_Count"], "sgpr": row["SGPR_Count"],
                             "scratch": row["Scratch_Size"]}
    # digest of the kernel sources these counters were taken from (bench.py flags a profile of other kernels as stale)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    import bench                                       # one list of kernel sources: bench.DEVICE_SOURCES
    res = {"_note": note, "kernel_source_digest": bench.kernel_source_digest(), "kernels": {}}
    for k, cs in sorted(acc.items()):
        e = {c: v[0] / v[1] for c, v in sorted(cs.items())}
        e["_n"] = max(v[1] for v in cs.values())
        e["_launch"] = launch[k]
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch_corrected"] = (2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024
        if "SQ_THREAD_CYCLES_VALU" in e and e.get("SQ_ACTIVE_INST_VALU"):
            e["valu_lane_utilisation"] = e["SQ_THREAD_CYCLES_VALU"] / (64.0 * e["SQ_ACTIVE_INST_VALU"])
        res["kernels"][k] = e
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, e in res["kernels"].items():
        print(k, {c: round(v, 1) for c, v in e.items() if isinstance(v, float)})


if __name__ == "__main__":
    main()
