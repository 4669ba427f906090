#!/bin/bash
# Register / spill / scratch use of every kernel of p3d_kernels.hip (CPU-only: device-only compile + readelf notes).
# usage: tools/kernel_regs.sh [name filter regex] [extra hipcc flags...]
set -e
here=$(cd "$(dirname "$0")/.." && pwd)
cs=$here/u_4a_2s_p3d_raytracer_template2_amd/csrc
filt=${1:-.}; shift || true
out=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc -I$here/include -I$cs --cuda-device-only --no-gpu-bundle-output -c $cs/p3d_kernels.hip -o $out/k.co "$@"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $out/k.co | python3 -c '
import re, sys, subprocess
txt = sys.stdin.read()
rows = []
for blk in txt.split("- .agpr_count")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    rows.append((g("name"), g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
print("%-110s %5s %5s %6s %6s %8s %6s" % ("kernel", "vgpr", "sgpr", "vspill", "sspill", "scratch", "lds"))
for r, n in sorted(zip(rows, names), key=lambda t: t[1]):
    if re.search(sys.argv[1], n):
        print("%-110s %5s %5s %6s %6s %8s %6s" % ((n.replace("p3d::", "").replace("(p3d::LaunchParams)", "")[:110],) + r[1:]))
' "$filt"
rm -rf $out
