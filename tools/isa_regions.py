#!/usr/bin/env python3
"""Static instruction account of one kernel by SOURCE REGION (CPU-only tool).

Compiles p3d_kernels.hip to gfx950 assembly with line tables (-gline-tables-only: same code, the .loc comments carry the
inlined-at chain of every instruction), maps every frame of a chain to the function that contains that line, and files
each instruction under a region: ray generation, closest-hit walk (node step / triangle / sphere / box / loop control),
shadow walk (the same split), shading (normal, light term, children), queue append, scene copy, kernel frame.
Per region: vector, scalar (without s_waitcnt / s_nop, which are listed beside them), LDS and vector-memory instructions,
and the v_readfirstlane / v_cndmask / s_cbranch counts.  STATIC counts: a loop body counts once.

usage: python tools/isa_regions.py 'wf_primary_kernel<false, true, 0, 1, false>' [extra hipcc flags...]
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(REPO, "u_4a_2s_p3d_raytracer_template2_amd", "csrc")
FILES = ["p3d_kernels.hip", "p3d_device_math.h", "p3d_traverse.h", "p3d_shade.h"]


def function_ranges(path):
    """[(first line, last line, name)] of the functions of a source file (brace matching from a definition line)."""
    out, lines = [], open(path).read().split("\n")
    head = re.compile(r"^(?:template\s*<[^>]*>\s*)?(?:static\s+)?(?:__device__|__global__|__host__)[^;{]*?\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;]*$")
    i = 0
    while i < len(lines):
        m = head.match(lines[i].strip())
        if not m and lines[i].strip().startswith("template") and i + 1 < len(lines):
            m = head.match((lines[i].strip() + " " + lines[i + 1].strip()))
        if m:
            name = m.group(1)
            j, depth, seen = i, 0, False
            while j < len(lines):
                depth += lines[j].count("{") - lines[j].count("}")
                seen = seen or "{" in lines[j]
                if seen and depth <= 0:
                    break
                j += 1
            out.append((i + 1, j + 1, name))
            i = j + 1
        else:
            i += 1
    return out


def main():
    want = sys.argv[1]
    flags = sys.argv[2:]
    tmp = tempfile.mkdtemp(prefix="p3d_isa_")
    asm = os.path.join(tmp, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                           "-fno-slp-vectorize", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-rdc", "-gline-tables-only",
                           "-I" + os.path.join(REPO, "include"), "-I" + CS, "--cuda-device-only", "-S", os.path.join(CS, "p3d_kernels.hip"),
                           "-o", asm] + flags, stderr=subprocess.DEVNULL)
    ranges = {f: function_ranges(os.path.join(CS, f)) for f in FILES}

    def func_of(path, line):
        f = os.path.basename(path)
        for a, b, n in ranges.get(f, ()):
            if a <= line <= b:
                return n
        return None

    # the kernel's mangled name
    names = [l.split(":")[0] for l in open(asm) if re.match(r"^_ZN3p3d\S+:", l)]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    mangled = [n for n, d in zip(names, dem) if want in d]
    if not mangled:
        raise SystemExit("no kernel matching %r" % want)
    mangled = mangled[0]

    def region(chain):
        fs = [x for x in chain if x]
        s = set(fs)
        inner = fs[0] if fs else ""
        walk = None
        if s & {"any_hit", "leaf_any", "any_hit_packet", "any_hit_shared"}:
            walk = "shadow walk"
        elif s & {"closest_hit", "leaf_closest", "closest_hit_packet", "closest_hit_shared", "take_closer"}:
            walk = "closest walk"
        if walk:
            if s & {"hit_triangle"}: return walk + ": triangle test"
            if s & {"hit_sphere"}: return walk + ": sphere test"
            if s & {"hit_aabox"}: return walk + ": box test"
            if s & {"hit_plane", "ref_unit_box_hit"}: return walk + ": planes"
            if s & {"node_test", "slab", "make_slab", "slab_rcp"}: return walk + ": node step (slab tests)"
            if s & {"sv_leaf", "sv_tri", "sv_sphere", "sv_sphere_meta", "sv_box", "leaf_closest", "leaf_any", "take_closer"}: return walk + ": leaf loop / fetch"
            return walk + ": loop control / stack"
        if s & {"camera_ray", "primary_ray_tab", "primary_ray", "primary_ray_lens"}: return "ray generation"
        if "light_term" in s: return "shade: light term (Blinn-Phong, powf)"
        if "prim_normal" in s: return "shade: normal"
        if "light_occluded" in s: return "shade: shadow ray set-up"
        if "shade_hit" in s: return "shade: hit point, children, material"
        if s & {"emit", "deliver", "sink_sample", "write_pixel", "lane_rank", "combine_pair", "combine_node"}: return "queue append / deliver"
        if s & {"make", "scene_dwords"}: return "scene copy into LDS"
        if s & {"stamp", "stamp_record", "stamps_on", "stamp_wave"}: return "frame: diagnostic stamps"
        if s & {"flush_counters"}: return "frame: counters"
        if s & {"tile_pixel"}: return "frame: tile -> pixel"
        if s & {"shard_of", "count_in_array", "count_out_array", "ncount_self_array", "wave_stack"}: return "frame: shard / stack set-up"
        return "frame: kernel body (parameters, first-block clearing, glue)"

    stats = collections.defaultdict(collections.Counter)
    cur, on = [], False
    loc = re.compile(r"([^\s;@\[\]]+):(\d+):(\d+)")
    for l in open(asm):
        if l.startswith(mangled + ":"):
            on = True
            continue
        if not on:
            continue
        if l.startswith(".Lfunc_end"):
            break
        t = l.strip()
        if t.startswith(".loc"):
            c = t.split(";", 1)[1] if ";" in t else ""
            cur = [func_of(p, int(ln)) for p, ln, _ in loc.findall(c)]
            continue
        if not t or t.startswith((".", ";")) or t.endswith(":"):
            continue
        op = t.split()[0]
        r = region(cur)
        k = stats[r]
        if op.startswith("v_"):
            k["valu"] += 1
            if op.startswith("v_readfirstlane") or op.startswith("v_readlane") or op.startswith("v_writelane"): k["lane_rw"] += 1
            if op.startswith("v_cndmask"): k["cndmask"] += 1
        elif op.startswith(("s_load", "s_buffer_load", "s_store", "s_memrealtime", "s_memtime", "s_dcache")): k["smem"] += 1
        elif op == "s_waitcnt": k["waitcnt"] += 1
        elif op == "s_nop": k["nop"] += 1
        elif op.startswith("s_"):
            k["salu"] += 1
            if op.startswith("s_cbranch") or op == "s_branch": k["branch"] += 1
            if "saveexec" in op: k["saveexec"] += 1
        elif op.startswith("ds_"): k["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): k["vmem"] += 1
    cols = ["valu", "salu", "branch", "saveexec", "smem", "waitcnt", "nop", "lane_rw", "cndmask", "lds", "vmem"]
    print("static instruction account of %s" % want)
    print("%-44s" % "region" + "".join("%9s" % c for c in cols))
    tot = collections.Counter()
    for r in sorted(stats, key=lambda r: -(stats[r]["valu"] + stats[r]["salu"])):
        print("%-44s" % r + "".join("%9d" % stats[r][c] for c in cols))
        tot.update(stats[r])
    print("%-44s" % "TOTAL" + "".join("%9d" % tot[c] for c in cols))


if __name__ == "__main__":
    main()
