#!/usr/bin/env python3
"""bench.py -- Mrays/s of the Whitted hot path, one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W                        (BASELINE config 2, the headline)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workloads (--workload):
  config2    mount_low.p3f 1920x1080 depth 4 BVH -- the configuration BASELINE.json's metric is quoted on.
             A step = --frames-per-step frames into HBM-resident buffers, --frames-in-flight of them overlapped
             (independent frames on separate scene handles / HIP streams, like the reference's render-another-
             image loop).
  config3    dragon.p3f (100 005 primitives) 1920x1080 depth 4 BVH (BASELINE config 3): every primitive is
             traversed, like in the reference; the cull_never_hit build is timed beside it and reported as a
             second figure (config.cull_never_hit), never as `value`.  CPU baseline at 128x128.
  config4    mount_low.p3f 4096x4096 depth 6, 2x2 jittered samples + thin lens (BASELINE config 4): one frame per
             step, the sample array uploaded once.
  synthetic  the SURVEY 8d scaling scene (--prims random spheres + triangles, BVH read from HBM).
  pathtracer BASELINE config 5 (P3D_RT.glsl), samples split over the ranks.

N > 1: every frame is cut into interleaved 16-row blocks (p3d_render rank / world: total work fixed, strong
scaling); each step ends with ONE gather of the compact tile buffers to rank 0 THROUGH THE C-ABI (p3d_gather:
grouped ncclSend / ncclRecv over RCCL, include/p3d_hip.h) on a communication stream, plus p3d_deinterleave_frames
there; with two tile-buffer sets the gather of step i overlaps the rendering of step i + 1 (--gather sync: one
set).  torch.distributed (gloo) only carries the 128-byte communicator id, the barrier and the timing reduction.
value = rays of all frames / max-over-ranks wall time; a ray is one closest-hit or one shadow query,
counted by the counting build of the same kernels on the same frame.

The line checks what it timed:
  "frame_matches_reference"  the LAST TIMED FRAME (device buffer, downloaded after the timed region) equals the
                 frame the cpu_baseline leg rendered -- by the reference's own object code where that renders the
                 same frame (config 2), else by the oracle port on all cores at full size, with the reference's
                 frame at the sample resolution compared against a GPU frame of that resolution; rgb8 EQUAL (the
                 device evaluates powf with the host libm's algorithm, csrc/p3d_powf.h) and equal ray counts.
                 Details in "frame_check".
  "gather_verified"  (N > 1) the gathered, de-interleaved frame on rank 0 equals a one-GPU render of the same frame.

Extra objects on the JSON line:
  "roofline"     for the ray kernel with the most device time (from the rocprofv3 kernel stats and PMC summaries
                 committed under profiles/, see profiles/current.json).  bound "valu_issue" for scenes served from
                 LDS (config 2 / 4: HBM is not what binds a 12-primitive scene): achieved = SQ_INSTS_VALU per launch
                 / its duration against one wave-instruction per 2 cycles per SIMD (measured:
                 profiles/r02_valu_rate_ubench.txt).  bound "fetch_latency" for scenes read from HBM (config 3,
                 synthetic): those kernels reach neither the issue nor the HBM roof, they wait on dependent
                 fetches -- the object carries lanes active, SQ_WAIT_ANY / SQ_WAVE_CYCLES, L1 / L2 hit rates, L2 and
                 fabric GB/s against their peaks, and the same vector-issue fraction (= achieved / peak / frac).
                 Every number can be recomputed from the files named in roofline.source;
                 roofline.kernel_ms_live is this run's own HIP-event duration of the launch named in
                 kernel_ms_live_of (the schedule's first launch: what p3d_get_profile times).  Those figures are of ONE
                 frame alone; roofline.whole_frame (N = 1) is the timed region as a whole: the vector instructions of
                 all launches of a frame (same PMC files) over this run's time per frame with its frames in flight.
  "cpu_baseline" at N = 1: the REFERENCE's own object code (oracle/_ref, built from /root/reference in the build
                 container and carried as a .so) rendering whole frames of the same configuration on one host
                 core -- the reference is single-threaded; the oracle port on all cores is reported beside it.
  "env"          the library that was loaded and every P3D_* environment variable that was set.
"""
import argparse
import hashlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
L2_PEAK_GBS = 34500.0            # aggregate L2 bandwidth (MI355X_MICROARCH.md, "L2 (per XCD)")
CLOCK_GHZ = 2.4                  # max shader clock (MI355X_MICROARCH.md)
N_SIMD, N_CU = 1024, 256
VALU_CYCLES_PER_INST = 2.0       # one wave64 VALU instruction per 2 cycles per SIMD (measured, ubench)
SALU_CYCLES_PER_INST_MIXED = 3.62   # a scalar instruction interleaved 1:2 with vector ones: (12 x 2.13 - 8 x 1.68) / 4 ticks, 1.68 ticks = 2 cycles
ROW_BLOCK = 16

WORKLOADS = {
    "config2": dict(scene="mount_low", res=(1920, 1080), depth=4, spp=0, frames=12,
                    name="mount_low.p3f 1920x1080 depth 4 BVH (BASELINE config 2)"),
    "config3": dict(scene="dragon", res=(1920, 1080), depth=4, spp=0, frames=4, cpu_res=(128, 128),
                    name="dragon.p3f (100 005 primitives) 1920x1080 depth 4 BVH (BASELINE config 3)"),
    "config4": dict(scene="mount_low", res=(4096, 4096), depth=6, spp=2, frames=1, cpu_res=(1024, 1024),
                    name="mount_low.p3f 4096x4096 depth 6, 2x2 samples + thin lens (BASELINE config 4)"),
    "synthetic": dict(scene=None, res=(1920, 1080), depth=4, spp=0, frames=12, name=None),
}
HBM_SCENE_WORKLOADS = ("config3", "synthetic")      # scenes the kernels read from HBM / L2, not from an LDS copy


# what the kernels are compiled from (host-side files do not make a profile stale)
DEVICE_SOURCES = ("p3d_kernels.hip", "p3d_shade.h", "p3d_traverse.h", "p3d_device_math.h", "p3d_powf.h", "p3d_device_types.h",
                  "bvh_device.hip", "pt_kernels.hip")


def kernel_source_digest():
    """sha256 over the kernel sources: profiles taken from other kernels are flagged as stale."""
    h = hashlib.sha256()
    d = os.path.join(REPO, "u_4a_2s_p3d_raytracer_template2_amd", "csrc")
    for f in DEVICE_SOURCES:
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def host_threads():
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except Exception:
        return os.cpu_count() or 1


def compare_u8(a, b):
    """rgb8 planes: equal."""
    if a.shape != b.shape:
        return {"values": int(a.size), "match": False, "note": "shapes differ: %s vs %s" % (a.shape, b.shape)}
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    n = int(np.count_nonzero(d))
    mx = int(d.max()) if d.size else 0
    return {"values": int(a.size), "differing": n, "max_level_diff": mx, "match": bool(n == 0)}


def cpu_baseline(scene_file, full_res, cpu_res, depth, spp, budget_s=25.0, full_size_port=True):
    """Frames of the configuration on ONE host core by the reference's own object code (oracle/_ref) at `cpu_res`
    (the full frame for config 2, a bounded sample otherwise); the oracle port (same structure, checked bit for bit
    against it) when the .so did not travel.  Plus the oracle with the fall-through removed on all cores at FULL
    size.  Checker-only code: never on the product path.
    Returns (cpu_baseline object, frames) with frames = {"sample": rgb8 at cpu_res + its ray count + who rendered it,
    "full": rgb8 at full_res by the all-core port, or None}."""
    from oracle import oracle_py as O
    from oracle import ref_py as R
    sc = O.Scene(scene_file)
    sc.set_resolution(*cpu_res)
    times, rays, kind, frame = [], 0, "port", None
    t_start = time.time()
    if R.available(depth):
        kind = "reference"
        rs = R.RefScene.from_oracle_scene(sc, scene_file, res=cpu_res, depth=depth)
        while len(times) < 5 and (time.time() - t_start) < budget_s * 0.6:
            t0 = time.perf_counter()
            r = rs.render(2, spp, 12345)
            times.append(time.perf_counter() - t0)
            rays, frame = r["rays"], r["rgb8"]
            if times[-1] > budget_s * 0.3:
                break
        rs.close()
    else:
        while len(times) < 5 and (time.time() - t_start) < budget_s * 0.6:
            t0 = time.perf_counter()
            r = sc.render(max_depth=depth, accel=2, spp=spp, threads=1, want_f32=False, want_hit=False)
            times.append(time.perf_counter() - t0)
            rays, frame = r["counters"]["rays"], r["rgb8"]
            if times[-1] > budget_s * 0.3:
                break
    st = float(np.median(times))
    who = "the reference's own object code (oracle/_ref)" if kind == "reference" else "the oracle port"
    out = {"value": rays / st / 1e6, "unit": "Mrays/s", "cores": 1, "kind": kind,
           "sample": "%d %s %dx%d depth-%d%s frame(s) on one core by %s (BVH traversal + the brute-force "
                     "fall-through of SURVEY Q1); median %.3f s/frame" % (
                         len(times), "full" if tuple(cpu_res) == tuple(full_res) else "reduced-size", cpu_res[0], cpu_res[1], depth,
                         " spp %d" % spp if spp else "", who, st)}
    frames = {"sample": {"rgb8": frame, "rays": int(rays), "res": tuple(cpu_res), "by": who}, "full": None}
    if full_size_port:
        ncpu = 1 if spp else host_threads()      # spp > 0 consumes libc rand() in pixel order: the port renders it on one thread
        sc.set_resolution(*full_res)
        mt, full, full_rays = [], None, 0
        t_mt = time.time()
        while len(mt) < 3 and (not mt or (time.time() - t_mt) + mt[-1] < budget_s * 0.5):
            t0 = time.perf_counter()
            # spp == 0: fall-through removed + all cores (config 3's reference structure would take 43 minutes at full size);
            # spp > 0: one thread anyway, so the reference's own structure (the BVH result discarded) -- on a
            # 4096^2 sample frame the two differ in a pixel or two
            r = sc.render(max_depth=depth, accel=2, spp=spp, threads=ncpu, break_fixed=0 if spp else 1, want_f32=False, want_hit=False)
            mt.append(time.perf_counter() - t0)
            full, full_rays = r["rgb8"], r["counters"]["rays"]
        m = float(np.median(mt))
        out["multithread"] = {"value": full_rays / m / 1e6, "unit": "Mrays/s", "cores": ncpu, "kind": "port",
                              "note": "oracle port, %s, full %dx%d frame; median %.4f s/frame"
                                      % ("fall-through removed + row blocks over all host cores" if not spp else
                                         "reference structure, one thread (serial rand() stream)", full_res[0], full_res[1], m)}
        frames["full"] = {"rgb8": full, "rays": int(full_rays), "res": tuple(full_res),
                          "by": ("the oracle port with the fall-through removed (bit-identical to the reference's frame "
                                 "where both were run: tests/test_oracle_pinned.py), %d threads" % ncpu) if not spp else
                                "the oracle port in the reference's structure (pinned bit for bit against the reference's object code), one thread"}
    return out, frames


def cpu_baseline_synthetic(n_prims, depth, budget_s=25.0):
    """Synthetic scaling scene: the brute-force fall-through of the reference is O(N) per ray, so the sample is
    a 96x54 frame (1/400 of the pixels) of the same scene.  Only up to 2e5 primitives (.p3f text)."""
    if n_prims > 200000:
        return {"value": None, "unit": "Mrays/s", "cores": 1, "kind": "port",
                "sample": "not run: the oracle loads .p3f text and its reference-structure closest hit is O(N) per ray"}, None
    import tempfile
    from oracle import oracle_py as O
    from u_4a_2s_p3d_raytracer_template2_amd import synthetic as SY
    path = os.path.join(tempfile.mkdtemp(prefix="p3d_bench_"), "synthetic.p3f")
    SY.write_p3f(path, n_prims, 96, 54)
    sc = O.Scene(path)
    t0 = time.perf_counter()
    r = sc.render(max_depth=depth, accel=2, threads=1, want_f32=False, want_hit=False)
    st = time.perf_counter() - t0
    who = "the oracle port (reference structure)"
    return ({"value": r["counters"]["rays"] / st / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
             "sample": "one 96x54 depth-%d frame of the same scene, single thread, reference structure; %.2f s" % (depth, st)},
            {"sample": {"rgb8": r["rgb8"], "rays": int(r["counters"]["rays"]), "res": (96, 54), "by": who}, "full": None})


def frame_check(frames, timed_frame, timed_rays, render_sample):
    """Compare the last timed frame (and, where the CPU leg rendered a reduced-size sample, a GPU frame of that size
    made by render_sample(res) -> (rgb8, rays)) with what the CPU leg rendered.  Returns (bool, details)."""
    out, ok = {}, True
    tol = "rgb8 equal; ray counts equal"
    smp, full = frames.get("sample"), frames.get("full")
    if smp is not None and tuple(smp["res"]) == (timed_frame.shape[1], timed_frame.shape[0]):
        c = compare_u8(timed_frame, smp["rgb8"])
        c.update({"against": smp["by"], "rays_gpu": int(timed_rays), "rays_cpu": smp["rays"], "rays_match": int(timed_rays) == smp["rays"]})
        out["timed_frame"] = c
        ok = ok and c["match"] and c["rays_match"]
        smp = None
    elif full is not None:
        c = compare_u8(timed_frame, full["rgb8"])
        c.update({"against": full["by"], "rays_gpu": int(timed_rays), "rays_cpu": full["rays"], "rays_match": int(timed_rays) == full["rays"]})
        out["timed_frame"] = c
        ok = ok and c["match"] and c["rays_match"]
    else:
        out["timed_frame"] = {"match": None, "note": "no CPU frame of the timed size"}
    if smp is not None:
        g8, grays = render_sample(smp["res"])
        c = compare_u8(g8, smp["rgb8"])
        c.update({"against": smp["by"], "res": list(smp["res"]), "rays_gpu": int(grays), "rays_cpu": smp["rays"], "rays_match": int(grays) == smp["rays"]})
        out["reference_sample"] = c
        ok = ok and c["match"] and c["rays_match"]
    if out["timed_frame"].get("match") is None and "reference_sample" not in out:
        return None, out
    out["tolerance"] = tol
    return bool(ok), out


def _kname(name):
    """'void p3d::k<false, true, 1, 1, false>(p3d::LaunchParams)' -> 'p3d::k<false, true, 1, 1, false>'"""
    name = name.strip()
    if name.startswith("void "):
        name = name[5:]
    if name.endswith(")") and "(" in name:
        name = name[:name.rindex("(")]
    return name.strip()


def load_profile(workload):
    """(entry of profiles/current.json, kernel-stats rows, PMC summary) or Nones."""
    try:
        cur = json.load(open(os.path.join(REPO, "profiles", "current.json")))[workload]
    except Exception:
        return None, None, None
    stats = pmc = None
    try:
        import csv
        rows = list(csv.DictReader(open(os.path.join(REPO, "profiles", cur["kernel_stats"]))))
        stats = [r for r in rows if "p3d::" in r.get("Name", "") or "p3dpt::" in r.get("Name", "")]
    except Exception:
        pass
    try:
        pmc = json.load(open(os.path.join(REPO, "profiles", cur["pmc"])))
    except Exception:
        pass
    return cur, stats, pmc


def roofline_from_profiles(workload, live, profile_key=None):
    """The roofline object: see the module docstring.  `live` = this run's own measurements."""
    cur, stats, pmc = load_profile(profile_key or workload)
    hbm_scene = workload in HBM_SCENE_WORKLOADS
    alg_note = ("SURVEY 8d: 32 B per slab test + 16/48/32/16 B per sphere/triangle/box/plane test + 3 B per pixel, counted by the "
                "counting build; ")
    alg_note += ("the walk is served by L1 / L2 / Infinity Cache, so this is NOT HBM traffic (roofline.hbm is): these kernels wait on "
                 "dependent fetches" if hbm_scene else
                 "the scene is served from LDS/L2, so this is NOT HBM traffic and HBM is not the binding resource")
    out = {"bound": "fetch_latency" if hbm_scene else "valu_issue", "achieved": None,
           "peak": N_SIMD * CLOCK_GHZ / VALU_CYCLES_PER_INST, "unit": "Gwave-instr/s",
           "frac": None, "frac_of": "vector-instruction issue (one wave64 VALU instruction per 2 cycles per SIMD)",
           "traffic": None, "kernel": live.get("kernel"), "kernel_ms_live": live.get("kernel_ms"), "kernel_ms_live_of": live.get("kernel"),
           "frame_device_ms_live": live.get("frame_ms"),
           "algorithmic": {"bytes_per_frame": live.get("alg_bytes"), "GBps": live.get("alg_gbps"),
                           "frac_of_hbm_peak": None if live.get("alg_gbps") is None else live["alg_gbps"] / HBM_PEAK_GBS,
                           "note": alg_note},
           "source": None, "stale": None}
    if not (cur and stats and pmc):
        out["note"] = "no profile registered in profiles/current.json for this workload: only live HIP-event timings"
        return out
    out["source"] = {"kernel_stats": "profiles/" + cur["kernel_stats"], "pmc": "profiles/" + cur["pmc"],
                     "issue_rates": "profiles/r02_valu_rate_ubench.txt"}
    if pmc.get("kernel_source_digest") is None:
        out["stale"] = None
    elif pmc["kernel_source_digest"] != kernel_source_digest():
        out["stale"] = True
    else:
        out["stale"] = False
    # the ray kernel with the most device time in the kernel stats of this workload
    def total_ns(r):
        return float(r.get("TotalDurationNs") or r.get("TotalDuration(ns)") or 0.0)
    ray = [r for r in stats if "<true" not in r["Name"]] or stats        # not the counting builds (first template argument)
    # ... of the schedule this run used (a profile run may hold frames of other schedules too: measured picks, comparisons)
    mine = {"tree": ("whitted_tree_kernel",), "tile": ("wf_tile_kernel",),
            "wavefront": ("wf_primary_kernel", "wf_secondary_kernel", "wf_resolve_kernel")}.get(live.get("schedule"), ())
    top = max([r for r in ray if any(m in r["Name"] for m in mine)] or ray, key=total_ns)
    kernels = {}
    for r in ray:
        name = r["Name"]
        e = None
        for k, v in pmc.get("kernels", {}).items():
            if _kname(k) == _kname(name):
                e = v
        calls = float(r.get("Calls") or 1)
        avg_us = total_ns(r) / calls / 1e3
        k = {"calls": int(calls), "avg_us": avg_us, "share_of_device_time": float(r.get("Percentage") or 0.0) / 100.0}
        if e and e.get("SQ_INSTS_VALU") and e.get("GRBM_GUI_ACTIVE"):
            cyc = e["GRBM_GUI_ACTIVE"] / 8.0                      # summed over the 8 XCDs (MI355X_MICROARCH.md)
            k.update({
                "valu_issue_frac": e["SQ_INSTS_VALU"] * VALU_CYCLES_PER_INST / (N_SIMD * cyc),
                "valu_busy_frac": e["SQ_ACTIVE_INST_VALU"] * 4.0 / (N_SIMD * cyc) if e.get("SQ_ACTIVE_INST_VALU") else None,
                "salu_issue_frac": e["SQ_INSTS_SALU"] / (N_CU * cyc) if e.get("SQ_INSTS_SALU") else None,
                "valu_per_wave": e["SQ_INSTS_VALU"] / e["SQ_WAVES"] if e.get("SQ_WAVES") else None,
                "salu_per_wave": e["SQ_INSTS_SALU"] / e["SQ_WAVES"] if e.get("SQ_WAVES") and e.get("SQ_INSTS_SALU") else None,
                "lane_utilisation": e.get("valu_lane_utilisation"),
                "hbm_bytes_per_launch": e.get("hbm_bytes_per_launch_corrected"),
                "kernel_cycles": cyc,
            })
            if e.get("SQ_WAVE_CYCLES"):
                # what a kernel that waits on fetches is bound by.  SQ_WAVE_CYCLES / SQ_WAIT_ANY count in units of 4 cycles
                # (calibrated on the persistent tile kernel of config 4: 4 resident waves per SIMD read 0.85 x 4 that way)
                f = {"wait_frac": e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"] if e.get("SQ_WAIT_ANY") else None,
                     "waves_per_simd_avg": e["SQ_WAVE_CYCLES"] * 4.0 / (N_SIMD * cyc),
                     "vmem_per_wave": e["SQ_INSTS_VMEM"] / e["SQ_WAVES"] if e.get("SQ_INSTS_VMEM") and e.get("SQ_WAVES") else None}
                if e.get("TCC_REQ_sum"):
                    f["l2_hit"] = e["TCC_HIT_sum"] / max(e["TCC_HIT_sum"] + e["TCC_MISS_sum"], 1.0)
                if e.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
                    f["l1_hit"] = 1.0 - e["TCP_TCC_READ_REQ_sum"] / e["TCP_TOTAL_CACHE_ACCESSES_sum"]
                    f["l1_accesses_per_vmem"] = e["TCP_TOTAL_CACHE_ACCESSES_sum"] / e["SQ_INSTS_VMEM"] if e.get("SQ_INSTS_VMEM") else None
                    f["l1_pending_stall_frac"] = e["TCP_PENDING_STALL_CYCLES_sum"] / (N_CU * cyc)
                    f["l2_read_GBps"] = e["TCP_TCC_READ_REQ_sum"] * 64.0 / (avg_us * 1e-6) / 1e9      # 64-B requests
                    f["l2_frac_of_peak"] = f["l2_read_GBps"] / L2_PEAK_GBS
                    if e.get("TCP_TCC_READ_REQ_LATENCY_sum") and e.get("TCP_TCC_READ_REQ_sum"):
                        f["l2_read_latency_cycles"] = e["TCP_TCC_READ_REQ_LATENCY_sum"] / e["TCP_TCC_READ_REQ_sum"]
                if k["hbm_bytes_per_launch"]:
                    f["fabric_GBps"] = k["hbm_bytes_per_launch"] / (avg_us * 1e-6) / 1e9
                    f["fabric_frac_of_hbm_peak"] = f["fabric_GBps"] / HBM_PEAK_GBS
                k["fetch"] = f
        kernels[_kname(name)] = k
    out["kernels"] = kernels
    tk = kernels[_kname(top["Name"])]
    out["kernel"] = _kname(top["Name"])
    out["kernel_us_profile"] = tk["avg_us"]
    if tk.get("valu_issue_frac") is not None:
        out["frac"] = min(tk["valu_issue_frac"], 1.0)
        out["achieved"] = out["frac"] * out["peak"]
        out["traffic"] = tk.get("hbm_bytes_per_launch")
        if out["traffic"]:
            out["hbm"] = {"GBps": out["traffic"] / (tk["avg_us"] * 1e-6) / 1e9,
                          "frac_of_hbm_peak": out["traffic"] / (tk["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                          "note": "PMC FETCH_SIZE x2 + WRITE_SIZE per launch (gfx950 correction, separate passes)"
                                  + (": node / primitive fetches that miss L2 + frame buffer" if hbm_scene else ": frame buffer + ray / node queues")}
    # the whole frame at the rate the timed region ran it: every ray kernel's vector instructions (PMC, per launch) x its
    # launches per frame (kernel stats) over this run's time per frame with its frames in flight
    first = {"tree": "whitted_tree_kernel", "tile": "wf_tile_kernel", "wavefront": "wf_primary_kernel"}.get(live.get("schedule"))
    frames_in_profile = sum(float(r.get("Calls") or 0) for r in ray if first and first in r["Name"])
    if frames_in_profile and live.get("frame_ms_in_flight"):
        tot, tot_s, missing = 0.0, 0.0, False
        for r in ray:
            if not any(m in r["Name"] for m in mine):
                continue
            e = None
            for k2, v2 in pmc.get("kernels", {}).items():
                if _kname(k2) == _kname(r["Name"]):
                    e = v2
            if not (e and e.get("SQ_INSTS_VALU")):
                missing = True
                continue
            tot += e["SQ_INSTS_VALU"] * float(r.get("Calls") or 0) / frames_in_profile
            tot_s += (e.get("SQ_INSTS_SALU") or 0.0) * float(r.get("Calls") or 0) / frames_in_profile
        if tot and not missing:
            ach = tot / (live["frame_ms_in_flight"] * 1e-3) / 1e9
            out["whole_frame"] = {"valu_wave_instr_per_frame": tot, "ms_per_frame": live["frame_ms_in_flight"],
                                  "achieved": ach, "peak": out["peak"], "unit": out["unit"], "frac": ach / out["peak"],
                                  "note": "all launches of a frame (SQ_INSTS_VALU per launch x launches per frame, from the committed profile) over "
                                          "this run's time per frame with %s frame(s) in flight: what the timed region as a whole makes of the "
                                          "vector-issue roof; the per-kernel figures above are one frame alone" % live.get("frames_in_flight", "?")}
            # vector AND scalar instructions share a SIMD's issue: interleaved 2:1 in one wave they cost 2.13 ticks each where a
            # vector instruction alone costs 1.68 (profiles/r02_valu_rate_ubench.txt, 4 waves per SIMD; 1.68 ticks = 2 cycles),
            # i.e. 3.6 cycles per scalar instruction beside 2 per vector instruction
            cyc = tot * VALU_CYCLES_PER_INST + tot_s * SALU_CYCLES_PER_INST_MIXED
            floor_ms = cyc / N_SIMD / (CLOCK_GHZ * 1e9) * 1e3
            out["whole_frame"]["issue_mix"] = {"salu_wave_instr_per_frame": tot_s, "cycles_per_valu": VALU_CYCLES_PER_INST,
                                               "cycles_per_salu_interleaved": SALU_CYCLES_PER_INST_MIXED, "floor_ms_per_frame": floor_ms,
                                               "frac": floor_ms / live["frame_ms_in_flight"],
                                               "note": "this run's time per frame against the time the frame's vector + scalar instructions need at "
                                                       "the measured interleaved issue rates: how close the timed region is to running out of issue slots"}
    if tk.get("valu_issue_frac") is not None:
        if hbm_scene and tk.get("fetch"):
            out["fetch_latency"] = dict(tk["fetch"], lanes_active=tk.get("lane_utilisation"), valu_issue_frac=tk.get("valu_issue_frac"),
                                        salu_issue_frac=tk.get("salu_issue_frac"),
                                        note="what binds a kernel that reads the scene from HBM / L2: the dependent fetch chain of the walk "
                                             "(lanes active, share of wave-cycles spent waiting, cache hit rates, L2 and fabric rates against "
                                             "their peaks); neither the vector-issue nor the HBM roof is near")
    return out


def bench_pathtracer(args, torch, dist, P, rank, world, local_rank, dev, comm):
    """BASELINE config 5: the Shadertoy path tracer, 1920x1080, --spp samples per pixel per step.
    N > 1: rank r traces samples r, r+N, ... and the linear sums are added up on rank 0 with ONE
    ncclReduce through the C-ABI (p3d_pt_reduce_sum)."""
    W, H = 1920, 1080
    spp = args.spp
    if spp % world:
        raise SystemExit("--spp must be a multiple of --gpus")
    pt = P.PathTracer(device=local_rank)
    pt.set_stream(torch.cuda.current_stream().cuda_stream)
    lin = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)

    def step():
        pt.render_device(0, lin.data_ptr(), W, H, spp // world, first_frame=rank, frame_stride=world)
        if world > 1:
            comm.pt_reduce_sum(pt, lin.data_ptr(), lin.numel())

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt_max = float(tt.item())
    if rank == 0:
        img = torch.nan_to_num(lin / spp).clamp(0, None).pow(1 / 2.2)
        line = {
            "metric": "Msamples/s (paths) + ms/image @1920x1080 %d spp" % spp,
            "value": W * H * spp * args.steps / dt_max / 1e6, "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32",
            "data": "synthetic: the procedural scene of GPU_PathTracer_template/P3D_RT.glsl (no mouse, iTime = k/60)",
            "config": {"workload": "P3D_RT.glsl path tracer 1920x1080 %d spp (BASELINE config 5)" % spp,
                       "parallelism": "1 GPU" if world == 1 else "%d GPUs: samples split, one ncclReduce(sum) to rank 0 through the C-ABI" % world,
                       "image_mean": float(img.mean().item())},
            "roofline": None,
        }
        if world == 1 and spp == 256:
            # one launch per step: the step IS the kernel (HIP-event bracket = the barrier-to-barrier time of the timed region)
            rl = roofline_from_profiles("pathtracer", {"kernel": "p3dpt::pt_frames_kernel", "kernel_ms": dt_max / args.steps * 1e3,
                                                       "frame_ms": dt_max / args.steps * 1e3})
            rl["algorithmic"] = {"note": "the path tracer's scene is procedural (hashed per ray, PT/P3D_RT.glsl:88-178): there is no scene data to read; "
                                         "HBM sees the running sums only (12 B per pixel per launch)"}
            line["roofline"] = rl
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle_py as O
            w, h, n = 240, 135, 4
            t = time.perf_counter()
            O.pt_render(w, h, n, threads=1, want_sum=False)
            el = time.perf_counter() - t
            line["cpu_baseline"] = {"value": w * h * n / el / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
                                    "sample": "%dx%d, %d samples per pixel, single thread: %.1f s" % (w, h, n, el)}
        emit_line(line)
    pt.close()


_JSON_FD = None


def emit_line(line):
    """The one JSON line, on the process's original stdout."""
    data = (json.dumps(line) + "\n").encode()
    sys.stdout.flush()
    if _JSON_FD is None:
        sys.stdout.write(data.decode()); sys.stdout.flush()
    else:
        os.write(_JSON_FD, data)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["config2", "config3", "config4", "pathtracer", "synthetic"], default="config2")
    ap.add_argument("--frames-per-step", type=int, default=0, help="frames of a step (default: 12, config3: 4, config4: 1)")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="independent frames overlapped on separate HIP streams (each with its own scene handle and "
                         "workspace); 1 = strictly one frame after the other; default min(frames per step, 3 x N)")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the frames of a step as ONE captured HIP graph (auto = off: measured slower than eager launches)")
    ap.add_argument("--gather", choices=["pipelined", "sync"], default="pipelined",
                    help="N > 1: two tile-buffer sets, so that a step's gather overlaps the next step's rendering, or one")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU leg (and with it frame_matches_reference)")
    ap.add_argument("--prims", type=int, default=1000000, help="synthetic workload: number of primitives")
    ap.add_argument("--schedule", choices=["default", "wavefront", "tree", "tile"], default="default",
                    help="force a kernel schedule (default: the library's measured choice)")
    ap.add_argument("--tune", choices=["on", "off"], default="on",
                    help="scenes read from HBM: let the library rank its schedules with the bench's frames in flight (p3d_tune_schedule)")
    ap.add_argument("--spp", type=int, default=256, help="pathtracer workload: samples (frames) per step")
    args = ap.parse_args()

    # stdout carries ONE line, the JSON: anything the libraries print on the way (the host loader echoes the camera like
    # the reference does) goes to stderr
    global _JSON_FD
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)

    # the host driver only supports dmabuf IPC: without this RCCL cannot share buffers between the ranks' processes
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from conftest import scene_path
    import u_4a_2s_p3d_raytracer_template2_amd as P
    from u_4a_2s_p3d_raytracer_template2_amd import multigpu as MG

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # Rehearsal switch for a ONE-GPU box (RCCL refuses two ranks on one device): P3D_BENCH_REHEARSAL=1 puts every
    # rank on cuda:0 and moves the tile buffers through host memory with gloo instead of p3d_gather.  It exercises
    # the sharding / buffer rotation / de-interleave code of the N > 1 path, not xGMI; its numbers mean nothing.
    rehearsal = os.environ.get("P3D_BENCH_REHEARSAL") == "1" and world > 1
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm = None
    if world > 1:
        # control plane only: the communicator id, barriers and the timing reduction go through gloo;
        # every byte of image data moves through p3d_gather (RCCL) below
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ident = [P.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)
        assert isinstance(ident[0], bytes) and len(ident[0]) == 128
        comm_error = None
        if not rehearsal:
            try:
                comm = P.Comm.create(ident[0], rank, world, local_rank)
            except Exception as e:                          # reported in the JSON line; the run goes on through host memory
                comm_error = "%s: %s" % (type(e).__name__, e)
        failed = torch.tensor([1 if comm_error else 0])
        dist.all_reduce(failed, op=dist.ReduceOp.MAX)
        if int(failed.item()):
            # safety net, not a product path: some rank could not build the RCCL communicator.  Every rank then moves
            # its tile buffers through host memory with gloo (like the rehearsal mode) and the line says so.
            if comm is not None:
                comm.close()
                comm = None
            errs = [None] * world
            dist.all_gather_object(errs, comm_error)
            comm_error = next((e for e in errs if e), "unknown")
            if args.workload == "pathtracer":
                raise SystemExit("p3d_comm_create failed: " + comm_error)
    host_gather = world > 1 and comm is None

    if args.workload == "pathtracer":
        if rehearsal:
            raise SystemExit("the pathtracer workload has no rehearsal mode")
        bench_pathtracer(args, torch, dist, P, rank, world, local_rank, dev, comm)
        if world > 1:
            dist.barrier()
            comm.close()
            dist.destroy_process_group()
        return

    wl = WORKLOADS[args.workload]
    synthetic = args.workload == "synthetic"
    W, H = wl["res"]
    depth, spp = wl["depth"], wl["spp"]
    sched = {"wavefront": {"wavefront": True}, "tree": {"tree": True}, "tile": {"tile": True}}.get(args.schedule, {})
    if synthetic:
        import tempfile
        from u_4a_2s_p3d_raytracer_template2_amd import synthetic as SY, api as API
        scene_file = None
        cam_file = os.path.join(tempfile.mkdtemp(prefix="p3d_bench_"), "camera.p3f")
        hs = P.HostScene(SY.camera_p3f(cam_file, W, H))
        cam = hs.camera()
        desc, keep = API.make_desc(*SY.arrays(args.prims))
        make_handle = lambda: P.DeviceScene(desc, device=local_rank, keepalive=keep)
    else:
        scene_file = scene_path(wl["scene"])
        hs = P.HostScene(scene_file)
        hs.set_resolution(W, H)
        cam = hs.camera()
        make_handle = lambda: P.DeviceScene.from_host(hs, device=local_rank)
    B = args.frames_per_step if args.frames_per_step > 0 else wl["frames"]
    # The deeper tree levels of one 1080p frame are too few rays to fill 256 CUs, so consecutive frames --
    # independent work -- are overlapped on F streams, one scene handle each.  With the frame tiled over N GPUs
    # each rank holds 1/N of every frame, so it takes N times as many frames in flight to fill it.
    # Measured (tools/shard_probe.py, explicit non-blocking streams): whole frame 4 in flight 0.072 ms/frame (3: 0.075,
    # 6: 0.077, 12: 0.074); half a frame 4: 0.038-0.042; a quarter / an eighth: 4, 8 and 12 within 3 % of each other.
    F = args.frames_in_flight if args.frames_in_flight > 0 else (4 if world <= 2 else 12)
    F = max(1, min(F, B))
    main_stream = torch.cuda.current_stream()
    # every frame stream is an explicit non-blocking stream; the (default) main stream only forks / joins them
    streams = [torch.cuda.Stream(device=dev) for _ in range(F)]
    handles = []
    for st in streams:
        h = make_handle()
        h.set_stream(st.cuda_stream)
        handles.append(h)
    ds = handles[0]
    rows = H if world == 1 else P.local_rows(H, ROW_BLOCK, world)
    samples_dev = None
    if spp:
        smp = hs.samples(12345, spp)                       # the reference's libc rand() stream, seed of the survey
        samples_dev = torch.from_numpy(smp).to(dev)        # uploaded ONCE (P3D_FLAG_DEVICE_SAMPLES)
        del smp
    kw = dict(max_depth=depth, accel=P.ACCEL_BVH, spp=spp, rank=rank, world=world, row_block=ROW_BLOCK,
              samples_ptr=samples_dev.data_ptr() if spp else 0)

    # HBM-resident outputs: B compact tile buffers per step (also the gather source); two sets when the gather
    # of step i overlaps the rendering of step i + 1
    nbuf = 2 if (world > 1 and args.gather == "pipelined") else 1
    tile_sets = [torch.zeros((B, rows, W, 3), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    tile_bytes = rows * W * 3
    gathered_sets = frames = gh = comm_stream = None
    if world > 1:
        comm_stream = torch.cuda.Stream(device=dev)
        gh = make_handle()                                  # the handle whose stream carries gather + de-interleave
        gh.set_stream(comm_stream.cuda_stream)
        if rank == 0:
            gathered_sets = [torch.zeros((world, B, rows, W, 3), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
            frames = torch.zeros((B, H, W, 3), dtype=torch.uint8, device=dev)

    # work counters of this rank's share of one frame (counting build, same traversal)
    ds.render_device(cam, rgb8_ptr=tile_sets[0][0].data_ptr(), counters=True, **sched, **kw)
    ctr = ds.counters()
    my_rays = ctr["rays"]
    alg_bytes = ctr["algorithmic_bytes"] + 3 * ctr["pixels"]          # + rgb8 written per pixel

    def render_frames(lead, buf, join=True):
        """The B frames of a step into tile set `buf`: fork the side streams off `lead`, enqueue, join back.
        join=False (one GPU, nothing downstream of a step on the device): the F streams simply run on -- the frames of
        consecutive steps are independent, and the barrier around the timed region waits for every stream."""
        if join:
            for k in range(F):
                streams[k].wait_stream(lead)
        for f in range(B):
            handles[f % F].render_device(cam, rgb8_ptr=tile_sets[buf][f].data_ptr(), **sched, **kw)
        if join:
            for k in range(F):
                lead.wait_stream(streams[k])

    # let every handle settle its schedule ALONE: for scenes read from HBM the library times every schedule on the
    # first frames of a configuration, and frames running next to them on other streams would falsify the timing
    for h in handles:
        for _ in range(8):
            h.render_device(cam, rgb8_ptr=tile_sets[0][0].data_ptr(), **sched, **kw)
        h.sync()
    torch.cuda.synchronize()
    # ... and then has the library rank the same candidates the way the timed region runs them: all F handles in flight
    # (p3d_tune_schedule; one frame at a time the dragon's ranking is tile 0.84 / tree 0.90 ms, with four in flight
    # tree 0.59 / tile 0.63 ms per frame: profiles/r03_exp15_25_libm_powf.txt)
    tuned = None
    if not sched and args.workload in HBM_SCENE_WORKLOADS and F > 1 and args.tune == "on":
        best, ms6 = P.tune_schedule(handles, cam, [tile_sets[0][k % B].data_ptr() for k in range(F)], frames=3, **kw)
        if best >= 0:
            tuned = {"best": ["wavefront", "tree", "tile"][best % 3] + (" / shared walks" if best < 3 else " / private walks"),
                     "ms_per_frame_in_flight": {("%s / %s" % (["wavefront", "tree", "tile"][k % 3], "shared" if k < 3 else "private")): round(v, 4)
                                                for k, v in enumerate(ms6) if v >= 0}}
    graphs = None
    # eager launches by default: replaying a step as one captured graph measured SLOWER on a rank's share of a tiled
    # frame (tools/shard_probe.py, 1/8 of config 2: 0.025-0.029 ms/frame eager, 0.033-0.035 as a graph; whole frame:
    # 0.073-0.076 vs 0.076-0.079)
    want_graph = args.graph == "on"
    if want_graph:
        try:
            captured = []
            for buf in range(nbuf):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    render_frames(torch.cuda.current_stream(), buf)
                g.replay()
                torch.cuda.synchronize()
                captured.append(g)
            graphs = captured
        except Exception as exc:                     # capture not possible here: keep launching eagerly
            if args.graph == "on":
                raise
            if rank == 0:
                print("bench.py: HIP graph capture unavailable (%s); launching eagerly" % exc, file=sys.stderr)

    state = {"next": 0}
    free_ev = [None] * nbuf                               # recorded on the comm stream when a set's gather is done

    def step():
        buf = state["next"]
        state["next"] = (buf + 1) % nbuf
        if free_ev[buf] is not None:
            main_stream.wait_event(free_ev[buf])          # the set is still on the wire from two steps ago
        if graphs is not None:
            graphs[buf].replay()
        else:
            render_frames(main_stream, buf, join=world > 1)
        if world > 1:
            comm_stream.wait_stream(main_stream)
            if host_gather:
                MG.gather_tiles_through_host(tile_sets[buf], gathered_sets[buf] if rank == 0 else None, dist, rank, world,
                                             torch, comm_stream)
            else:
                comm.gather(gh, tile_sets[buf].data_ptr(), gathered_sets[buf].data_ptr() if rank == 0 else 0, B * tile_bytes)
            if rank == 0:                                 # one launch restores the row order of all B frames
                gh.deinterleave_frames(gathered_sets[buf].data_ptr(), frames.data_ptr(), W, H, ROW_BLOCK, world, 3, B,
                                       rank_stride_bytes=B * tile_bytes, tile_stride_bytes=tile_bytes)
            ev = torch.cuda.Event()
            ev.record(comm_stream)
            free_ev[buf] = ev

    def barrier():
        torch.cuda.synchronize()                          # all streams of this rank, the comm stream included
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0

    # single-frame latency (one frame in flight) and the dominant kernel's duration: HIP events on the launch stream
    nl = max(min(args.steps * B, 200), 8)
    frame_ms_sum = kern_ms_sum = 0.0
    for _ in range(nl):
        ds.render_device(cam, rgb8_ptr=tile_sets[0][0].data_ptr(), profile=True, **sched, **kw)
        f_ms, k_ms = ds.profile()
        frame_ms_sum += f_ms
        kern_ms_sum += k_ms
    frame_dev_ms, kern_ms = frame_ms_sum / nl, kern_ms_sum / nl
    chosen = ds.last_schedule()

    tt = torch.tensor([dt], dtype=torch.float64)
    rr = torch.tensor([float(my_rays)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
    dt_max = float(tt.item())
    rays_frame = float(rr.item())

    if rank == 0:
        final = (frames[B - 1] if world > 1 else tile_sets[(state["next"] - 1) % nbuf][B - 1][:H]).cpu().numpy()
        total_rays = rays_frame * B * args.steps
        dominant = {"tree": "p3d::whitted_tree_kernel", "tile": "p3d::wf_tile_kernel", "wavefront": "p3d::wf_primary_kernel"}[chosen]
        live = {"kernel": dominant, "schedule": chosen, "kernel_ms": kern_ms, "frame_ms": frame_dev_ms, "alg_bytes": int(alg_bytes),
                "alg_gbps": alg_bytes / (frame_dev_ms * 1e-3) / 1e9, "frames_in_flight": F,
                "frame_ms_in_flight": dt_max / args.steps / B * 1e3 if world == 1 else None}
        gather_verified = None
        if world > 1:
            # the gathered, de-interleaved frame against a one-GPU render of the same frame on this rank
            whole = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
            kw1 = dict(kw, rank=0, world=1)
            ds.render_device(cam, rgb8_ptr=whole.data_ptr(), **sched, **kw1)
            ds.sync()
            gather_verified = bool(np.array_equal(final, whole.cpu().numpy()))
        if synthetic:
            data = "synthetic: %d random spheres+triangles (SURVEY 8d scaling scene, seed 2024), mount_low camera" % args.prims
        else:
            data = ("P3D_Scenes/%s.p3f, the reference's own scene asset (%d primitives, %d light(s)); resolution / accel / depth%s "
                    "overridden to the configuration" % (wl["scene"], hs.n_prims, hs.n_lights,
                                                          " / spp (host libc rand() sample stream, seed 12345)" if spp else ""))
        line = {
            "metric": "Mrays/s + ms/frame @%dx%d depth%d" % (W, H, depth),
            "value": total_rays / dt_max / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "ms_per_frame": dt_max / args.steps / B * 1e3,
            "ms_per_frame_latency": frame_dev_ms,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": data,
            "config": {"workload": ("SURVEY 8d scaling scene, %d primitives, 1920x1080 depth 4 BVH" % args.prims) if synthetic else wl["name"],
                       "frames_per_step": B, "frames_in_flight": F, "hip_graph": graphs is not None,
                       "schedule": chosen + (" (forced)" if sched else
                                             (" (library: measured with %d frames in flight, p3d_tune_schedule)" % F) if tuned else " (library default)"),
                       "schedule_tuning": tuned,
                       "gather": None if world == 1 else
                                 ("FALLBACK gloo through host memory: " + ("rehearsal mode" if rehearsal else "p3d_comm_create failed (%s)" % comm_error))
                                 if host_gather else
                                 ("p3d_gather over RCCL on a communication stream, %s"
                                  % ("two buffer sets (overlaps the next step)" if nbuf == 2 else "one buffer set")),
                       "rays_per_frame": int(rays_frame),
                       "row_block": ROW_BLOCK,
                       "parallelism": "1 GPU" if world == 1 else "%d GPUs: interleaved 16-row blocks + one RCCL gather to rank 0 through the C-ABI" % world,
                       "frame_checksum": int(final.astype(np.uint64).sum())},
            "roofline": roofline_from_profiles(args.workload, live, "synthetic_%d" % args.prims if synthetic else None),
            "env": {"lib": os.path.relpath(P.api.LIB_PATH, REPO), "P3D": {k: v for k, v in sorted(os.environ.items()) if k.startswith("P3D_")}},
        }
        if world > 1:
            line["gather_verified"] = gather_verified
        if args.workload == "config3" and world == 1:
            # the exact shortcut of DESIGN.md 12 as a SECOND figure: triangles no ray can hit left out of the BVH
            hc = P.DeviceScene.from_host(hs, device=local_rank, cull_never_hit=True)
            hc.set_stream(streams[0].cuda_stream)
            chk = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
            for _ in range(8):
                hc.render_device(cam, rgb8_ptr=chk.data_ptr(), **kw)
            hc.sync()
            ms = 0.0
            for _ in range(20):
                hc.render_device(cam, rgb8_ptr=chk.data_ptr(), profile=True, **kw)
                ms += hc.profile()[0]
            line["config"]["cull_never_hit"] = {
                "ms_per_frame_latency": ms / 20, "Mrays_per_s": rays_frame / (ms / 20) / 1e3, "schedule": hc.last_schedule(),
                "triangles_left_out": hc.stats()["n_culled"], "frame_equals_default_build": bool(np.array_equal(chk.cpu().numpy(), final)),
                "note": "p3d_build_opts::cull_never_hit: same image, same ray count; NOT the headline (the default build traverses every triangle, as the reference does)"}
            hc.close()
        if world == 1 and not args.no_cpu_baseline:
            if synthetic:
                line["cpu_baseline"], ref_frames = cpu_baseline_synthetic(args.prims, depth)
            else:
                line["cpu_baseline"], ref_frames = cpu_baseline(scene_file, (W, H), wl.get("cpu_res", (W, H)), depth, spp)

            def render_sample(res):
                """A frame of the CPU leg's sample size through the same handle: (rgb8, rays)."""
                if synthetic:
                    cs = P.HostScene(SY.camera_p3f(cam_file, res[0], res[1]))
                    c2, smp2 = cs.camera(), None
                else:
                    cs = P.HostScene(scene_file)
                    cs.set_resolution(res[0], res[1])
                    c2, smp2 = cs.camera(), (cs.samples(12345, spp) if spp else None)
                r = ds.render(c2, max_depth=depth, accel=P.ACCEL_BVH, spp=spp, samples=smp2, counters=True, want_f32=False, want_hit=False, **sched)
                return r["rgb8"], r["counters"]["rays"]

            if ref_frames is not None:
                ok, details = frame_check(ref_frames, final, int(rays_frame), render_sample)
                line["frame_matches_reference"] = ok
                line["frame_check"] = details
            else:
                line["frame_matches_reference"] = None
                line["frame_check"] = {"note": "no CPU frame for this size of the synthetic scene"}
        emit_line(line)
    if world > 1:
        dist.barrier()
        gh.close()
        if comm is not None:
            comm.close()
        dist.destroy_process_group()
    for h in handles:
        h.close()


if __name__ == "__main__":
    main()
