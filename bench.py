#!/usr/bin/env python3
"""bench.py -- Mrays/s of the Whitted hot path on BASELINE config 2
(mount_low.p3f, 1920x1080, depth 4, BVH), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = `--frames-per-step` frames of the configuration's camera, each rendered by the HIP
kernel into HBM-resident buffers.  With N > 1 every frame is split into interleaved 16-row
blocks across the ranks (total work fixed: strong scaling) and each step ends with ONE RCCL
gather of the compact tile buffers to rank 0 plus the de-interleave kernel there.  The gather of
step i is waited for while step i+1 renders into a second set of tile buffers (--gather sync waits
at once); every step's frames are on rank 0 when the timed region closes.
value = rays of all frames / max-over-ranks wall time; a ray is one closest-hit or one shadow
query, counted by the counting build of the same kernel on the same frame.

A frame is the wavefront schedule: wf_primary_kernel (camera rays + their shadow rays +
shading), one wf_secondary_kernel per deeper tree level, and the resolve passes.

Extra objects on the JSON line: "roofline" for the dominant kernel wf_primary_kernel
(algorithmic bytes of ITS launch -- counted by the counting build run at depth 1, which is
exactly the level-1 work -- divided by its mean duration from HIP events on the launch
stream, vs the 8 TB/s HBM peak; traffic = PMC-measured HBM bytes per launch from
profiles/) and, at N=1, "cpu_baseline" (the CPU oracle timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
RES = (1920, 1080)
MAX_DEPTH = 4
ROW_BLOCK = 16


def cpu_baseline(scene_file, budget_s=20.0):
    """The oracle (CPU restatement of the reference, incl. the brute-force fall-through of
    SURVEY Q1) single-threaded on full config-2 frames, plus the break-fixed multi-threaded
    variant as the honest strong baseline.  Checker-only code: never on the product path."""
    from oracle import oracle_py as O
    sc = O.Scene(scene_file)
    sc.set_resolution(*RES)
    times, rays = [], 0
    t_start = time.time()
    while len(times) < 5 and (time.time() - t_start) < budget_s * 0.6:
        t0 = time.perf_counter()
        r = sc.render(max_depth=MAX_DEPTH, accel=2, threads=1, want_f32=False, want_hit=False)
        times.append(time.perf_counter() - t0)
        rays = r["counters"]["rays"]
    st = float(np.median(times))
    ncpu = os.cpu_count() or 1
    mt_times = []
    while len(mt_times) < 5 and (time.time() - t_start) < budget_s:
        t0 = time.perf_counter()
        sc.render(max_depth=MAX_DEPTH, accel=2, threads=ncpu, break_fixed=1, want_f32=False, want_hit=False)
        mt_times.append(time.perf_counter() - t0)
    out = {"value": rays / st / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
           "sample": "%d full 1920x1080 depth-4 frames, single thread, reference structure "
                     "(BVH traversal + brute-force fall-through); median %.3f s/frame" % (len(times), st)}
    if mt_times:
        mt = float(np.median(mt_times))
        out["multithread"] = {"value": rays / mt / 1e6, "unit": "Mrays/s", "cores": ncpu,
                              "note": "fall-through removed + row blocks over all host cores; median %.4f s/frame" % mt}
    return out


def cpu_baseline_synthetic(n_prims, budget_s=25.0):
    """Synthetic scaling scene: the brute-force fall-through of the reference is O(N) per ray, so the
    sample is a 96x54 frame (1/400 of the pixels) of the same scene, single-threaded with the reference
    structure, plus the break-fixed all-core variant at 480x270.  Only up to 2e5 primitives (the oracle
    reads .p3f text)."""
    if n_prims > 200000:
        return {"value": None, "unit": "Mrays/s", "cores": 1, "kind": "port",
                "sample": "not run: the oracle loads .p3f text and its reference-structure closest hit is O(N) per ray"}
    import tempfile
    from oracle import oracle_py as O
    from u_4a_2s_p3d_raytracer_template2_amd import synthetic as SY
    path = os.path.join(tempfile.mkdtemp(prefix="p3d_bench_"), "synthetic.p3f")
    SY.write_p3f(path, n_prims, 96, 54)
    sc = O.Scene(path)
    t0 = time.perf_counter()
    r = sc.render(max_depth=MAX_DEPTH, accel=2, threads=1, want_f32=False, want_hit=False)
    st = time.perf_counter() - t0
    out = {"value": r["counters"]["rays"] / st / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
           "sample": "one 96x54 depth-4 frame of the same scene, single thread, reference structure; %.2f s" % st}
    if st < budget_s:
        ncpu = os.cpu_count() or 1
        sc.set_resolution(480, 270)
        t0 = time.perf_counter()
        r = sc.render(max_depth=MAX_DEPTH, accel=2, threads=ncpu, break_fixed=1, want_f32=False, want_hit=False)
        mt = time.perf_counter() - t0
        out["multithread"] = {"value": r["counters"]["rays"] / mt / 1e6, "unit": "Mrays/s", "cores": ncpu,
                              "note": "fall-through removed (reference BVH only) + all host cores, 480x270; %.3f s" % mt}
    return out


def bench_pathtracer(args, torch, dist, P, rank, world, local_rank, dev, rehearsal):
    """BASELINE config 5: the Shadertoy path tracer, 1920x1080, --spp samples per pixel per step.
    One step = one converged image.  N > 1: rank r traces samples r, r+N, ... (weak in samples per
    pixel would change the image, so the image is fixed: strong scaling) and the linear sums are
    reduced to rank 0 with one RCCL reduce."""
    W, H = RES
    spp = args.spp
    if spp % world:
        raise SystemExit("--spp must be a multiple of --gpus")
    pt = P.PathTracer(device=local_rank)
    pt.set_stream(torch.cuda.current_stream().cuda_stream)
    lin = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)

    def step():
        pt.render_device(0, lin.data_ptr(), W, H, spp // world, first_frame=rank, frame_stride=world)
        if world > 1:
            if rehearsal:
                torch.cuda.synchronize()
                h = lin.cpu()
                dist.reduce(h, dst=0, op=dist.ReduceOp.SUM)
                if rank == 0:
                    lin.copy_(h)
            else:
                dist.reduce(lin, dst=0, op=dist.ReduceOp.SUM)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    red_dev = torch.device("cpu") if rehearsal else dev
    tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
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
                       "parallelism": "1 GPU" if world == 1 else "%d GPUs: samples split, RCCL sum-reduce to rank 0" % world,
                       "image_mean": float(img.mean().item())},
            "roofline": None,
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle_py as O
            w, h, n = 240, 135, 4
            t = time.perf_counter()
            O.pt_render(w, h, n, threads=1, want_sum=False)
            el = time.perf_counter() - t
            line["cpu_baseline"] = {"value": w * h * n / el / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
                                    "sample": "%dx%d, %d samples per pixel, single thread: %.1f s" % (w, h, n, el)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    pt.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames-per-step", type=int, default=12)
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="independent frames overlapped on separate HIP streams (each with its own scene handle "
                         "and workspace); 1 = strictly one frame after the other; default 3 per GPU-th of a frame "
                         "(3 on one GPU, min(frames-per-step, 3 N) when the frame is tiled over N GPUs)")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the frames of a step as ONE captured HIP graph instead of ~8 launches per frame "
                         "(auto: use it when capture succeeds)")
    ap.add_argument("--gather", choices=["pipelined", "sync"], default="pipelined",
                    help="N > 1: wait for a step's gather (and de-interleave it) while the NEXT step renders into a "
                         "second tile buffer, or right after the step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["config2", "pathtracer", "synthetic"], default="config2",
                    help="config2 = the headline Whitted frame; pathtracer = BASELINE config 5 (P3D_RT.glsl scene, "
                         "1920x1080, --spp samples per step, samples split across ranks + RCCL sum-reduce); "
                         "synthetic = the SURVEY 8d scaling scene (--prims random spheres+triangles, HBM-resident BVH), "
                         "same camera / resolution / depth as config 2")
    ap.add_argument("--prims", type=int, default=1000000, help="synthetic workload: number of primitives")
    ap.add_argument("--schedule", choices=["default", "wavefront", "tree"], default="default",
                    help="force a kernel schedule (default: the library's choice)")
    ap.add_argument("--pmc-json", default=None,
                    help="PMC summary (tools/pmc_summary.py) to take roofline.traffic from; default: the committed one "
                         "for the workload under profiles/")
    ap.add_argument("--spp", type=int, default=256, help="pathtracer workload: samples (frames) per step")
    args = ap.parse_args()

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
    # Rehearsal switch for a ONE-GPU box: P3D_BENCH_REHEARSAL=1 puts every rank on cuda:0 and moves
    # the gather through host memory with gloo (RCCL refuses two ranks on one device).  It exercises
    # the sharding / gather / de-interleave code path, not xGMI; numbers from it mean nothing.
    rehearsal = os.environ.get("P3D_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.workload == "pathtracer":
        return bench_pathtracer(args, torch, dist, P, rank, world, local_rank, dev, rehearsal)

    synthetic = args.workload == "synthetic"
    sched = {"wavefront": {"wavefront": True}, "tree": {"tree": True}}.get(args.schedule, {})
    if synthetic:
        import tempfile
        from u_4a_2s_p3d_raytracer_template2_amd import synthetic as SY, api as API
        scene_file = None
        cam_file = os.path.join(tempfile.mkdtemp(prefix="p3d_bench_"), "camera.p3f")
        cam = P.HostScene(SY.camera_p3f(cam_file, *RES)).camera()
        t_b = time.time()
        desc, keep = API.make_desc(*SY.arrays(args.prims))
        make_handle = lambda: P.DeviceScene(desc, device=local_rank, keepalive=keep)
    else:
        scene_file = scene_path("mount_low")
        hs = P.HostScene(scene_file)
        hs.set_resolution(*RES)
        cam = hs.camera()
        make_handle = lambda: P.DeviceScene.from_host(hs, device=local_rank)
    # The deeper tree levels of one 1080p frame are too few rays to fill 256 CUs (each level launch is
    # bounded by single-wave latency), so consecutive frames -- independent work, exactly like the
    # reference's render-another-image loop -- are overlapped on F streams, one scene handle each.
    # With the frame tiled over N GPUs each rank holds 1/N of every frame, so it takes N times as many
    # frames in flight to fill it.
    F = args.frames_in_flight if args.frames_in_flight > 0 else 3 * world
    F = max(1, min(F, args.frames_per_step))
    main_stream = torch.cuda.current_stream()
    streams = [main_stream] + [torch.cuda.Stream(device=dev) for _ in range(F - 1)]
    handles = []
    for st in streams:
        h = make_handle()
        h.set_stream(st.cuda_stream)
        handles.append(h)
    ds = handles[0]
    W, H = RES
    B = args.frames_per_step
    rows = H if world == 1 else MG.padded_rows(H, ROW_BLOCK, world)

    # HBM-resident outputs: B compact tile buffers per step (double use: gather source).  With the
    # pipelined gather there are two sets: step i renders into set i % 2 while set (i - 1) % 2 is on the wire.
    nbuf = 2 if (world > 1 and args.gather == "pipelined") else 1
    tile_sets = [torch.zeros((B, rows, W, 3), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    tiles = tile_sets[0]
    gathered_sets = frames = None
    if world > 1 and rank == 0:
        gathered_sets = [torch.zeros((world, B, rows, W, 3), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
        frames = torch.zeros((B, H, W, 3), dtype=torch.uint8, device=dev)
    tile_bytes = rows * W * 3

    # work counters of this rank's share of one frame (counting build, same traversal)
    ds.render_device(cam, rgb8_ptr=tiles[0].data_ptr(), max_depth=MAX_DEPTH, accel=P.ACCEL_BVH,
                     rank=rank, world=world, row_block=ROW_BLOCK, **sched, counters=True)
    ctr = ds.counters()
    my_rays = ctr["rays"]
    px_local = ctr["pixels"]
    alg_bytes = ctr["algorithmic_bytes"] + 3 * px_local          # + rgb8 written per pixel

    def render_frames(lead, buf=0):
        """The B frames of a step into tile set `buf`: fork the side streams off `lead`, enqueue, join back."""
        for k in range(1, F):
            streams[k].wait_stream(lead)
        for f in range(B):
            handles[f % F].render_device(cam, rgb8_ptr=tile_sets[buf][f].data_ptr(), max_depth=MAX_DEPTH, accel=P.ACCEL_BVH,
                                         rank=rank, world=world, row_block=ROW_BLOCK, **sched)
        for k in range(1, F):
            lead.wait_stream(streams[k])

    # A step is ~8 launches per frame; at a fraction of a millisecond per step the host's launch rate
    # matters, most of all when N GPUs each hold 1/N of the work.  The frames of a step are captured once
    # into a HIP graph (after an eager step has sized every workspace) and replayed with one launch.
    graphs = None
    # On one GPU the step is GPU-bound and eager launches measured marginally faster (42.6 vs 40.8 Grays/s),
    # so "auto" captures only when the frame is tiled over several GPUs.
    want_graph = args.graph == "on" or (args.graph == "auto" and world > 1)
    if want_graph and not (synthetic and args.schedule == "default"):     # the schedule pick reads events: eager only
        try:
            render_frames(main_stream)
            torch.cuda.synchronize()
            captured = []
            for buf in range(nbuf):
                g = torch.cuda.CUDAGraph()
                # thread_local: RCCL's watchdog thread may query events while this thread captures
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    lead = torch.cuda.current_stream()
                    handles[0].set_stream(lead.cuda_stream)
                    render_frames(lead, buf)
                handles[0].set_stream(main_stream.cuda_stream)
                g.replay()
                torch.cuda.synchronize()
                captured.append(g)
            graphs = captured
        except Exception as exc:                     # capture not possible here: keep launching eagerly
            handles[0].set_stream(main_stream.cuda_stream)
            if args.graph == "on":
                raise
            if rank == 0:
                print("bench.py: HIP graph capture unavailable (%s); launching eagerly" % exc, file=sys.stderr)

    state = {"next": 0, "pending": None}

    def finish(pending):
        """Second half of a step: wait for its gather, restore the frames' row order on rank 0."""
        work, buf, host_gathered = pending
        work.wait()                                     # RCCL: the current stream waits, the host does not
        if rank == 0:
            if host_gathered is not None:
                gathered_sets[buf].copy_(host_gathered)
            ds.deinterleave_frames(gathered_sets[buf].data_ptr(), frames.data_ptr(), W, H, ROW_BLOCK, world, 3, B,
                                   rank_stride_bytes=B * tile_bytes, tile_stride_bytes=tile_bytes)   # one launch

    def step():
        buf = state["next"]
        state["next"] = (buf + 1) % nbuf
        if graphs is not None:
            graphs[buf].replay()
        else:
            render_frames(main_stream, buf)
        if world > 1:
            if rehearsal:
                torch.cuda.synchronize()
                src = tile_sets[buf].cpu()
                dst = torch.zeros((world,) + tuple(src.shape), dtype=torch.uint8) if rank == 0 else None
            else:
                src, dst = tile_sets[buf], (gathered_sets[buf] if rank == 0 else None)
            work = MG.gather_to_root(src, dist, rank, world, dst, async_op=True)
            this = (work, buf, dst if rehearsal else None)
            if nbuf == 1:
                finish(this)
            else:                                       # the previous step's gather had a whole render to complete
                if state["pending"] is not None:
                    finish(state["pending"])
                state["pending"] = this

    def barrier():
        if state["pending"] is not None:
            finish(state["pending"])
            state["pending"] = None
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0

    # roofline pass: HIP events on the launch stream around each frame and around its dominant
    # kernel (wf_primary_kernel), averaged over steps*B frames
    nl = max(args.steps, 1) * B
    frame_ms_sum = kern_ms_sum = 0.0
    for _ in range(nl):
        ds.render_device(cam, rgb8_ptr=tiles[0].data_ptr(), max_depth=MAX_DEPTH, accel=P.ACCEL_BVH,
                         rank=rank, world=world, row_block=ROW_BLOCK, **sched, profile=True)
        f_ms, k_ms = ds.profile()
        frame_ms_sum += f_ms
        kern_ms_sum += k_ms
    kern_ms = kern_ms_sum / nl
    frame_dev_ms = frame_ms_sum / nl
    # level-1 work of this rank's rows = a depth-1 frame (primary closest hits + their shadow queries)
    ds.render_device(cam, rgb8_ptr=tiles[0].data_ptr(), max_depth=1, accel=P.ACCEL_BVH,
                     rank=rank, world=world, row_block=ROW_BLOCK, **sched, counters=True)
    c1 = ds.counters()
    alg_bytes_l1 = c1["algorithmic_bytes"] + 3 * c1["pixels"]

    red_dev = torch.device("cpu") if rehearsal else dev
    tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    rr = torch.tensor([float(my_rays)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
    dt_max = float(tt.item())
    rays_frame = float(rr.item())

    if rank == 0:
        # the frame that was timed (all ranks' tiles stitched on rank 0 when N > 1)
        final = (frames[B - 1] if world > 1 else tiles[B - 1][:H]).cpu().numpy()
        total_rays = rays_frame * B * args.steps
        achieved = alg_bytes_l1 / (kern_ms * 1e-3) / 1e9
        traffic = None
        stats = ds.stats()
        dominant = "whitted_tree_kernel" if ds.last_schedule() == "tree" else "wf_primary_kernel"
        pmc_file = args.pmc_json or os.path.join(REPO, "profiles", "r01_synthetic_%d_pmc.json" % args.prims if synthetic
                                                 else "r01_final_pmc.json")
        try:
            prof = json.load(open(pmc_file))
            for name, e in prof["kernels"].items():       # the timed build: first template argument (COUNT) false
                if dominant in name and "<true" not in name and "hbm_bytes_per_launch_corrected" in e:
                    traffic = e["hbm_bytes_per_launch_corrected"]
        except Exception:
            pass
        line = {
            "metric": "Mrays/s + ms/frame @1920x1080 depth4",
            "value": total_rays / dt_max / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "ms_per_frame": dt_max / args.steps / B * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": ("synthetic: %d random spheres+triangles (SURVEY 8d scaling scene, seed 2024), mount_low camera" % args.prims)
                    if synthetic else
                    "synthetic: P3D_Scenes/mount_low.p3f (12 primitives, 1 light) resolution/accel overridden to the config",
            "config": {"workload": ("SURVEY 8d scaling scene, %d primitives, 1920x1080 depth 4 BVH" % args.prims) if synthetic
                       else "mount_low.p3f 1920x1080 depth 4 BVH (BASELINE config 2)",
                       "frames_per_step": B, "frames_in_flight": F, "hip_graph": graphs is not None,
                       "gather": ("pipelined" if nbuf == 2 else "after each step") if world > 1 else None,
                       "rays_per_frame": int(rays_frame),
                       "row_block": ROW_BLOCK,
                       "parallelism": "1 GPU" if world == 1 else "%d GPUs: interleaved 16-row blocks + RCCL gather to rank 0" % world,
                       "frame_checksum": int(final.astype(np.uint64).sum())},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": dominant, "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": int(alg_bytes_l1),
                         "frame_device_ms": frame_dev_ms,
                         "frame_algorithmic_bytes": int(alg_bytes),
                         "frame_algorithmic_GBps": alg_bytes / (frame_dev_ms * 1e-3) / 1e9,
                         "note": ("algorithmic bytes = 32 B/slab test + 16/48/32 B per sphere/triangle/box test + 3 B/px "
                                  "(SURVEY 8d); scene read from HBM/L2 (%d MB on the device); `traffic` = PMC FETCH_SIZE x2 + "
                                  "WRITE_SIZE per launch from %s" % (stats["device_bytes"] >> 20, os.path.basename(pmc_file)))
                                 if synthetic else
                                 "algorithmic bytes = 32 B/slab test + 16/48/32 B per sphere/triangle/box test + 3 B/px "
                                 "(SURVEY 8d). The 12-primitive scene is LDS-resident: `traffic` (PMC FETCH_SIZE x2 + "
                                 "WRITE_SIZE per launch, profiles/r01_final_pmc.json, 1-GPU whole-frame launch) is "
                                 "frame buffer + ray/node queues, far below the algorithmic figure (see DESIGN.md)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_synthetic(args.prims) if synthetic else cpu_baseline(scene_file)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for h in handles:
        h.close()


if __name__ == "__main__":
    main()
