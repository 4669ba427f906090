#!/usr/bin/env python3
"""One rank's share of a config-2 frame (interleaved 16-row blocks) on every schedule: ms per shard-frame (GPU box tool)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import scene_path
import torch
import u_4a_2s_p3d_raytracer_template2_amd as P
hs = P.HostScene(scene_path("mount_low")); hs.set_resolution(1920, 1080)
cam = hs.camera()
FMAX = 12
dss = [P.DeviceScene.from_host(hs) for _ in range(FMAX)]
streams = [torch.cuda.Stream() for _ in range(FMAX)]
for ds, st in zip(dss, streams):
    ds.set_stream(st.cuda_stream)
scheds = sys.argv[1].split(",") if len(sys.argv) > 1 else ["wavefront", "tile", "tree"]
for world in (1, 2, 4, 8):
  rows = P.local_rows(1080, 16, world)
  bufs = [torch.zeros((rows + 16, 1920, 3), dtype=torch.uint8, device="cuda") for _ in range(FMAX)]
  for F in (3, 4, 6, 12):
    for sched in scheds:
        kw = dict(max_depth=4, accel=2, rank=0, world=world, **{sched: True})
        for ds, b in zip(dss, bufs):
            for _ in range(3):
                ds.render_device(cam, rgb8_ptr=b.data_ptr(), **kw)
        torch.cuda.synchronize()
        n = 240
        dss[0].timer_begin()
        for _ in range(n):
            dss[0].render_device(cam, rgb8_ptr=bufs[0].data_ptr(), **kw)
        single = dss[0].timer_end() / n
        # F in flight on separate handles / streams
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for st in streams[:F]: st.wait_event(e0)
        for i in range(n):
            dss[i % F].render_device(cam, rgb8_ptr=bufs[i % F].data_ptr(), **kw)
        for st in streams[:F]: torch.cuda.current_stream().wait_stream(st)
        e1.record(); torch.cuda.synchronize()
        eager = e0.elapsed_time(e1) / n
        nb = 12
        # the same n frames as ONE captured graph (what bench.py replays when the frame is tiled over several GPUs)
        lead = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(lead):
            with torch.cuda.graph(g, stream=lead, capture_error_mode="thread_local"):
                for st in streams[:F]: st.wait_stream(lead)
                for i in range(nb):
                    dss[i % F].render_device(cam, rgb8_ptr=bufs[i % F].data_ptr(), **kw)
                for st in streams[:F]: lead.wait_stream(st)
        torch.cuda.synchronize()
        with torch.cuda.stream(lead):
            g.replay(); torch.cuda.synchronize()
            e0.record(lead)
            for _ in range(n // nb): g.replay()
            e1.record(lead)
        torch.cuda.synchronize()
        print("world %d (%4d rows) %-9s single %.4f ms   %2d in flight %.4f ms/frame   as a 12-frame graph %.4f ms/frame" % (world, rows, sched, single, F, eager, e0.elapsed_time(e1) / (n // nb * nb)), flush=True)
        del g
