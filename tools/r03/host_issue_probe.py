#!/usr/bin/env python3
"""Is config 2's timed region bound by the host's launch rate?  Issues the bench's frames (4 handles, 4 streams, no joins)
and reports the time the host needed to ENQUEUE them next to the time until the GPU finished."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
from conftest import scene_path
import u_4a_2s_p3d_raytracer_template2_amd as P

W, H = 1920, 1080
hs = P.HostScene(scene_path("mount_low")); hs.set_resolution(W, H)
cam = hs.camera()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 4
handles = [P.DeviceScene.from_host(hs) for _ in range(F)]
streams = [torch.cuda.Stream() for _ in handles]
for h, st in zip(handles, streams):
    h.set_stream(st.cuda_stream)
bufs = [torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(12)]
kw = dict(max_depth=4, accel=2)
for _ in range(3):
    for f in range(12):
        handles[f % F].render_device(cam, rgb8_ptr=bufs[f].data_ptr(), **kw)
torch.cuda.synchronize()
for rep in range(3):
    n = 0
    t0 = time.perf_counter()
    for _ in range(40):
        for f in range(12):
            handles[f % F].render_device(cam, rgb8_ptr=bufs[f].data_ptr(), **kw)
            n += 1
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%d frames on %d handles: host enqueue %.4f ms/frame, until the GPU is done %.4f ms/frame (GPU behind the host by %.3f ms at the end)" % (
        n, F, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, (t2 - t1) * 1e3), flush=True)
