#!/usr/bin/env python3
"""Host-side enqueue cost of p3d_render vs device time (tuning probe, not part of the product)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import scene_path
import torch
import u_4a_2s_p3d_raytracer_template2_amd as P
hs = P.HostScene(scene_path("mount_low")); hs.set_resolution(1920, 1080); cam = hs.camera()
ds = P.DeviceScene.from_host(hs)
buf = torch.zeros((1088, 1920, 3), dtype=torch.uint8, device="cuda")
for tree in (False, True):
    for _ in range(10): ds.render_device(cam, rgb8_ptr=buf.data_ptr(), tree=tree)
    ds.sync()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n): ds.render_device(cam, rgb8_ptr=buf.data_ptr(), tree=tree)
    t1 = time.perf_counter()
    ds.sync()
    t2 = time.perf_counter()
    print("tree" if tree else "wavefront", "host enqueue %.1f us/frame, total %.1f us/frame" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
