#!/usr/bin/env python3
"""Kernel-only timing probe for tuning runs on the GPU box (not part of the product).
usage: python tools/perf_probe.py [scene] [W H] [--chunks 1,4,16] [--depth 4] [--accel 2] [--n 50]"""
import argparse
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import scene_path  # noqa: E402
import u_4a_2s_p3d_raytracer_template2_amd as P  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("scene", nargs="?", default="mount_low")
ap.add_argument("res", nargs="*", type=int, default=[1920, 1080])
ap.add_argument("--chunks", default="1")
ap.add_argument("--depth", type=int, default=4)
ap.add_argument("--accel", type=int, default=2)
ap.add_argument("--leaf", type=int, default=0)
ap.add_argument("--n", type=int, default=50)
ap.add_argument("--tree", action="store_true")
ap.add_argument("--occ", default="0")
ap.add_argument("--eye", nargs=3, type=float, default=None)
ap.add_argument("--f32", action="store_true")
ap.add_argument("--synthetic", type=int, default=0, help="N primitives of the SURVEY 8d scaling scene instead of a .p3f")
a = ap.parse_args()

import torch  # noqa: E402
import time  # noqa: E402
if a.synthetic:
    from u_4a_2s_p3d_raytracer_template2_amd import synthetic as S, api  # noqa: E402
    hs = P.HostScene(S.camera_p3f("/tmp/synth_camera.p3f", *a.res))
    cam = hs.camera()
    desc, keep = api.make_desc(*S.arrays(a.synthetic))
    t0 = time.time()
    ds = P.DeviceScene(desc, leaf_max=a.leaf, keepalive=keep)
    print("scene_create %.2f s" % (time.time() - t0))
else:
    hs = P.HostScene(scene_path(a.scene))
    hs.set_resolution(*a.res)
    if a.eye:
        hs.set_eye(*a.eye)
    cam = hs.camera()
    ds = P.DeviceScene.from_host(hs, leaf_max=a.leaf)
print("stats", ds.stats())
buf = torch.zeros((a.res[1] + 16, a.res[0], 3), dtype=torch.uint8, device="cuda")
ds.render_device(cam, rgb8_ptr=buf.data_ptr(), max_depth=a.depth, accel=a.accel, counters=True)
c = ds.counters()
print("counters", c)
ref = None
for sched in (["tile", "wavefront", "tree"] if not a.tree else ["tree"]):
  tree = sched == "tree"
  for no_lds, no_packet in ((False, False), (False, True), (True, False), (True, True)):
   if tree and not no_packet: continue
   for occ in [int(v) for v in a.occ.split(",")]:
    for ch in [int(v) for v in a.chunks.split(",")]:
        ds.set_tuning(xcd_chunk=ch, waves_per_simd=occ)
        kw = dict(max_depth=a.depth, accel=a.accel, no_lds=no_lds, packet=not no_packet, **{sched: True})
        for _ in range(5):
            ds.render_device(cam, rgb8_ptr=buf.data_ptr(), **kw)
        ds.timer_begin()
        for _ in range(a.n):
            ds.render_device(cam, rgb8_ptr=buf.data_ptr(), **kw)
        ms = ds.timer_end() / a.n
        img = buf.cpu().numpy()
        if ref is None:
            ref = img
        print("%s %s occ %d xcd_chunk %6d: %.4f ms/frame  %.1f Mrays/s  alg %.0f GB/s  same_image=%s" % (
            "%-9s" % sched, ("hbm" if no_lds else "lds") + ("/lane  " if no_packet else "/packet"), occ, ch, ms, c["rays"] / ms / 1e3,
            (c["algorithmic_bytes"] + 3 * c["pixels"]) / ms / 1e6, np.array_equal(img, ref)))
