#!/usr/bin/env python3
"""Render a few frames of one scene (name of a golden .p3f, or N = primitives of the synthetic scaling
scene) with one schedule: wavefront | tree | tile | default (| X_private: forced with private walks).  For rocprofv3 runs.
usage: render_frames.py SCENE SCHEDULE [FRAMES [W H [DEPTH]]]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import scene_path
import torch
import u_4a_2s_p3d_raytracer_template2_amd as P
from u_4a_2s_p3d_raytracer_template2_amd import synthetic as S, api
arg, sched = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
res = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (1920, 1080)
depth = int(sys.argv[6]) if len(sys.argv) > 6 else 4
if arg.isdigit():
    cam = P.HostScene(S.camera_p3f("/tmp/synth_camera.p3f", *res)).camera()
    desc, keep = api.make_desc(*S.arrays(int(arg)))
    ds = P.DeviceScene(desc, keepalive=keep)
else:
    hs = P.HostScene(scene_path(arg)); hs.set_resolution(*res)
    ds, cam = P.DeviceScene.from_host(hs), hs.camera()
buf = torch.zeros((res[1] + 16, res[0], 3), dtype=torch.uint8, device="cuda")
kw = {"wavefront": dict(wavefront=True), "tree": dict(tree=True), "tile": dict(tile=True), "default": {},
      "wavefront_packet": dict(wavefront=True, packet=True),
      "wavefront_private": dict(wavefront=True, private_walk=True), "tree_private": dict(tree=True, private_walk=True),
      "tile_private": dict(tile=True, private_walk=True)}[sched]
for _ in range(n):
    ds.render_device(cam, rgb8_ptr=buf.data_ptr(), max_depth=depth, **kw)
ds.sync()
print(ds.last_schedule())
