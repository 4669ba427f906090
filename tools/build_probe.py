#!/usr/bin/env python3
"""Host SAH vs device LBVH: build time, tree statistics and frame time (GPU box).
usage: build_probe.py SCENE_OR_N [SCENE_OR_N ...]"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import scene_path
import torch
import u_4a_2s_p3d_raytracer_template2_amd as P
from u_4a_2s_p3d_raytracer_template2_amd import synthetic as S, api

res = (1920, 1080)
for arg in sys.argv[1:]:
    if arg.isdigit():
        cam = P.HostScene(S.camera_p3f("/tmp/synth_camera.p3f", *res)).camera()
        desc, keep = api.make_desc(*S.arrays(int(arg)))
        make = lambda b: P.DeviceScene(desc, keepalive=keep, builder=b & 1, cull_never_hit=bool(b & 2))
    else:
        hs = P.HostScene(scene_path(arg)); hs.set_resolution(*res); cam = hs.camera()
        make = lambda b: P.DeviceScene.from_host(hs, builder=b & 1, cull_never_hit=bool(b & 2))
    make(1).close()                                   # first-use costs (module load) out of the timings
    frames = []
    for b in (0, 1, 2):
        t0 = time.time(); ds = make(b); dt = time.time() - t0
        st = ds.stats()
        buf = torch.zeros((res[1] + 16, res[0], 3), dtype=torch.uint8, device="cuda")
        line = "%-9s %s: scene_create %.3f s  nodes %d depth %d sah %.1f |" % (arg, ("host SAH", "device LBVH", "host SAH + cull_never_hit")[b], dt, st["n_nodes"], st["max_depth"], st["sah_cost"])
        for name, kw in (("wavefront", dict(wavefront=True)), ("tree", dict(tree=True))):
            for _ in range(3): ds.render_device(cam, rgb8_ptr=buf.data_ptr(), max_depth=4, **kw)
            ds.timer_begin()
            for _ in range(8): ds.render_device(cam, rgb8_ptr=buf.data_ptr(), max_depth=4, **kw)
            line += " %s %.3f ms" % (name, ds.timer_end() / 8)
        frames.append(buf.cpu().numpy().copy())
        print(line)
        ds.close()
    print("   same frame:", all(np.array_equal(frames[0], f) for f in frames[1:]))
