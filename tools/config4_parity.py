#!/usr/bin/env python3
"""BASELINE config 4 at FULL size against the oracle (one-off check, ~1 min of CPU): mount_low 4096x4096,
depth 6, spp 2 (4 jittered thin-lens samples, summed and divided by 16), seed 12345."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import scene_path
from oracle import oracle_py as O
import u_4a_2s_p3d_raytracer_template2_amd as P
res = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sc = O.Scene(scene_path("mount_low")); sc.set_resolution(res, res)
t0 = time.time()
ref = sc.render(max_depth=6, accel=2, spp=2, seed=12345, threads=1)
print("oracle: %.1f s, %d rays" % (time.time() - t0, ref["counters"]["rays"]), flush=True)
hs = P.HostScene(scene_path("mount_low")); hs.set_resolution(res, res)
ds = P.DeviceScene.from_host(hs)
out = ds.render(hs.camera(), max_depth=6, accel=2, spp=2, samples=hs.samples(12345, 2), counters=True)
d = np.abs(out["rgb32f"] - ref["rgb32f"])
print("hit ids identical:", np.array_equal(out["hit_id"], ref["hit_id"]),
      "| rays", out["counters"]["rays"], "vs", ref["counters"]["rays"],
      "| max |dRGB| %.3g" % d.max(), "| rgb8 mismatches", int((out["rgb8"] != ref["rgb8"]).sum()), "of", out["rgb8"].size)
