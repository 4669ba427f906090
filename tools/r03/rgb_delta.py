#!/usr/bin/env python3
"""Per golden case: how far the device frame is from the oracle's in rgb32f (max |delta|, number of floats whose bits differ)."""
import json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import GOLDEN, scene_path
import u_4a_2s_p3d_raytracer_template2_amd as P

cases = json.load(open(os.path.join(GOLDEN, "cases.json")))
frames = np.load(os.path.join(GOLDEN, "frames.npz"))
tot = 0
for name in sorted(cases):
    m = dict(cases[name], name=name)
    hs = P.HostScene(scene_path(m["scene"])); hs.set_resolution(*m["res"])
    ds = P.DeviceScene.from_host(hs)
    samples = hs.samples(m["seed"], m["spp"]) if m["spp"] else None
    out = ds.render(hs.camera(), max_depth=m["max_depth"], accel=m["accel"], spp=m["spp"], samples=samples)
    ds.close()
    ref = frames[m["name"] + "/rgb32f"] if (m["name"] + "/rgb32f") in frames.files else None
    if ref is None:
        print(m["name"], "no rgb32f fixture"); continue
    a, b = out["rgb32f"], ref
    fin = np.isfinite(a) & np.isfinite(b)
    nd = int(((a.view(np.uint32) != b.view(np.uint32)) & fin).sum())
    tot += nd
    print("%-40s max|d| %.3g  differing floats %d of %d" % (m["name"], float(np.abs(a[fin].astype(np.float64) - b[fin]).max()), nd, a.size))
print("TOTAL differing floats", tot)
