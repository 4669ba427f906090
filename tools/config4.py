#!/usr/bin/env python3
"""BASELINE config 4 on one GPU: mount_low 4096x4096, depth 6, spp 2 (4 jittered samples + thin lens,
summed and divided by 16 like the reference), host-generated libc rand() sample stream (seed 12345)."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import scene_path
import torch
import u_4a_2s_p3d_raytracer_template2_amd as P
res = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
hs = P.HostScene(scene_path("mount_low")); hs.set_resolution(res, res); cam = hs.camera()
t0 = time.time(); samples = hs.samples(12345, 2); t1 = time.time()
print("sample stream: %.2f s on the host, %.0f MB" % (t1 - t0, samples.nbytes / 1e6))
ds = P.DeviceScene.from_host(hs)
if len(sys.argv) > 2 and int(sys.argv[2]) > 0:
    ds.set_tuning(workspace_mib=int(sys.argv[2]))          # workspace budget in MiB (default 65536)
if len(sys.argv) > 3:
    ds.set_tuning(waves_per_simd=int(sys.argv[3]))         # register budget of the ray kernels: 0 default, 5, 6
out = torch.zeros((res, res, 3), dtype=torch.uint8, device="cuda")
kw = dict(max_depth=6, accel=P.ACCEL_BVH, spp=2, samples=samples)
ds.render_device(cam, rgb8_ptr=out.data_ptr(), counters=True, **kw)
c = ds.counters(); print("counters", c)
dev_samples = torch.from_numpy(samples).cuda()            # uploaded once (P3D_FLAG_DEVICE_SAMPLES)
kwd = dict(max_depth=6, accel=P.ACCEL_BVH, spp=2, samples_ptr=dev_samples.data_ptr())
for sched in ("tile", "wavefront", "tree"):
    for _ in range(1): ds.render_device(cam, rgb8_ptr=out.data_ptr(), **{sched: True}, **kwd)
    ds.sync()
    n = 3
    t = time.perf_counter()
    for _ in range(n): ds.render_device(cam, rgb8_ptr=out.data_ptr(), profile=True, **{sched: True}, **kwd)
    f_ms, k_ms = ds.profile(); ds.sync()
    wall = (time.perf_counter() - t) / n * 1e3
    print("%s: device %.2f ms/frame (wall %.1f ms, samples resident)  %.1f Mrays/s  checksum %d" % (
        sched, f_ms, wall, c["rays"] / f_ms / 1e3, int(out.sum().item())))
