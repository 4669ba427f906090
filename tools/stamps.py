#!/usr/bin/env python3
"""Per-wave timeline of the fused level-1 kernel (diagnostic)."""
import os, sys
import numpy as np
# stamps exist only in the diagnostic build (make -C u_4a_2s_p3d_raytracer_template2_amd/csrc stamps)
_stamps = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "u_4a_2s_p3d_raytracer_template2_amd", "libp3d_hip_stamps.so")
if "P3D_LIB" not in os.environ:
    if not os.path.exists(_stamps):
        raise SystemExit("the timeline needs %s: run `make -C u_4a_2s_p3d_raytracer_template2_amd/csrc stamps`" % _stamps)
    os.environ["P3D_LIB"] = _stamps
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import scene_path
import torch
import u_4a_2s_p3d_raytracer_template2_amd as P
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 4
factor = 0
syn = int(sys.argv[2]) if len(sys.argv) > 2 else 0
kw = {}
if syn:
    from u_4a_2s_p3d_raytracer_template2_amd import synthetic as S, api
    cam = P.HostScene(S.camera_p3f("/tmp/synth_camera.p3f", 1920, 1080)).camera()
    desc, keep = api.make_desc(*S.arrays(syn))
    ds = P.DeviceScene(desc, keepalive=keep)
    kw = dict(wavefront=True)
else:
    hs = P.HostScene(scene_path("mount_low")); hs.set_resolution(1920, 1080); cam = hs.camera()
    ds = P.DeviceScene.from_host(hs)
buf = torch.zeros((1088, 1920, 3), dtype=torch.uint8, device="cuda")
ntile = 120 * 68 + 64
st = torch.zeros((ntile * 4, 8), dtype=torch.int64, device="cuda")
for _ in range(3): ds.render_device(cam, rgb8_ptr=buf.data_ptr(), max_depth=depth, **kw)
ds.sync()
ds.debug_set_stamps(st.data_ptr())
ds.render_device(cam, rgb8_ptr=buf.data_ptr(), max_depth=depth, profile=True, **kw)
print("profile (frame_ms, kernel_ms):", ds.profile())
ds.debug_set_stamps(0)
s = st.cpu().numpy()
s = s[s[:, 0] != 0]
t0 = s[:, 0].min()
T = (s[:, :5] - t0) * 0.01   # us
print("waves stamped:", len(s), " kernel span (first start .. last end): %.1f us" % T[:, 4].max())
d = np.diff(T, axis=1)
names = ["raygen", "closest", "shade", "emit"]
for k, n in enumerate(names):
    print("%8s: mean %.2f us  p50 %.2f  p90 %.2f  p99 %.2f  max %.2f" % (n, d[:, k].mean(), *np.percentile(d[:, k], [50, 90, 99]), d[:, k].max()))
tot = T[:, 4] - T[:, 0]
print("   total: mean %.2f us  p50 %.2f  p90 %.2f  p99 %.2f  max %.2f   sum %.0f us (=%.1f wave-slots busy on average)" % (
    tot.mean(), *np.percentile(tot, [50, 90, 99]), tot.max(), tot.sum(), tot.sum() / T[:, 4].max()))
# concurrency over time
ev = np.concatenate([np.stack([T[:, 0], np.ones(len(T))], 1), np.stack([T[:, 4], -np.ones(len(T))], 1)])
ev = ev[np.argsort(ev[:, 0])]
conc = np.cumsum(ev[:, 1])
for q in (0.05, 0.25, 0.5, 0.75, 0.95):
    i = int(q * len(ev)); print("  t=%.1f us: %d waves in flight" % (ev[i, 0], conc[i]))
hw = s[:, 7]
xcc = (hw >> 32) & 0xF
print("waves per XCC:", np.bincount(xcc.astype(int), minlength=8))
