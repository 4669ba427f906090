#!/usr/bin/env python3
"""Per-wave timeline of one launch of a frame from s_memrealtime stamps (p3d_debug_set_stamps /
p3d_debug_set_stamp_level; GPU box tool).

usage: python tools/wave_timeline.py SCENE SCHEDULE LEVEL [depth]
  SCENE     mount_low | dragon | synthetic:N
  SCHEDULE  wavefront | tree
  LEVEL     1 = the level-1 launch (wavefront) or the tree launch; l >= 2 = the wavefront schedule's level-l launch
Prints the launch's span, the distribution of wave lifetimes, waves in flight over the span (20 slices), how much
of the span is the ramp and the tail, and for level launches the split of a wave's first batch into
queue read / closest hit / shading (incl. shadow queries) / queue append.
"""
import os
import sys

import numpy as np

# stamps exist only in the diagnostic build (make -C u_4a_2s_p3d_raytracer_template2_amd/csrc stamps)
_stamps = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "u_4a_2s_p3d_raytracer_template2_amd", "libp3d_hip_stamps.so")
if "P3D_LIB" not in os.environ:
    if not os.path.exists(_stamps):
        raise SystemExit("the timeline needs %s: run `make -C u_4a_2s_p3d_raytracer_template2_amd/csrc stamps`" % _stamps)
    os.environ["P3D_LIB"] = _stamps

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import scene_path  # noqa: E402
import torch  # noqa: E402
import u_4a_2s_p3d_raytracer_template2_amd as P  # noqa: E402

scene, sched, level = sys.argv[1], sys.argv[2], int(sys.argv[3])
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 4
W, H = 1920, 1080
if scene.startswith("synthetic:"):
    from u_4a_2s_p3d_raytracer_template2_amd import synthetic as S, api
    cam = P.HostScene(S.camera_p3f("/tmp/synth_camera.p3f", W, H)).camera()
    desc, keep = api.make_desc(*S.arrays(int(scene.split(":")[1])))
    ds = P.DeviceScene(desc, keepalive=keep)
else:
    hs = P.HostScene(scene_path(scene))
    hs.set_resolution(W, H)
    cam = hs.camera()
    ds = P.DeviceScene.from_host(hs)
kw = {sched: True}
buf = torch.zeros((H + 16, W, 3), dtype=torch.uint8, device="cuda")
nrec = max(120 * 272 * 4 + 64, 65536)
st = torch.zeros((nrec, 8), dtype=torch.int64, device="cuda")
for _ in range(3):
    ds.render_device(cam, rgb8_ptr=buf.data_ptr(), max_depth=depth, **kw)
ds.sync()
ds.debug_set_stamp_level(level)
ds.debug_set_stamps(st.data_ptr())
ds.render_device(cam, rgb8_ptr=buf.data_ptr(), max_depth=depth, profile=True, **kw)
f_ms, k_ms = ds.profile()
ds.debug_set_stamps(0)
print("%s %s level %d depth %d: frame %.4f ms, dominant kernel %.4f ms (with stamps on)" % (scene, sched, level, depth, f_ms, k_ms))
s = st.cpu().numpy()
s = s[s[:, 0] != 0]
end_slot = 4 if level == 1 else 5
s = s[s[:, end_slot] != 0]
t0 = s[:, 0].min()
start = (s[:, 0] - t0) * 0.01              # us (100 MHz counter)
end = (s[:, end_slot] - t0) * 0.01
span = end.max()
life = end - start
print("waves stamped: %d   span (first start .. last end): %.1f us" % (len(s), span))
print("wave lifetime: mean %.2f us  p10 %.2f  p50 %.2f  p90 %.2f  p99 %.2f  max %.2f;  sum %.0f us = %.1f waves in flight on average"
      % (life.mean(), *np.percentile(life, [10, 50, 90, 99]), life.max(), life.sum(), life.sum() / span))
if level >= 2:
    nb = s[:, 6]
    print("batches per wave: mean %.2f  max %d   (waves with work: %d)" % (nb.mean(), nb.max(), len(s)))
    T = (s[:, :5] - s[:, :1]) * 0.01
    names = ["queue read", "closest hit", "shade+shadow", "emit/combine"]
    if (s[:, 1] == 0).all():               # HBM path has no separate queue-read stamp
        T[:, 1] = 0.0
    d = np.diff(T, axis=1)
    for k, n in enumerate(names):
        print("  first batch %-13s mean %6.2f us  p50 %6.2f  p90 %6.2f  p99 %6.2f" % (n, d[:, k].mean(), *np.percentile(d[:, k], [50, 90, 99])))
    print("  first batch total         mean %6.2f us; later batches of the wave: %.2f us" % (T[:, 4].mean(), (life - T[:, 4]).mean()))
# waves in flight over the span
edges = np.linspace(0.0, span, 21)
mid = 0.5 * (edges[1:] + edges[:-1])
conc = [(int(((start <= m) & (end > m)).sum())) for m in mid]
print("waves in flight at the middle of 20 slices of the span:", conc)
peak = max(conc)
ramp = next(m for m, c in zip(mid, conc) if c >= 0.9 * peak)
tail = span - next(m for m, c in zip(mid[::-1], conc[::-1]) if c >= 0.5 * peak)
print("peak %d waves in flight; reaches 90 %% of it at %.1f us; below half of it for the last %.1f us (%.0f %% of the span)"
      % (peak, ramp, tail, 100.0 * tail / span))
print("last-starting wave starts at %.1f us (%.0f %% of the span)" % (start.max(), 100.0 * start.max() / span))
xcc = (s[:, 7] >> 32) & 0xF
print("waves per XCC:", np.bincount(xcc.astype(int), minlength=8).tolist())
