#!/usr/bin/env python3
"""Per-TILE durations of the tile schedule (one launch, persistent workgroups drawing 16x16 tiles) from stamps, and what
the same tiles would take if they were drawn heaviest first (longest-processing-time order) instead of row by row.
usage: python tools/tile_timeline.py SCENE [depth] [spp]      SCENE: mount_low | dragon | synthetic:N"""
import heapq
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

scene = sys.argv[1]
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 4
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
buf = torch.zeros((H + 16, W, 3), dtype=torch.uint8, device="cuda")
ntiles = 120 * 68 * 4
st = torch.zeros((ntiles + 64, 8), dtype=torch.int64, device="cuda")
for _ in range(3):
    ds.render_device(cam, rgb8_ptr=buf.data_ptr(), max_depth=depth, tile=True)
ds.sync()
ds.debug_set_stamps(st.data_ptr())
ds.render_device(cam, rgb8_ptr=buf.data_ptr(), max_depth=depth, tile=True, profile=True)
f_ms, k_ms = ds.profile()
ds.debug_set_stamps(0)
s = st.cpu().numpy()[:ntiles]
ok = (s[:, 0] != 0) & (s[:, 1] != 0)
s = s[ok]
t0 = s[:, 0].min()
start, end = (s[:, 0] - t0) * 0.01, (s[:, 1] - t0) * 0.01
dur = end - start
wgs = len(np.unique(s[:, 2]))
print("%s tile schedule depth %d: frame %.4f ms; %d tiles on %d workgroups; span %.1f us" % (scene, depth, f_ms, len(s), wgs, end.max()))
print("tile duration: mean %.2f us  p50 %.2f  p90 %.2f  p99 %.2f  max %.2f;  sum %.0f us = %.1f us per workgroup if perfectly balanced"
      % (dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max(), dur.sum(), dur.sum() / wgs))
busy = np.zeros(int(s[:, 2].max()) + 1)
np.add.at(busy, s[:, 2].astype(int), dur)
print("workgroup busy time: min %.1f  mean %.1f  max %.1f us; last tile starts at %.1f us" % (busy[busy > 0].min(), busy[busy > 0].mean(), busy.max(), start.max()))


def simulate(order):
    free = [0.0] * wgs
    heapq.heapify(free)
    last = 0.0
    for d in order:
        t = heapq.heappop(free) + d
        last = max(last, t)
        heapq.heappush(free, t)
    return last


print("list-scheduling the measured durations on %d workgroups: row order %.1f us, heaviest first %.1f us, lightest first %.1f us"
      % (wgs, simulate(dur), simulate(np.sort(dur)[::-1]), simulate(np.sort(dur))))
