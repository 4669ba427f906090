"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C-ABI.

Bars (BASELINE.json north_star): primary hit ids identical to the reference semantics
(nearest hit, lowest scene index on ties); float RGB EQUAL to the oracle's (conftest.RGB_TOL = 0: the device runs
the host libm's powf, csrc/p3d_powf.h); ray counts identical; quantised RGB8 equal.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, RGB_TOL, assert_rgb8_equal, scene_path
from oracle import oracle_py as O
import u_4a_2s_p3d_raytracer_template2_amd as P

pytestmark = pytest.mark.gpu


CASES = json.load(open(os.path.join(GOLDEN, "cases.json")))


@pytest.fixture(scope="module")
def frames():
    return np.load(os.path.join(GOLDEN, "frames.npz"))


def gpu_render(m, counters=False, leaf_max=0, **over):
    hs = P.HostScene(scene_path(m["scene"]))
    hs.set_resolution(*m["res"])
    ds = P.DeviceScene.from_host(hs, leaf_max=leaf_max)
    samples = hs.samples(m["seed"], m["spp"]) if m["spp"] else None
    kw = dict(max_depth=m["max_depth"], accel=m["accel"], spp=m["spp"], samples=samples, counters=counters)
    kw.update(over)
    out = ds.render(hs.camera(), **kw)
    ds.close()
    return out


def compare(out, rgb8, rgb32f, hit_id, name):
    assert np.array_equal(out["hit_id"], hit_id), "%s: primary hit ids differ in %d px" % (
        name, int((out["hit_id"] != hit_id).sum()))
    diff = np.abs(out["rgb32f"].astype(np.float64) - rgb32f.astype(np.float64))
    assert np.isfinite(out["rgb32f"]).all()
    assert diff.max() <= RGB_TOL, "%s: max |rgb diff| = %g" % (name, diff.max())
    n8 = assert_rgb8_equal(out["rgb8"], rgb8, name)
    return diff.max(), n8


def test_device_intersectors_match_oracle_kat():
    k = np.load(os.path.join(GOLDEN, "kat.npz"))
    prim = k["prim12"].copy()
    # C-ABI plane record is (unit normal, D); the fixture holds the loader's three points
    for i in np.where(k["type"] == O.PLANE)[0]:
        p = prim[i, :9].reshape(3, 3).astype(np.float32)
        n = np.cross(p[1] - p[0], p[2] - p[0]).astype(np.float32)   # not bit-exact vs host: skip planes below
        prim[i, :] = 0
    nonplane = k["type"] != O.PLANE
    hit, t, nrm = P.debug_intersect(k["type"][nonplane], prim[nonplane], k["origin"][nonplane], k["dir"][nonplane])
    assert np.array_equal(hit, k["hit"][nonplane].astype(bool))
    h = hit
    assert np.array_equal(t[h].view(np.uint32), k["t"][nonplane][h].view(np.uint32))
    assert np.array_equal(nrm[h].view(np.uint32), k["normal"][nonplane][h].view(np.uint32))
    assert h.sum() > 3000


def test_device_intersectors_match_oracle_live_including_planes():
    rng = np.random.default_rng(99)
    n = 20000
    types = rng.integers(0, 4, n).astype(np.uint32)
    prim = np.zeros((n, 12), np.float32)
    o = (rng.standard_normal((n, 3)) * 3).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    hs = P.HostScene(scene_path("balls_medium"))    # gives a real host-made plane record
    t_, d_, *_ = hs.arrays()
    plane_rec = d_[np.where(t_ == 3)[0][0]]
    for i in range(n):
        if types[i] == 0:
            prim[i, :3] = rng.standard_normal(3); prim[i, 3] = abs(rng.standard_normal()) + 0.05
        elif types[i] == 1:
            prim[i, :9] = rng.standard_normal(9) * 2
        elif types[i] == 2:
            lo = rng.standard_normal(3); prim[i, :3] = lo; prim[i, 3:6] = lo + np.abs(rng.standard_normal(3)) + 0.01
        else:
            prim[i] = plane_rec
    hit, t, nrm = P.debug_intersect(types, prim, o, d)
    exp_hit = np.zeros(n, bool); exp_t = np.zeros(n, np.float32)
    for i in range(n):
        if types[i] == 3:
            # oracle plane from three points on z=-0.5 as the scene file has them
            h, tt, _ = O.intersect(3, [12, 12, -0.5, -12, 12, -0.5, -12, -12, -0.5], o[i], d[i])
        else:
            h, tt, _ = O.intersect(int(types[i]), prim[i], o[i], d[i])
        exp_hit[i] = h; exp_t[i] = tt if h else 0
    assert np.array_equal(hit, exp_hit)
    assert np.array_equal(t[hit].view(np.uint32), exp_t[hit].view(np.uint32))


@pytest.mark.parametrize("schedule", ["tile", "wavefront", "tree"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_frame_matches_golden_fixture(frames, name, schedule):
    m = CASES[name]
    out = gpu_render(m, counters=True, **{schedule: True})
    compare(out, frames[name + "/rgb8"], frames[name + "/rgb32f"], frames[name + "/hit_id"], name)
    c = out["counters"]
    assert c["closest_queries"] == m["counters"]["closest_queries"], name
    assert c["shadow_queries"] == m["counters"]["shadow_queries"], name
    assert c["pixels"] == m["res"][0] * m["res"][1]


@pytest.mark.parametrize("leaf_max", [1, 2, 8])
def test_result_is_independent_of_the_bvh_shape(frames, leaf_max):
    for name in ("c2_mount_low_256x144_d4_bvh", "balls_box_128_d4_none", "c3_dragon_96_d4_bvh"):
        out = gpu_render(CASES[name], leaf_max=leaf_max)
        compare(out, frames[name + "/rgb8"], frames[name + "/rgb32f"], frames[name + "/hit_id"], name)


def test_schedules_and_scene_placements_are_bit_identical():
    """tile vs wavefront vs tree schedule, wave-wide vs per-lane BVH walk, scene read from its LDS copy vs
    from HBM/L2: same bits, same ray counts."""
    for name in ("c2_mount_low_256x144_d4_bvh", "c4_mount_low_96_d6_spp2", "balls_box_128_d4_bvh", "dof_64_d4_spp4",
                 "balls_medium_128_d4_none"):
        a = gpu_render(CASES[name], counters=True, wavefront=True)
        for kw in (dict(tree=True), dict(no_lds=True), dict(tree=True, no_lds=True), dict(wavefront=True, packet=True),
                   dict(wavefront=True, packet=True, no_lds=True), dict(tile=True), dict(tile=True, no_lds=True),
                   dict(tile=True, packet=True), dict(tile=True, packet=True, no_lds=True), dict(wavefront=True, no_lds=True),
                   # scenes read from HBM: lanes sharing their walks (default when forced) vs every lane for itself
                   dict(wavefront=True, no_lds=True, private_walk=True), dict(tile=True, no_lds=True, private_walk=True),
                   dict(tree=True, no_lds=True, private_walk=True)):
            b = gpu_render(CASES[name], counters=True, **kw)
            assert np.array_equal(a["rgb8"], b["rgb8"]) and np.array_equal(a["hit_id"], b["hit_id"]), (name, kw)
            assert np.array_equal(a["rgb32f"].view(np.uint32), b["rgb32f"].view(np.uint32)), (name, kw)
            for k in ("closest_queries", "shadow_queries", "pixels"):      # test counts depend on the walk
                assert a["counters"][k] == b["counters"][k], (name, kw, k)


@pytest.mark.parametrize("depth", [3, 7, 10])
def test_tree_schedule_frame_placements(depth):
    """The tree kernel keeps its per-level frames in private memory for scenes read from HBM up to depth 8 (36 or 84
    dwords per lane) and in LDS otherwise: depths on both sides of each limit, against the wavefront schedule."""
    m = dict(CASES["c2_mount_low_256x144_d4_bvh"])
    m["max_depth"] = depth
    a = gpu_render(m, counters=True, wavefront=True)
    for kw in (dict(tree=True), dict(tree=True, no_lds=True), dict(tile=True, no_lds=True)):
        b = gpu_render(m, counters=True, **kw)
        assert np.array_equal(a["rgb8"], b["rgb8"]) and np.array_equal(a["hit_id"], b["hit_id"]), (depth, kw)
        assert np.array_equal(a["rgb32f"].view(np.uint32), b["rgb32f"].view(np.uint32)), (depth, kw)
        assert a["counters"]["rays"] == b["counters"]["rays"], (depth, kw)


def test_wavefront_bands_do_not_change_the_image():
    """A tiny workspace budget forces the wavefront schedule to run the frame in bands."""
    m = CASES["c2_mount_low_256x144_d4_bvh"]
    hs = P.HostScene(scene_path(m["scene"])); hs.set_resolution(*m["res"])
    ds = P.DeviceScene.from_host(hs)
    full = ds.render(hs.camera(), accel=2, wavefront=True)
    ds.set_tuning(workspace_mib=8)          # a row of 16x16 tiles needs 3.2 MB at depth 4: two tile rows per band
    banded = ds.render(hs.camera(), accel=2, wavefront=True)
    assert ds.last_schedule() == "wavefront"
    assert np.array_equal(full["rgb8"], banded["rgb8"]) and np.array_equal(full["hit_id"], banded["hit_id"])
    assert np.array_equal(full["rgb32f"].view(np.uint32), banded["rgb32f"].view(np.uint32))
    ds.close()


def test_grid_mode_full_size_against_live_oracle():
    """mount_low 1920x1080 depth 4 in GRID mode (accel 1): the reference's grid accepts hits per cell, which at this
    size differs from brute force / BVH mode in one pixel; the device walks the same grid, so
    hit ids, ray counts and colours match the oracle's GRID render, including that pixel."""
    m = dict(scene="mount_low", res=[1920, 1080], accel=1, spp=0, max_depth=4, seed=0)
    sc = O.Scene(scene_path("mount_low")); sc.set_resolution(1920, 1080)
    ref = sc.render(max_depth=4, accel=1, threads=8)
    bvh = sc.render(max_depth=4, accel=2, threads=8)
    n_diff = int((ref["rgb8"] != bvh["rgb8"]).any(axis=2).sum())
    assert n_diff >= 1, "GRID and BVH mode are expected to differ at this size"
    for kw in (dict(tile=True), dict(wavefront=True), dict(tree=True)):
        out = gpu_render(m, counters=True, **kw)
        compare(out, ref["rgb8"], ref["rgb32f"], ref["hit_id"], "grid-1080p")
        assert out["counters"]["rays"] == ref["counters"]["rays"]
        where = (ref["rgb8"] != bvh["rgb8"]).any(axis=2)
        assert np.array_equal(out["rgb8"][where], ref["rgb8"][where])
    print("grid mode: %d px differ from BVH mode, all reproduced" % n_diff)


def test_config4_at_1024_against_live_oracle():
    """BASELINE config 4's shape (mount_low, depth 6, 2x2 jittered samples + thin lens, seed 12345) at 1024x1024:
    every pixel against the oracle's serial render (its libc rand() stream is consumed in pixel order, so one
    thread: ~3 s), on the schedule the library picks (tile) and on the wavefront schedule.  The full 4096x4096
    frame is tools/config4_parity.py (30 s of CPU)."""
    m = dict(scene="mount_low", res=[1024, 1024], accel=2, spp=2, max_depth=6, seed=12345)
    sc = O.Scene(scene_path("mount_low")); sc.set_resolution(1024, 1024)
    ref = sc.render(max_depth=6, accel=2, spp=2, seed=12345)
    for kw in (dict(), dict(wavefront=True)):
        out = gpu_render(m, counters=True, **kw)
        mx, frac = compare(out, ref["rgb8"], ref["rgb32f"], ref["hit_id"], "config4-1024")
        assert out["counters"]["rays"] == ref["counters"]["rays"]
    print("config4 @1024: max rgb diff %.3g, %d rgb8 values differ, %d rays" % (mx, frac, ref["counters"]["rays"]))


def test_config4_full_size_4096():
    """BASELINE config 4 AT FULL SIZE: mount_low 4096x4096, depth 6, 2x2 jittered samples + thin lens, seed 12345.
    The ray count is the reference's (195 972 594: the oracle's count of the whole frame, tools/config4_parity.py), the
    tile and the wavefront schedule agree in every float bit, and the first 64 rows -- 262 144 pixels, whose samples
    are the first 4.2 M draws of the serial rand() stream -- equal the live oracle's render of that strip."""
    torch = pytest.importorskip("torch")
    R, D, SPP = 4096, 6, 2
    hs = P.HostScene(scene_path("mount_low")); hs.set_resolution(R, R)
    cam = hs.camera()
    dev_samples = torch.from_numpy(hs.samples(12345, SPP)).cuda()          # 1.07 GB, uploaded once
    ds = P.DeviceScene.from_host(hs)
    kw = dict(max_depth=D, accel=2, spp=SPP, samples_ptr=dev_samples.data_ptr())
    planes = {}
    for sched in ("tile", "wavefront"):
        f32 = torch.zeros((R, R, 3), dtype=torch.float32, device="cuda")
        u8 = torch.zeros((R, R, 3), dtype=torch.uint8, device="cuda")
        hid = torch.zeros((R, R), dtype=torch.int32, device="cuda")
        ds.render_device(cam, rgb8_ptr=u8.data_ptr(), rgb32f_ptr=f32.data_ptr(), hit_ptr=hid.data_ptr(), counters=True,
                         **{sched: True}, **kw)
        c = ds.counters()
        assert ds.last_schedule() == sched
        assert c["rays"] == 195972594 and c["pixels"] == R * R, (sched, c)
        planes[sched] = (u8, f32, hid)
    a, b = planes["tile"], planes["wavefront"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
    assert torch.equal(a[1].view(torch.int32), b[1].view(torch.int32))
    sc = O.Scene(scene_path("mount_low")); sc.set_resolution(R, R)
    ref = sc.render(max_depth=D, accel=2, spp=SPP, seed=12345, y0=0, y1=64)
    strip = dict(rgb8=a[0][:64].cpu().numpy(), rgb32f=a[1][:64].cpu().numpy(), hit_id=a[2][:64].cpu().numpy())
    compare(strip, ref["rgb8"][:64], ref["rgb32f"][:64], ref["hit_id"][:64], "config4-4096-strip")
    ds.close()


def test_config2_full_size_against_live_oracle():
    """BASELINE config 2: mount_low 1920x1080 depth 4 BVH, every pixel against the oracle."""
    m = dict(scene="mount_low", res=[1920, 1080], accel=2, spp=0, max_depth=4, seed=0)
    out = gpu_render(m, counters=True)
    sc = O.Scene(scene_path("mount_low")); sc.set_resolution(1920, 1080)
    ref = sc.render(max_depth=4, accel=2, threads=8)
    mx, frac = compare(out, ref["rgb8"], ref["rgb32f"], ref["hit_id"], "config2")
    assert out["counters"]["rays"] == ref["counters"]["rays"] == 3808269   # the reference's own count
    print("config2: max rgb diff %.3g, %d rgb8 values differ" % (mx, frac))


def test_config3_full_size_against_live_oracle():
    """BASELINE config 3: dragon 1920x1080 depth 4 BVH (oracle run break-fixed + threaded:
    byte-identical to the unmodified path, SURVEY §6, and 170x faster)."""
    m = dict(scene="dragon", res=[1920, 1080], accel=2, spp=0, max_depth=4, seed=0)
    out = gpu_render(m, counters=True)
    sc = O.Scene(scene_path("dragon")); sc.set_resolution(1920, 1080)
    ref = sc.render(max_depth=4, accel=2, threads=16, break_fixed=1)
    compare(out, ref["rgb8"], ref["rgb32f"], ref["hit_id"], "config3")
    assert out["counters"]["rays"] == ref["counters"]["rays"]


@pytest.mark.parametrize("world", [2, 3, 8])
def test_row_block_tiling_stitches_to_the_single_launch_image(world):
    hs = P.HostScene(scene_path("mount_low"))
    hs.set_resolution(200, 120)                      # 8 row blocks, last one ragged
    ds = P.DeviceScene.from_host(hs)
    cam = hs.camera()
    full = ds.render(cam, accel=2)
    rows = P.local_rows(120, 16, world)
    st8 = np.zeros((120, 200, 3), np.uint8); stf = np.zeros((120, 200, 3), np.float32); sth = np.zeros((120, 200), np.int32)
    for r in range(world):
        part = ds.render(cam, accel=2, rank=r, world=world)
        assert part["rgb8"].shape[0] == rows
        for lb in range(rows // 16):
            y0 = (lb * world + r) * 16
            if y0 >= 120:
                continue
            n = min(16, 120 - y0)
            st8[y0:y0 + n] = part["rgb8"][lb * 16:lb * 16 + n]
            stf[y0:y0 + n] = part["rgb32f"][lb * 16:lb * 16 + n]
            sth[y0:y0 + n] = part["hit_id"][lb * 16:lb * 16 + n]
    assert np.array_equal(st8, full["rgb8"]) and np.array_equal(sth, full["hit_id"])
    assert np.array_equal(stf.view(np.uint32), full["rgb32f"].view(np.uint32))
    ds.close()


def test_render_is_idempotent_and_device_outputs_match_host_outputs():
    torch = pytest.importorskip("torch")
    hs = P.HostScene(scene_path("balls_low"))
    hs.set_resolution(320, 200)
    ds = P.DeviceScene.from_host(hs)
    cam = hs.camera()
    a = ds.render(cam, accel=2)
    b = ds.render(cam, accel=2)
    assert np.array_equal(a["rgb8"], b["rgb8"]) and np.array_equal(a["rgb32f"].view(np.uint32), b["rgb32f"].view(np.uint32))
    dev8 = torch.zeros((200, 320, 3), dtype=torch.uint8, device="cuda")
    ds.render_device(cam, rgb8_ptr=dev8.data_ptr(), accel=2)
    ds.sync()
    assert np.array_equal(dev8.cpu().numpy(), a["rgb8"])
    ds.close()


def test_sample_frames_never_write_past_the_image_in_device_buffers():
    """spp > 0 into caller-owned DEVICE planes of exactly res_y rows (world == 1, include/p3d_hip.h) at a
    height that is not a multiple of the 16-row block: the rows the compact buffer is padded with must not
    be written (canary rows behind the image stay intact), and the image equals the host-buffer render."""
    torch = pytest.importorskip("torch")
    hs = P.HostScene(scene_path("mount_low"))
    W, H, pad = 96, 70, 12
    hs.set_resolution(W, H)
    ds = P.DeviceScene.from_host(hs)
    cam = hs.camera()
    smp = hs.samples(99, 2)
    ref = ds.render(cam, max_depth=4, accel=2, spp=2, samples=smp)
    dev8 = torch.full((H + pad, W, 3), 0xA5, dtype=torch.uint8, device="cuda")
    devf = torch.full((H + pad, W, 3), -7.0, dtype=torch.float32, device="cuda")
    devh = torch.full((H + pad, W), -99, dtype=torch.int32, device="cuda")
    for kw in (dict(tile=True), dict(wavefront=True), dict(tree=True)):
        ds.render_device(cam, rgb8_ptr=dev8.data_ptr(), rgb32f_ptr=devf.data_ptr(), hit_ptr=devh.data_ptr(),
                         max_depth=4, accel=2, spp=2, samples=smp, **kw)
        ds.sync()
        assert (dev8[H:] == 0xA5).all() and (devf[H:] == -7.0).all() and (devh[H:] == -99).all(), kw
        assert np.array_equal(dev8[:H].cpu().numpy(), ref["rgb8"])
        assert np.array_equal(devf[:H].cpu().numpy().view(np.uint32), ref["rgb32f"].view(np.uint32))
        assert np.array_equal(devh[:H].cpu().numpy(), ref["hit_id"])
    ds.close()


def test_small_workspace_budget_falls_back_from_the_tile_schedule():
    """The tile schedule needs one private slot per resident workgroup; when the budget holds too few it is
    the wavefront schedule (in bands) that renders -- same bits."""
    m = CASES["c2_mount_low_256x144_d4_bvh"]
    hs = P.HostScene(scene_path(m["scene"])); hs.set_resolution(*m["res"])
    ds = P.DeviceScene.from_host(hs)
    a = ds.render(hs.camera(), accel=2, tile=True)
    assert ds.last_schedule() == "tile"
    ds.set_tuning(workspace_mib=8)          # a depth-4 slot is 0.2 MB: only 41 workgroups would get one
    b = ds.render(hs.camera(), accel=2, tile=True)
    assert ds.last_schedule() == "wavefront"
    ds.set_tuning(workspace_mib=1)          # ... and not even one band of the wavefront queues fits 1 MiB
    c = ds.render(hs.camera(), accel=2, tile=True)
    assert ds.last_schedule() == "tree" and np.array_equal(a["rgb32f"].view(np.uint32), c["rgb32f"].view(np.uint32))
    assert np.array_equal(a["rgb32f"].view(np.uint32), b["rgb32f"].view(np.uint32)) and np.array_equal(a["hit_id"], b["hit_id"])
    ds.close()


@pytest.mark.parametrize("schedule", ["tile", "wavefront"])
def test_captured_frame_replays_identically(schedule):
    """p3d_render keeps no per-frame state on the host: a frame captured into a HIP graph can be replayed any
    number of times (tile schedule: the kernel re-arms its own tile counter; wavefront: the counter memset is
    part of the captured stream)."""
    torch = pytest.importorskip("torch")
    hs = P.HostScene(scene_path("balls_low"))
    hs.set_resolution(320, 200)
    cam = hs.camera()
    ds = P.DeviceScene.from_host(hs)
    ref = ds.render(cam, accel=2, max_depth=3)
    for kw in (dict([(schedule, True)]),):
        out8 = torch.zeros((200, 320, 3), dtype=torch.uint8, device="cuda")
        ds.render_device(cam, rgb8_ptr=out8.data_ptr(), accel=2, max_depth=3, **kw)      # sizes every workspace
        ds.sync()
        side = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            ds.set_stream(torch.cuda.current_stream().cuda_stream)
            ds.render_device(cam, rgb8_ptr=out8.data_ptr(), accel=2, max_depth=3, **kw)
        ds.set_stream(0)
        for k in range(3):                       # an odd number of replays of ONE graph
            out8.fill_(0)
            g.replay()
            torch.cuda.synchronize()
            assert np.array_equal(out8.cpu().numpy(), ref["rgb8"]), (kw, k)
        eager = ds.render(cam, accel=2, max_depth=3, counters=True, **kw)                 # and eager frames still work
        assert np.array_equal(eager["rgb8"], ref["rgb8"]) and eager["counters"]["pixels"] == 320 * 200
    ds.close()


def test_error_behaviour():
    hs = P.HostScene(scene_path("mount_low"))
    ds = P.DeviceScene.from_host(hs)
    cam = hs.camera()
    with pytest.raises(P.P3DError):
        ds.render(cam, max_depth=0)
    with pytest.raises(P.P3DError):
        ds.render(cam, spp=2, samples=None)
    with pytest.raises(P.P3DError):
        ds.render(cam, rank=3, world=2)
    ds.close()
    with pytest.raises(P.P3DError):
        P.DeviceScene.from_host(hs, device=99)


@pytest.mark.parametrize("width", [64, 50])      # 192-byte rows take the 16-byte path, 150-byte rows the byte path
def test_deinterleave_kernels_restore_the_frame(width):
    """Rank-0 side of SURVEY 8e on the device: compact per-rank tile buffers of a batch of frames ->
    full bottom-up frames, one launch (p3d_deinterleave_frames) or one per frame (p3d_deinterleave)."""
    import torch
    from u_4a_2s_p3d_raytracer_template2_amd import multigpu as MG
    hs = P.HostScene(scene_path("balls_low"))
    H, world, B, rb = 72, 3, 2, 16
    hs.set_resolution(width, H)
    cam = hs.camera()
    ds = P.DeviceScene.from_host(hs)
    rows = MG.padded_rows(H, rb, world)
    full = [ds.render(cam, max_depth=2 + f, accel=2, spp=0) for f in range(B)]
    gathered = torch.zeros((world, B, rows, width, 3), dtype=torch.uint8, device="cuda")
    for r in range(world):
        for f in range(B):
            ds.render_device(cam, rgb8_ptr=gathered[r, f].data_ptr(), max_depth=2 + f, accel=2, spp=0,
                             rank=r, world=world, row_block=rb)
    ds.sync()
    tile = rows * width * 3
    frames = torch.zeros((B, H, width, 3), dtype=torch.uint8, device="cuda")
    ds.deinterleave_frames(gathered.data_ptr(), frames.data_ptr(), width, H, rb, world, 3, B,
                           rank_stride_bytes=B * tile, tile_stride_bytes=tile)
    ds.sync()
    single = torch.zeros((H, width, 3), dtype=torch.uint8, device="cuda")
    for f in range(B):
        assert np.array_equal(frames[f].cpu().numpy(), full[f]["rgb8"])
        ds.deinterleave(gathered[0, f].data_ptr(), single.data_ptr(), width, H, rb, world, 3, rank_stride_bytes=B * tile)
        ds.sync()
        assert np.array_equal(single.cpu().numpy(), full[f]["rgb8"])
    ds.close()
