"""The multi-GPU frame through the C-ABI on the one GPU a test box has: rank/world sharding of
p3d_render, p3d_gather / p3d_gather_all over an RCCL communicator of size 1 (both ways of forming it),
p3d_deinterleave on rank 0.  RCCL refuses two ranks on one device, so world > 1 transfers are exercised by
the driver's multi-GPU run and by the two tests here that only run where two devices are visible (image == the
one-GPU image, byte for byte); the shard -> gather-layout -> de-interleave chain for world 2/3/8 is checked with the
transfers replaced by the device copies a gather amounts to."""
import numpy as np
import pytest

from conftest import scene_path
import u_4a_2s_p3d_raytracer_template2_amd as P

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _scene(w=200, h=120):
    hs = P.HostScene(scene_path("balls_low"))
    hs.set_resolution(w, h)
    return hs, P.DeviceScene.from_host(hs)


@pytest.mark.parametrize("how", ["create_all", "unique_id"])
def test_gather_world1_through_rccl_communicator(how):
    hs, ds = _scene()
    cam = hs.camera()
    ref = ds.render(cam, accel=2)["rgb8"]
    if how == "create_all":
        comm = P.Comm.create_all([0])[0]
    else:
        comm = P.Comm.create(P.comm_unique_id(), 0, 1, 0)
    assert comm.info() == (0, 1, 0)
    rows = P.local_rows(cam.res_y, 16, 1)
    tile = torch.zeros((rows, cam.res_x, 3), dtype=torch.uint8, device="cuda")
    gathered = torch.zeros_like(tile)
    frame = torch.zeros((cam.res_y, cam.res_x, 3), dtype=torch.uint8, device="cuda")
    ds.render_device(cam, rgb8_ptr=tile.data_ptr(), accel=2, rank=0, world=1)
    if how == "create_all":
        P.gather_all([comm], [ds], [tile.data_ptr()], gathered.data_ptr(), tile.numel())
    else:
        comm.gather(ds, tile.data_ptr(), gathered.data_ptr(), tile.numel())
    ds.deinterleave(gathered.data_ptr(), frame.data_ptr(), cam.res_x, cam.res_y, 16, 1, 3)
    ds.sync()
    assert np.array_equal(frame.cpu().numpy(), ref)
    with pytest.raises(P.P3DError):
        comm.gather(ds, 0, gathered.data_ptr(), tile.numel())          # NULL tile
    comm.close()
    ds.close()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_shards_in_gather_layout_deinterleave_to_the_single_gpu_frame(world):
    hs, ds = _scene(208, 150)
    cam = hs.camera()
    ref = ds.render(cam, accel=2, max_depth=3)["rgb8"]
    rows = P.local_rows(cam.res_y, 16, world)
    gathered = torch.full((world, rows, cam.res_x, 3), 7, dtype=torch.uint8, device="cuda")
    for r in range(world):       # what rank r would send lands at gathered + r * tile_bytes
        ds.render_device(cam, rgb8_ptr=gathered[r].data_ptr(), accel=2, max_depth=3, rank=r, world=world)
    frame = torch.zeros((cam.res_y, cam.res_x, 3), dtype=torch.uint8, device="cuda")
    ds.deinterleave(gathered.data_ptr(), frame.data_ptr(), cam.res_x, cam.res_y, 16, world, 3)
    ds.sync()
    assert np.array_equal(frame.cpu().numpy(), ref)
    ds.close()


needs_two = pytest.mark.skipif(P.device_count() < 2, reason="needs two GPUs (the test boxes of a round have one; "
                                                              "the driver's multi-GPU node runs these)")


@needs_two
def test_gather_all_over_two_devices_equals_the_single_gpu_frame():
    """The first real N > 1 transfer: two devices driven from one thread (p3d_comm_create_all + p3d_gather_all: grouped
    ncclSend / ncclRecv into rank 0's buffer over xGMI), de-interleaved on rank 0 == the one-GPU frame, byte for byte --
    and == the device-copy stand-in of the test above."""
    hs = P.HostScene(scene_path("balls_low"))
    hs.set_resolution(208, 150)
    cam = hs.camera()
    world = 2
    scenes = [P.DeviceScene.from_host(hs, device=r) for r in range(world)]
    ref = scenes[0].render(cam, accel=2, max_depth=3)["rgb8"]
    comms = P.Comm.create_all(list(range(world)))
    rows = P.local_rows(cam.res_y, 16, world)
    tiles = [torch.zeros((rows, cam.res_x, 3), dtype=torch.uint8, device="cuda:%d" % r) for r in range(world)]
    gathered = torch.full((world, rows, cam.res_x, 3), 7, dtype=torch.uint8, device="cuda:0")
    frame = torch.zeros((cam.res_y, cam.res_x, 3), dtype=torch.uint8, device="cuda:0")
    for r in range(world):
        scenes[r].render_device(cam, rgb8_ptr=tiles[r].data_ptr(), accel=2, max_depth=3, rank=r, world=world)
    P.gather_all(comms, scenes, [t.data_ptr() for t in tiles], gathered.data_ptr(), tiles[0].numel())
    scenes[0].deinterleave(gathered.data_ptr(), frame.data_ptr(), cam.res_x, cam.res_y, 16, world, 3)
    for sc in scenes:
        sc.sync()
    assert np.array_equal(frame.cpu().numpy(), ref)
    stand_in = torch.zeros_like(gathered)
    for r in range(world):
        scenes[0].render_device(cam, rgb8_ptr=stand_in[r].data_ptr(), accel=2, max_depth=3, rank=r, world=world)
    scenes[0].sync()
    assert torch.equal(stand_in, gathered)
    for c in comms:
        c.close()
    for sc in scenes:
        sc.close()


@needs_two
def test_cli_two_gpus_writes_the_one_gpu_image(tmp_path):
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(P.api.LIB_PATH), "p3d_render")
    a, b = str(tmp_path / "a.ppm"), str(tmp_path / "b.ppm")
    subprocess.check_call([exe, scene_path("mount_low"), "--res", "320", "200", "--accel", "2", "--out", a])
    subprocess.check_call([exe, scene_path("mount_low"), "--res", "320", "200", "--accel", "2", "--gpus", "2", "--out", b])
    assert open(a, "rb").read() == open(b, "rb").read()


def test_cli_renders_with_gpus_1(tmp_path):
    """p3d_render --gpus 1 and the plain call write the same image (the --gpus N > 1 path needs N devices)."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(P.api.LIB_PATH), "p3d_render")
    a, b = str(tmp_path / "a.ppm"), str(tmp_path / "b.ppm")
    subprocess.check_call([exe, scene_path("mount_low"), "--res", "160", "90", "--accel", "2", "--out", a])
    subprocess.check_call([exe, scene_path("mount_low"), "--res", "160", "90", "--accel", "2", "--gpus", "1", "--out", b])
    assert open(a, "rb").read() == open(b, "rb").read()
    rc = subprocess.call([exe, scene_path("mount_low"), "--res", "160", "90", "--gpus", "2", "--out", b],
                         stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ndev = P.device_count()
    assert (rc == 0) == (ndev >= 2)
