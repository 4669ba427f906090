"""Distribution-ray-tracing switches of RT/main.cpp:41,43 on the GPU.

SOFT_SHADOW without anti-aliasing is the reference's deterministic 4x4 sub-light grid: compared with
the oracle like any Whitted frame.  SOFT_SHADOW with anti-aliasing and FUZZY_REFLECTION draw random
numbers inside the recursion -- the reference from one serial rand() stream, the GPU from per-node
counter-based streams -- so those are compared statistically: the mean image over K seeds must agree
within the standard error of the two sample means.  PARITY UNPINNED beyond the oracle: the reference
holds no output with these switches on (they are compile-time false there)."""
import numpy as np
import pytest

from conftest import scene_path, synthetic_cube_map, assert_rgb8_equal, RGB_TOL
from oracle import oracle_py as O
import u_4a_2s_p3d_raytracer_template2_amd as P
from u_4a_2s_p3d_raytracer_template2_amd import multigpu as MG

pytestmark = pytest.mark.gpu

RES = (64, 48)


def load(name):
    sc = O.Scene(scene_path(name)); sc.set_resolution(*RES)
    hs = P.HostScene(scene_path(name)); hs.set_resolution(*RES)
    return sc, hs, P.DeviceScene.from_host(hs)


@pytest.mark.parametrize("name,accel,depth", [("mount_low", 2, 4), ("balls_low", 0, 3), ("balls_low", 1, 2), ("balls_medium", 2, 3)])
def test_soft_shadow_grid_matches_oracle(name, accel, depth):
    sc, hs, ds = load(name)
    ref = sc.render(max_depth=depth, accel=accel, spp=0, soft_shadow=True)
    plain = sc.render(max_depth=depth, accel=accel, spp=0)
    assert ref["counters"]["shadow_queries"] > 8 * plain["counters"]["shadow_queries"]      # 16 sub-lights per light
    for kw in (dict(), dict(tree=True), dict(wavefront=True), dict(no_lds=True), dict(packet=True)):
        out = ds.render(hs.camera(), max_depth=depth, accel=accel, spp=0, soft_shadow=True, counters=True, **kw)
        assert np.array_equal(out["hit_id"], ref["hit_id"])
        assert np.abs(out["rgb32f"] - ref["rgb32f"]).max() <= RGB_TOL
        assert out["counters"]["rays"] == ref["counters"]["rays"]
        d8 = np.abs(out["rgb8"].astype(int) - ref["rgb8"].astype(int))
        assert d8.max() <= 1 and (d8 != 0).mean() <= 1e-3
    ds.close()


@pytest.mark.parametrize("name,accel,depth", [("mount_low", 2, 4), ("balls_low", 0, 3), ("balls_box", 1, 3), ("dragon", 2, 3)])
def test_skybox_frames_match_oracle(name, accel, depth):
    """P3D_FEATURE_SKYBOX: rays that hit nothing return Scene::GetSkyboxColor(ray) (RT/scene.cpp:383-461) from a synthetic
    cube map (the reference's asset directory ships none).  The lookup is pinned bit for bit against the reference's
    object code on the CPU (tests/test_oracle_vs_ref.py); here whole frames -- primary misses, reflected and "refracted"
    misses -- against the oracle with the same switch, on both schedules that carry features and both scene placements."""
    sc, hs, ds = load(name)
    faces = synthetic_cube_map()
    sc.set_skybox(faces)
    ds.set_skybox(faces)
    ref = sc.render(max_depth=depth, accel=accel, spp=0, skybox=True, threads=8 if accel != 1 else 1)
    plain = sc.render(max_depth=depth, accel=accel, spp=0)
    assert (ref["rgb8"] != plain["rgb8"]).any(axis=2).mean() > 0.05            # the sky shows (and reflects)
    for kw in (dict(), dict(wavefront=True), dict(tile=True), dict(wavefront=True, no_lds=True), dict(tile=True, no_lds=True, private_walk=True)):
        out = ds.render(hs.camera(), max_depth=depth, accel=accel, spp=0, skybox=True, counters=True, **kw)
        assert np.array_equal(out["hit_id"], ref["hit_id"]), kw
        assert np.abs(out["rgb32f"] - ref["rgb32f"]).max() <= RGB_TOL, kw
        assert out["counters"]["rays"] == ref["counters"]["rays"], kw
        assert_rgb8_equal(out["rgb8"], ref["rgb8"], str(kw))
    with pytest.raises(P.P3DError):
        ds.render(hs.camera(), max_depth=depth, accel=accel, skybox=True, tree=True)      # features need tile / wavefront
    ds.close()
    hs2 = P.HostScene(scene_path(name)); hs2.set_resolution(*RES)
    ds2 = P.DeviceScene.from_host(hs2)
    with pytest.raises(P.P3DError):
        ds2.render(hs2.camera(), skybox=True)                                               # no cube map given
    ds2.close()


def mean_and_var(frames):
    a = np.stack(frames).astype(np.float64)
    return a.mean(0), a.var(0, ddof=1)


@pytest.mark.parametrize("feature,spp", [("fuzzy_reflection", 0), ("soft_shadow", 2), ("both", 2)])
def test_random_features_match_oracle_statistically(feature, spp):
    K = 24
    sc, hs, ds = load("balls_low")
    kw = dict(soft_shadow=feature in ("soft_shadow", "both"), fuzzy_reflection=feature in ("fuzzy_reflection", "both"))
    cpu, gpu = [], []
    for seed in range(K):
        cpu.append(sc.render(max_depth=3, accel=2, spp=spp, seed=1000 + seed, want_hit=False, **kw)["rgb32f"])
        samples = hs.samples(2000 + seed, spp) if spp else None
        gpu.append(ds.render(hs.camera(), max_depth=3, accel=2, spp=spp, samples=samples, seed=seed, want_hit=False, **kw)["rgb32f"])
    mc, vc = mean_and_var(cpu)
    mg, vg = mean_and_var(gpu)
    assert vg.mean() > 0.2 * vc.mean() and vc.mean() > 0.2 * vg.mean(), "per-pixel variances are of different orders"
    se = np.sqrt((vc + vg) / K)
    bad = np.abs(mc - mg) > 5.0 * se + 0.01
    assert bad.mean() <= 0.005, "%.2f%% of the channel means differ by more than 5 standard errors" % (100 * bad.mean())
    assert abs(mc.mean() - mg.mean()) <= 3e-3
    # the switch does something: the mean image differs from the plain Whitted frame
    plain = ds.render(hs.camera(), max_depth=3, accel=2, spp=0, want_hit=False)["rgb32f"]
    if spp == 0:
        assert np.abs(mg - plain).mean() > 1e-3
    ds.close()


def test_random_streams_are_reproducible_and_shard_independent():
    _, hs, ds = load("balls_low")
    cam = hs.camera()
    kw = dict(max_depth=3, accel=2, spp=2, samples=hs.samples(7, 2), soft_shadow=True, fuzzy_reflection=True, seed=42)
    a = ds.render(cam, **kw)
    b = ds.render(cam, **kw)
    assert np.array_equal(a["rgb32f"], b["rgb32f"])
    c = ds.render(cam, **dict(kw, seed=43))
    assert not np.array_equal(a["rgb32f"], c["rgb32f"])
    # two ranks' interleaved row blocks stitched = the one-GPU frame: streams are keyed by the pixel's
    # position in the full frame
    parts = [ds.render(cam, rank=r, world=2, row_block=16, **kw)["rgb32f"] for r in range(2)]
    full = MG.stitch_reference(parts, RES[1], 16)
    assert np.array_equal(full, a["rgb32f"])
    # LDS-resident and HBM-resident scene paths draw the same numbers
    d = ds.render(cam, no_lds=True, **kw)
    assert np.array_equal(d["rgb32f"], a["rgb32f"])
    # ... and so do the tile schedule and the wavefront schedule
    for sched in ("tile", "wavefront"):
        e = ds.render(cam, **dict(kw, **{sched: True}))
        assert ds.last_schedule() == sched and np.array_equal(e["rgb32f"], a["rgb32f"])
    ds.close()


def test_feature_argument_errors():
    _, hs, ds = load("mount_low")
    with pytest.raises(P.P3DError):
        ds.render(hs.camera(), fuzzy_reflection=True, tree=True)
    ds.close()
