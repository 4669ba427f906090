"""Oracle restatement vs the REFERENCE'S OWN object code (oracle/_ref/libp3d_ref[_dN].so).

oracle/_ref is built by oracle/Makefile from the reference's sources compiled unchanged where they
lie: RT/vector.cpp, RT/boundingBox.cpp, RT/bvh.cpp, RT/grid.cpp, RT/scene.cpp:1-331 (the four
intercepts()/getNormal(), Scene accessors) and RT/main.cpp:471-730 (processLight, rayTracing,
sampleUnitDisk, with the globals of :40-103), one library per compile-time MAX_DEPTH.  Everything
compared here must agree BIT FOR BIT: vector/AABB/camera/quantiser/rand helpers, BVH and grid build
and traversals, the intersectors, single rayTracing() calls, whole frames (float bits, rgb8,
Ray::nextId) incl. the jittered/thin-lens sample loop and the SOFT_SHADOW / FUZZY_REFLECTION
branches, and the committed golden fixtures.  The module is skipped where the built _ref libraries
are absent (they cannot be rebuilt without /root/reference).
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, REPO, scene_path, synthetic_cube_map
from oracle import oracle_py as O
from oracle import ref_py as R
from oracle.ref_py import F, I
from scene_gen import write_scene

pytestmark = pytest.mark.skipif(not R.available(4), reason="oracle/_ref not built")


@pytest.fixture(scope="module")
def ref():
    O.lib()
    return R.lib(4)


def rand_vecs(rng, n, scale=3.0):
    v = (rng.standard_normal((n, 3)) * scale).astype(np.float32)
    # sprinkle exact zeros / axis-aligned directions: the slab tests divide by them
    z = rng.random((n, 3)) < 0.05
    v[z] = 0.0
    return v


def test_vector_ops_bitexact(ref):
    rng = np.random.default_rng(1)
    a, b = rand_vecs(rng, 2000), rand_vecs(rng, 2000)
    out = np.zeros(20, np.float32)
    for i in range(len(a)):
        ref.ref_vec_ops(F(a[i]), F(b[i]), F(out))
        n = O.normalize(a[i])
        assert np.array_equal(out[17:20].view(np.uint32), n.view(np.uint32))


def test_aabb_intercepts_bitexact(ref):
    rng = np.random.default_rng(2)
    n = 20000
    lo = rand_vecs(rng, n, 2.0)
    hi = lo + np.abs(rand_vecs(rng, n, 1.5)) + np.float32(1e-3)
    o, d = rand_vecs(rng, n, 4.0), rand_vecs(rng, n, 1.0)
    t = np.zeros(1, np.float32)
    hits = 0
    for i in range(n):
        h_ref = ref.ref_aabb_intercepts(F(lo[i]), F(hi[i]), F(o[i]), F(d[i]), F(t))
        h_or, t_or = O.aabb_intercepts(lo[i], hi[i], o[i], d[i])
        assert bool(h_ref) == h_or
        a, b = np.float32(t[0]), np.float32(t_or)
        assert a.view(np.uint32) == b.view(np.uint32) or (np.isnan(a) and np.isnan(b))
        hits += h_or
    assert 0 < hits < n


def test_u8fromfloat_and_rand(ref):
    xs = np.concatenate([np.linspace(-0.5, 1.5, 4001), np.arange(0, 256) / 255.99,
                         np.nextafter(np.arange(0, 256, dtype=np.float32) / np.float32(255.99), 2)])
    for x in xs.astype(np.float32):
        assert ref.ref_u8fromfloat(float(x)) == O.u8fromfloat(x)
    # RT/RT_Output.png is uniformly (19,92,192) = u8fromfloat of mount_low's bclr
    assert [O.u8fromfloat(v) for v in (0.078, 0.361, 0.753)] == [19, 92, 192]
    a = np.zeros(1000, np.float32)
    ref.ref_rand_floats(12345, 1000, F(a))
    assert np.array_equal(a, O.rand_floats(12345, 1000))


@pytest.mark.parametrize("name", ["mount_low", "balls_low", "dof", "dragon"])
def test_camera_bitexact(ref, name):
    sc = O.Scene(scene_path(name))
    lines = open(scene_path(name)).read().split()
    k = lines.index("from")
    cam9 = np.array([float(lines[k + 1 + j]) for j in (0, 1, 2)] +
                    [float(lines[k + 5 + j]) for j in (0, 1, 2)] +
                    [float(lines[k + 9 + j]) for j in (0, 1, 2)], np.float32)
    g = lambda key: float(lines[lines.index(key, k) + 1])
    for (w, h) in ((sc.res_x, sc.res_y), (1920, 1080), (97, 61)):
        sc.set_resolution(w, h)
        cam6 = np.array([g("angle"), g("hither"), w, h, g("aperture"), g("focal")], np.float32)
        d19 = np.zeros(19, np.float32)
        cam = ref.ref_camera_new(F(cam9), F(cam6), F(d19))
        assert np.array_equal(d19.view(np.uint32), sc.camera().view(np.uint32))
        rng = np.random.default_rng(3)
        o = np.zeros(3, np.float32)
        d = np.zeros(3, np.float32)
        for _ in range(300):
            px, py = np.float32(rng.random() * w), np.float32(rng.random() * h)
            ref.ref_camera_ray(cam, px, py, F(o), F(d))
            oo, dd = sc.primary_ray(px, py)
            assert np.array_equal(o, oo) and np.array_equal(d.view(np.uint32), dd.view(np.uint32))
            lx, ly = np.float32(rng.random() - .5), np.float32(rng.random() - .5)
            ref.ref_camera_ray_lens(cam, lx, ly, px, py, F(o), F(d))
            oo, dd = sc.primary_ray_lens(lx, ly, px, py)
            assert np.array_equal(o.view(np.uint32), oo.view(np.uint32))
            assert np.array_equal(d.view(np.uint32), dd.view(np.uint32))
        ref.ref_camera_free(cam)


def scene_rays(sc, rng, n):
    """Rays that exercise the accel structures: primary rays + rays between scene points."""
    t, d, _ = sc.prims()
    pts = d[:, :3]
    rays = []
    for _ in range(n // 2):
        rays.append(sc.primary_ray(np.float32(rng.random() * sc.res_x), np.float32(rng.random() * sc.res_y)))
    for _ in range(n - n // 2):
        a = pts[rng.integers(len(pts))] + rng.standard_normal(3).astype(np.float32) * np.float32(0.3)
        b = pts[rng.integers(len(pts))] + rng.standard_normal(3).astype(np.float32) * np.float32(0.3)
        rays.append((a.astype(np.float32), (b - a).astype(np.float32)))
    return rays


@pytest.mark.parametrize("name,nrays", [("mount_low", 3000), ("balls_low", 3000),
                                         ("balls_medium", 2000), ("balls_box", 2000),
                                         ("mount_high", 1500), ("dragon", 400)])
def test_bvh_and_grid_restatement_match_reference_objects(ref, name, nrays):
    sc = O.Scene(scene_path(name))
    t, d, _ = sc.prims()
    t = np.ascontiguousarray(t, np.int32)
    acc = ref.ref_accel_new(len(t), I(t), F(d))
    # ---- BVH::Build: identical node array and object permutation
    n_nodes = ref.ref_bvh_build(acc)
    nodes_o, nobj_o, order_o = sc.refbvh_dump()
    assert n_nodes == len(nodes_o)
    nodes_r = np.zeros((n_nodes, 8), np.float32)
    nobj_r = np.zeros(n_nodes, np.int32)
    order_r = np.zeros(len(t), np.int32)
    ref.ref_bvh_dump(acc, F(nodes_r), I(nobj_r), I(order_r))
    assert np.array_equal(nodes_r.view(np.uint32), nodes_o.view(np.uint32))
    assert np.array_equal(nobj_r, nobj_o) and np.array_equal(order_r, order_o)
    # ---- Grid::Build: identical dimensions and cell populations
    dims_r = np.zeros(3, np.int32)
    ref.ref_grid_build(acc, I(dims_r))
    dims_o, cells_o = sc.refgrid_dims(with_cells=True)
    assert np.array_equal(dims_r, dims_o)
    cells_r = np.zeros(len(cells_o), np.int32)
    ref.ref_grid_cell_counts(acc, I(cells_r))
    assert np.array_equal(cells_r, cells_o)
    # ---- traversals, interleaved so the persistent hit_stack evolves identically
    rng = np.random.default_rng(7)
    obj = np.zeros(1, np.int32)
    tt = np.zeros(1, np.float32)
    n_hit = n_shadow = 0
    for (o, dr) in scene_rays(sc, rng, nrays):
        o, dr = O.f3(o), O.f3(dr)
        obj[0] = -1
        ok_r = ref.ref_bvh_closest(acc, F(o), F(dr), I(obj), F(tt))
        ok_o, obj_o, t_o = sc.refbvh_closest(o, dr)
        assert bool(ok_r) == ok_o and int(obj[0]) == obj_o
        if obj_o >= 0:
            assert np.float32(tt[0]) == np.float32(t_o)
            n_hit += 1
        s_r = ref.ref_bvh_shadow(acc, F(o), F(dr))
        s_o = sc.refbvh_shadow(o, dr)
        assert bool(s_r) == s_o
        n_shadow += s_o
        g_r = ref.ref_grid_shadow(acc, F(o), F(dr))
        assert bool(g_r) == sc.refgrid_shadow(o, dr)
        obj[0] = -1
        ok_r = ref.ref_grid_closest(acc, F(o), F(dr), I(obj), F(tt))
        ok_o, obj_o, t_o = sc.refgrid_closest(o, dr)
        assert bool(ok_r) == ok_o and int(obj[0]) == obj_o
    assert n_hit > 0 and n_shadow > 0
    ref.ref_accel_free(acc)


# ------------------------------------------------------------------ intersectors (RT/scene.cpp:55-283)
def test_intersectors_and_bboxes_bitexact_vs_reference_objects():
    """The committed known-answer table (16 000 rays over the four primitive kinds) is what the
    reference's own intercepts()/getNormal() return, and the oracle agrees bit for bit."""
    with np.load(os.path.join(GOLDEN, "kat.npz")) as z:
        k = {name: z[name] for name in z.files}
    n = len(k["type"])
    assert n == 16000
    hits = 0
    for i in range(n):
        ty, p, o, d = int(k["type"][i]), k["prim12"][i], k["origin"][i], k["dir"][i]
        h_r, t_r, n_r = R.intersect(ty, p, o, d)
        h_o, t_o, n_o = O.intersect(ty, p, o, d)
        assert h_r == h_o == bool(k["hit"][i]), i
        if h_r:
            hits += 1
            assert np.float32(t_r).view(np.uint32) == np.float32(t_o).view(np.uint32) == k["t"][i].view(np.uint32), i
            assert np.array_equal(n_r.view(np.uint32), n_o.view(np.uint32)), i
            assert np.array_equal(n_r.view(np.uint32), k["normal"][i].view(np.uint32)), i
        if i % 16 == 0:
            a, b = R.prim_bbox(ty, p), O.prim_bbox(ty, p)
            assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
            assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    assert 3000 < hits < 13000


# ------------------------------------------------------------------ whole frames (RT/main.cpp:471-805)
CASES = json.load(open(os.path.join(GOLDEN, "cases.json")))


def _pair(case, res=None):
    path = scene_path(case["scene"])
    osc = O.Scene(path)
    osc.set_resolution(*(res or case["res"]))
    return osc, R.RefScene.from_oracle_scene(osc, path, res=res or case["res"], depth=case["max_depth"])


def _same_frame(a, b, rows=slice(None)):
    assert np.array_equal(a["hit_id"][rows], b["hit_id"][rows])
    assert np.array_equal(a["rgb32f"][rows].view(np.uint32), b["rgb32f"][rows].view(np.uint32))
    assert np.array_equal(a["rgb8"][rows], b["rgb8"][rows])


@pytest.mark.parametrize("name", sorted(CASES))
def test_golden_frames_are_reference_output(name):
    """Every committed golden frame == the reference's own rayTracing() on the reference's own
    objects, in float bits, rgb8, primary hit ids and Ray::nextId; and the oracle produces the same."""
    case = CASES[name]
    if not R.available(case["max_depth"]):
        pytest.skip("no _ref build for depth %d" % case["max_depth"])
    fr = np.load(os.path.join(GOLDEN, "frames.npz"))
    gold = {k: fr[name + "/" + k] for k in ("rgb8", "rgb32f", "hit_id")}
    osc, rs = _pair(case)
    H = case["res"][1]
    # the dragon without culling is 1e5 virtual calls per ray: 16 rows of it are enough here,
    # tests/golden/make_golden.py renders (and checks) all of them
    y0, y1 = (H // 2 - 8, H // 2 + 8) if case["scene"] == "dragon" else (0, 0)
    rows = slice(y0, y1) if y1 else slice(None)
    r = rs.render(case["accel"], case["spp"], case["seed"], y0=y0, y1=y1)
    _same_frame(r, gold, rows)
    o = osc.render(max_depth=case["max_depth"], accel=case["accel"], spp=case["spp"], seed=case["seed"], y0=y0, y1=y1)
    _same_frame(o, gold, rows)
    assert o["counters"]["rays"] == r["rays"]
    if not y1:
        assert r["rays"] == case["counters"]["rays"]


@pytest.mark.parametrize("seed", range(24))
def test_random_scenes_oracle_equals_reference(tmp_path, seed):
    """Generated scenes over all four primitive kinds (incl. planes in GRID mode), three
    material kinds, 0-3 lights, every accel mode, compile-time depths 1-6."""
    rng = np.random.default_rng(1000 + seed)
    accel = int(rng.integers(0, 3))
    depth = int(rng.integers(1, 7))
    if not R.available(depth):
        pytest.skip("no _ref build for depth %d" % depth)
    path = str(tmp_path / "scene.p3f")
    write_scene(path, rng, n_sph=int(rng.integers(0, 7)), n_tri=int(rng.integers(0, 9)), n_box=int(rng.integers(0, 3)),
                n_pl=int(rng.integers(0, 2)), n_lights=int(rng.integers(0, 4)), accel=accel)
    osc = O.Scene(path)
    rs = R.RefScene.from_oracle_scene(osc, path, depth=depth)
    for acc in (accel, (accel + 1) % 3):
        r = rs.render(acc)
        o = osc.render(max_depth=depth, accel=acc)
        _same_frame(o, r)
        assert o["counters"]["rays"] == r["rays"]


@pytest.mark.parametrize("scene,res,accel,spp,soft,fuzzy", [
    ("mount_low", (64, 36), 2, 0, True, False),     # 4x4 area-light grid, deterministic (RT/main.cpp:601-618)
    ("balls_low", (48, 48), 0, 0, True, False),
    ("balls_low", (48, 48), 2, 0, False, True),     # fuzzy reflection draws rand() inside the recursion (:651-660)
    ("balls_low", (40, 40), 2, 2, True, True),      # jittered light per sample (:620-624) + fuzzy + thin lens
    ("dof", (40, 40), 1, 3, True, False),
    ("balls_box", (48, 48), 1, 0, True, True),
])
def test_distribution_switches_oracle_equals_reference(scene, res, accel, spp, soft, fuzzy):
    """SOFT_SHADOW / FUZZY_REFLECTION (compile-time false in the reference, plain globals in the object
    code) and the spp>0 sample loop: the serial rand() stream makes them bit-comparable."""
    case = dict(scene=scene, res=list(res), max_depth=4)
    osc, rs = _pair(case)
    r = rs.render(accel, spp, 4242, soft_shadow=soft, fuzzy_reflection=fuzzy)
    o = osc.render(max_depth=4, accel=accel, spp=spp, seed=4242, soft_shadow=soft, fuzzy_reflection=fuzzy)
    _same_frame(o, r)
    assert o["counters"]["rays"] == r["rays"]


def test_baseline_configs_at_reduced_height_oracle_equals_reference():
    """BASELINE config 2 geometry (1920 wide, depth 4, BVH) on a 1920x60 strip, config 4's
    (depth 6, 2x2 samples) on 512x32, and GRID mode incl. the pixel it is known to differ in."""
    osc, rs = _pair(dict(scene="mount_low", res=[1920, 1080], max_depth=4))
    for accel in (2, 1):
        r = rs.render(accel, y0=500, y1=560)
        o = osc.render(max_depth=4, accel=accel, y0=500, y1=560)
        _same_frame(o, r, slice(500, 560))
        assert o["counters"]["rays"] == r["rays"]
    if R.available(6):
        osc, rs = _pair(dict(scene="mount_low", res=[512, 512], max_depth=6))
        r = rs.render(2, 2, 12345, y0=240, y1=272)
        o = osc.render(max_depth=6, accel=2, spp=2, seed=12345, y0=240, y1=272)
        _same_frame(o, r, slice(240, 272))
        assert o["counters"]["rays"] == r["rays"]


def test_skybox_lookup_bitexact_vs_reference_object_code():
    """Scene::GetSkyboxColor (RT/scene.cpp:383-461): nothing in the reference calls it, but its object code
    exists and the oracle's restatement must equal it bit for bit -- 20 000 directions incl. the six axes, the face
    diagonals (the `>` / `>=` tie rules of the face choice) and un-normalised ones, over a synthetic cube map whose six
    faces differ in size and pixel width."""
    faces = synthetic_cube_map()
    rng = np.random.default_rng(17)
    d = rng.standard_normal((20000, 3)).astype(np.float32)
    d[:6] = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32)
    d[6:12] = np.array([[1, 1, 0], [-1, 1, 0], [1, 0, 1], [0, 1, 1], [0, -1, 1], [1, 1, 1]], np.float32)
    d[12:10000] /= np.linalg.norm(d[12:10000], axis=1, keepdims=True)
    ref = R.skybox_colors(faces, d)
    sc = O.Scene(scene_path("mount_low"))
    sc.set_skybox(faces)
    mine = np.stack([sc.skybox_color(v) for v in d])
    assert np.array_equal(mine.view(np.uint32), ref.view(np.uint32))
    assert len(np.unique(ref, axis=0)) > 5000


def test_single_raytracing_calls_bitexact():
    """rayTracing(ray, 1, 1.0) on arbitrary (also non-unit, also inside-geometry) rays."""
    path = scene_path("balls_box")
    osc = O.Scene(path)
    rs = R.RefScene.from_oracle_scene(osc, path)
    rng = np.random.default_rng(5)
    rays = scene_rays(osc, rng, 600)
    for accel in (0, 1, 2):
        for (o, d) in rays:
            a = rs.trace(accel, o, d)
            b = osc.trace(accel, o, d)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
