"""Oracle restatement vs the REFERENCE'S OWN objects (oracle/_ref/libp3d_ref.so).

oracle/_ref is built by oracle/Makefile from RT/vector.cpp, RT/boundingBox.cpp, RT/bvh.cpp
and RT/grid.cpp compiled in place (plus RT/camera.h, RT/maths.h, RT/color.h through the
harness).  Everything compared here must agree BIT FOR BIT.  The module is skipped where
the built _ref library is absent (it cannot be rebuilt without /root/reference).
"""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import REPO, scene_path
from oracle import oracle_py as O

REF_SO = os.path.join(REPO, "oracle", "_ref", "libp3d_ref.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built")

fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int)


def F(a):
    return a.ctypes.data_as(fp)


def I(a):
    return a.ctypes.data_as(ip)


@pytest.fixture(scope="module")
def ref():
    O.lib()
    L = C.CDLL(REF_SO)
    L.ref_camera_new.restype = C.c_void_p
    L.ref_camera_new.argtypes = [fp, fp, fp]
    L.ref_camera_free.argtypes = [C.c_void_p]
    L.ref_camera_ray.argtypes = [C.c_void_p, C.c_float, C.c_float, fp, fp]
    L.ref_camera_ray_lens.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, fp, fp]
    L.ref_u8fromfloat.restype = C.c_uint8
    L.ref_u8fromfloat.argtypes = [C.c_float]
    L.ref_accel_new.restype = C.c_void_p
    L.ref_accel_new.argtypes = [C.c_int, ip, fp]
    for n in ("ref_accel_free", "ref_bvh_build", "ref_bvh_stack_size"):
        getattr(L, n).argtypes = [C.c_void_p]
    L.ref_bvh_dump.argtypes = [C.c_void_p, fp, ip, ip]
    L.ref_bvh_shadow.argtypes = [C.c_void_p, fp, fp]
    L.ref_bvh_closest.argtypes = [C.c_void_p, fp, fp, ip, fp]
    L.ref_grid_build.argtypes = [C.c_void_p, ip]
    L.ref_grid_cell_counts.argtypes = [C.c_void_p, ip]
    L.ref_grid_shadow.argtypes = [C.c_void_p, fp, fp]
    L.ref_grid_closest.argtypes = [C.c_void_p, fp, fp, ip, fp]
    return L


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
