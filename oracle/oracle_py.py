"""ctypes binding of the CPU oracle (oracle/libp3d_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libp3d_oracle.so")

SPHERE, TRIANGLE, BOX, PLANE = 0, 1, 2, 3


class Params(C.Structure):
    _fields_ = [("max_depth", C.c_int32), ("accel", C.c_int32), ("spp", C.c_int32),
                ("seed", C.c_uint32), ("threads", C.c_int32), ("break_fixed", C.c_int32),
                ("y0", C.c_int32), ("y1", C.c_int32), ("soft_shadow", C.c_int32), ("fuzzy_reflection", C.c_int32),
                ("skybox", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("rays", "closest_queries", "shadow_queries", "aabb_tests", "sphere_tests",
                 "tri_tests", "box_tests", "plane_tests", "get_object")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build():
    """Compile the oracle (and oracle/_ref when the reference tree is present)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        ip = C.POINTER(C.c_int32)
        L.p3o_scene_load.restype = C.c_void_p
        L.p3o_scene_load.argtypes = [C.c_char_p]
        L.p3o_scene_free.argtypes = [C.c_void_p]
        L.p3o_scene_info.argtypes = [C.c_void_p, ip]
        L.p3o_scene_set_resolution.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.p3o_scene_prims.argtypes = [C.c_void_p, ip, fp, ip]
        L.p3o_scene_materials.argtypes = [C.c_void_p, fp]
        L.p3o_scene_lights.argtypes = [C.c_void_p, fp]
        L.p3o_scene_bg.argtypes = [C.c_void_p, fp]
        L.p3o_scene_camera.argtypes = [C.c_void_p, fp]
        L.p3o_render.restype = C.c_int
        L.p3o_render.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.POINTER(Counters)]
        L.p3o_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, fp, fp, fp]
        L.p3o_scene_set_skybox.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.p3o_skybox_color.argtypes = [C.c_void_p, fp, fp]
        L.p3o_intersect.restype = C.c_int
        L.p3o_intersect.argtypes = [C.c_int, fp, fp, fp, fp, fp]
        L.p3o_aabb_intercepts.restype = C.c_int
        L.p3o_aabb_intercepts.argtypes = [fp, fp, fp, fp, fp]
        L.p3o_prim_bbox.argtypes = [C.c_int, fp, fp, fp]
        L.p3o_normalize.argtypes = [fp]
        L.p3o_primary_ray.argtypes = [C.c_void_p, C.c_float, C.c_float, fp, fp]
        L.p3o_primary_ray_lens.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float,
                                           C.c_float, fp, fp]
        L.p3o_u8fromfloat.restype = C.c_uint8
        L.p3o_u8fromfloat.argtypes = [C.c_float]
        L.p3o_rand_floats.argtypes = [C.c_uint32, C.c_int32, fp]
        L.p3o_refbvh_node_count.restype = C.c_int32
        L.p3o_refbvh_node_count.argtypes = [C.c_void_p]
        L.p3o_refbvh_dump.argtypes = [C.c_void_p, fp, ip, ip]
        for name in ("p3o_refbvh_shadow", "p3o_refgrid_shadow"):
            f = getattr(L, name)
            f.restype = C.c_int
            f.argtypes = [C.c_void_p, fp, fp]
        for name in ("p3o_refbvh_closest", "p3o_refgrid_closest"):
            f = getattr(L, name)
            f.restype = C.c_int
            f.argtypes = [C.c_void_p, fp, fp, ip, fp]
        L.p3o_refgrid_dims.argtypes = [C.c_void_p, ip, ip]
        # GLSL path tracer restatement (pt_oracle.cpp)
        L.pto_base_hash.restype = C.c_uint32
        L.pto_base_hash.argtypes = [C.c_uint32, C.c_uint32]
        L.pto_hash_stream.argtypes = [C.c_float, C.c_int, fp, fp]
        L.pto_sample.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, fp]
        L.pto_render.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, fp, fp]
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def f3(v):
    return np.ascontiguousarray(v, dtype=np.float32)


class Scene:
    """A .p3f scene loaded by the oracle's own parser (RT/scene.cpp:476-675 grammar)."""

    def __init__(self, path):
        self.h = lib().p3o_scene_load(os.fsencode(path))
        if not self.h:
            raise IOError("oracle could not load %s" % path)
        self._info()

    def _info(self):
        out = np.zeros(8, np.int32)
        lib().p3o_scene_info(self.h, _i(out))
        (self.n_prims, self.n_lights, self.n_materials, self.res_x, self.res_y, self.accel,
         self.spp, self.parse_ok) = [int(v) for v in out]

    def close(self):
        if self.h:
            lib().p3o_scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_resolution(self, w, h):
        lib().p3o_scene_set_resolution(self.h, int(w), int(h))
        self._info()

    def prims(self):
        t = np.zeros(self.n_prims, np.int32)
        d = np.zeros((self.n_prims, 12), np.float32)
        m = np.zeros(self.n_prims, np.int32)
        lib().p3o_scene_prims(self.h, _i(t), _f(d), _i(m))
        return t, d, m

    def materials(self):
        m = np.zeros((self.n_materials, 12), np.float32)
        lib().p3o_scene_materials(self.h, _f(m))
        return m

    def lights(self):
        li = np.zeros((self.n_lights, 6), np.float32)
        lib().p3o_scene_lights(self.h, _f(li))
        return li

    def bg(self):
        b = np.zeros(3, np.float32)
        lib().p3o_scene_bg(self.h, _f(b))
        return b

    def camera(self):
        c = np.zeros(19, np.float32)
        lib().p3o_scene_camera(self.h, _f(c))
        return c

    def primary_ray(self, px, py):
        o = np.zeros(3, np.float32)
        d = np.zeros(3, np.float32)
        lib().p3o_primary_ray(self.h, px, py, _f(o), _f(d))
        return o, d

    def primary_ray_lens(self, lx, ly, px, py):
        o = np.zeros(3, np.float32)
        d = np.zeros(3, np.float32)
        lib().p3o_primary_ray_lens(self.h, lx, ly, px, py, _f(o), _f(d))
        return o, d

    def set_skybox(self, faces):
        """faces: six uint8 arrays [H, W, 3 or 4] (right, left, top, bottom, front, back; row 0 = bottom row)."""
        faces = [np.ascontiguousarray(f, np.uint8) for f in faces]
        ptrs = (C.c_void_p * 6)(*[f.ctypes.data for f in faces])
        rx = (C.c_uint32 * 6)(*[f.shape[1] for f in faces])
        ry = (C.c_uint32 * 6)(*[f.shape[0] for f in faces])
        bpp = (C.c_uint32 * 6)(*[f.shape[2] for f in faces])
        lib().p3o_scene_set_skybox(self.h, ptrs, rx, ry, bpp)

    def skybox_color(self, d):
        d = f3(d)
        c = np.zeros(3, np.float32)
        lib().p3o_skybox_color(self.h, _f(d), _f(c))
        return c

    def render(self, max_depth=4, accel=-1, spp=-1, seed=12345, threads=1, break_fixed=0,
               want_f32=True, want_hit=True, y0=0, y1=0, soft_shadow=False, fuzzy_reflection=False, skybox=False):
        """Returns dict(rgb8 [H,W,3] u8 bottom row first, rgb32f, hit_id, counters)."""
        W, H = self.res_x, self.res_y
        rgb8 = np.zeros((H, W, 3), np.uint8)
        f32 = np.zeros((H, W, 3), np.float32) if want_f32 else None
        hid = np.full((H, W), -2, np.int32) if want_hit else None
        prm = Params(max_depth, accel, spp, seed, threads, break_fixed, y0, y1, int(soft_shadow), int(fuzzy_reflection), int(skybox))
        ctr = Counters()
        rc = lib().p3o_render(self.h, C.byref(prm), rgb8.ctypes.data,
                              f32.ctypes.data if want_f32 else None,
                              hid.ctypes.data if want_hit else None, C.byref(ctr))
        if rc != 0:
            raise RuntimeError("p3o_render failed: %d" % rc)
        return {"rgb8": rgb8, "rgb32f": f32, "hit_id": hid, "counters": ctr.as_dict()}

    def trace(self, accel, o, d, max_depth=4, soft_shadow=False):
        """One rayTracing(ray, 1, 1.0) call: unclamped colour."""
        o, d = f3(o), f3(d)
        c = np.zeros(3, np.float32)
        lib().p3o_trace(self.h, int(accel), int(max_depth), int(soft_shadow), _f(o), _f(d), _f(c))
        return c

    # reference-BVH / grid restatement probes
    def refbvh_dump(self):
        n = lib().p3o_refbvh_node_count(self.h)
        nodes = np.zeros((n, 8), np.float32)
        nobj = np.zeros(n, np.int32)
        order = np.zeros(self.n_prims, np.int32)
        lib().p3o_refbvh_dump(self.h, _f(nodes), _i(nobj), _i(order))
        return nodes, nobj, order

    def refbvh_shadow(self, o, d):
        o, d = f3(o), f3(d)
        return bool(lib().p3o_refbvh_shadow(self.h, _f(o), _f(d)))

    def refbvh_closest(self, o, d):
        o, d = f3(o), f3(d)
        obj = np.zeros(1, np.int32)
        t = np.zeros(1, np.float32)
        ok = lib().p3o_refbvh_closest(self.h, _f(o), _f(d), _i(obj), _f(t))
        return bool(ok), int(obj[0]), float(t[0])

    def refgrid_dims(self, with_cells=False):
        d = np.zeros(3, np.int32)
        lib().p3o_refgrid_dims(self.h, _i(d), None)
        if not with_cells:
            return d
        cc = np.zeros(int(d[0]) * int(d[1]) * int(d[2]), np.int32)
        lib().p3o_refgrid_dims(self.h, _i(d), _i(cc))
        return d, cc

    def refgrid_shadow(self, o, d):
        o, d = f3(o), f3(d)
        return bool(lib().p3o_refgrid_shadow(self.h, _f(o), _f(d)))

    def refgrid_closest(self, o, d):
        o, d = f3(o), f3(d)
        obj = np.zeros(1, np.int32)
        t = np.zeros(1, np.float32)
        ok = lib().p3o_refgrid_closest(self.h, _f(o), _f(d), _i(obj), _f(t))
        return bool(ok), int(obj[0]), float(t[0])


def intersect(ptype, prim12, o, d):
    """(hit, t, unit normal at hit) through the restated intersectors."""
    p = np.zeros(12, np.float32)
    p[:len(prim12)] = prim12
    o, d = f3(o), f3(d)
    t = np.zeros(1, np.float32)
    n = np.zeros(3, np.float32)
    h = lib().p3o_intersect(int(ptype), _f(p), _f(o), _f(d), _f(t), _f(n))
    return bool(h), float(t[0]), n


def aabb_intercepts(mn, mx, o, d):
    mn, mx, o, d = f3(mn), f3(mx), f3(o), f3(d)
    t = np.zeros(1, np.float32)
    h = lib().p3o_aabb_intercepts(_f(mn), _f(mx), _f(o), _f(d), _f(t))
    return bool(h), float(t[0])


def prim_bbox(ptype, prim12):
    p = np.zeros(12, np.float32)
    p[:len(prim12)] = prim12
    mn = np.zeros(3, np.float32)
    mx = np.zeros(3, np.float32)
    lib().p3o_prim_bbox(int(ptype), _f(p), _f(mn), _f(mx))
    return mn, mx


def normalize(v):
    a = f3(v).copy()
    lib().p3o_normalize(_f(a))
    return a


def u8fromfloat(x):
    return int(lib().p3o_u8fromfloat(float(np.float32(x))))


def rand_floats(seed, n):
    out = np.zeros(n, np.float32)
    lib().p3o_rand_floats(int(seed), int(n), _f(out))
    return out


# ---- GLSL path tracer restatement (oracle/pt_oracle.cpp; PARITY UNPINNED, see its header)
def pt_base_hash(a, b):
    return int(lib().pto_base_hash(int(a) & 0xFFFFFFFF, int(b) & 0xFFFFFFFF))


def pt_hash_stream(seed, n):
    out = np.zeros((n, 3), np.float32)
    s = np.zeros(1, np.float32)
    lib().pto_hash_stream(float(np.float32(seed)), int(n), _f(out), _f(s))
    return out, float(s[0])


def pt_sample(res_x, res_y, x, y, itime, mouse=(0.0, 0.0)):
    rgb = np.zeros(3, np.float32)
    lib().pto_sample(int(res_x), int(res_y), int(x), int(y), float(np.float32(itime)), float(mouse[0]), float(mouse[1]), _f(rgb))
    return rgb


def pt_render(res_x, res_y, n_frames, time0=0.0, dt=1.0 / 60.0, threads=1, want_sum=True, mouse=(0.0, 0.0)):
    rgba = np.zeros((res_y, res_x, 4), np.float32)
    lin = np.zeros((res_y, res_x, 3), np.float32) if want_sum else None
    lib().pto_render(int(res_x), int(res_y), int(n_frames), float(np.float32(time0)), float(np.float32(dt)),
                     float(mouse[0]), float(mouse[1]), int(threads),
                     _f(rgba), _f(lin) if want_sum else None)
    return rgba, lin
