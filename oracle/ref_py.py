"""ctypes binding of oracle/_ref/libp3d_ref[_dN].so: the REFERENCE'S OWN object code
(RT/scene.cpp:1-331, RT/main.cpp:471-730, vector/boundingBox/bvh/grid.cpp compiled unchanged by
oracle/Makefile) behind oracle/ref_harness.cpp.

TEST INFRASTRUCTURE ONLY.  Exists only where oracle/_ref was built, i.e. in the build container
(and on the GPU box as a prebuilt .so); used to pin the oracle restatement and to generate
tests/golden/*.npz.  The product package never imports this module.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(_HERE, "_ref")

fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int)


def so_path(depth=4):
    return os.path.join(REF_DIR, "libp3d_ref.so" if depth == 4 else "libp3d_ref_d%d.so" % depth)


def available(depth=4):
    return os.path.exists(so_path(depth))


def F(a):
    return a.ctypes.data_as(fp)


def I(a):
    return a.ctypes.data_as(ip)


_libs = {}


def lib(depth=4):
    """The reference build whose compile-time MAX_DEPTH (RT/main.cpp:34) is `depth`."""
    if depth in _libs:
        return _libs[depth]
    L = C.CDLL(so_path(depth))          # RTLD_LOCAL: the per-depth copies do not see each other
    L.ref_max_depth.restype = C.c_int
    assert L.ref_max_depth() == depth
    L.ref_vec_ops.argtypes = [fp, fp, fp]
    L.ref_aabb_intercepts.argtypes = [fp, fp, fp, fp, fp]
    L.ref_intersect.argtypes = [C.c_int, fp, fp, fp, fp, fp]
    L.ref_prim_bbox.argtypes = [C.c_int, fp, fp, fp]
    L.ref_camera_new.restype = C.c_void_p
    L.ref_camera_new.argtypes = [fp, fp, fp]
    L.ref_camera_free.argtypes = [C.c_void_p]
    L.ref_camera_ray.argtypes = [C.c_void_p, C.c_float, C.c_float, fp, fp]
    L.ref_camera_ray_lens.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, fp, fp]
    L.ref_u8fromfloat.restype = C.c_uint8
    L.ref_u8fromfloat.argtypes = [C.c_float]
    L.ref_rand_floats.argtypes = [C.c_uint, C.c_int, fp]
    L.ref_color_ops.argtypes = [fp, fp, fp]
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
    L.ref_scene_new.restype = C.c_void_p
    L.ref_scene_new.argtypes = [C.c_int, ip, fp, ip, C.c_int, fp, C.c_int, fp, fp, fp, fp]
    L.ref_scene_free.argtypes = [C.c_void_p]
    L.ref_scene_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_ulonglong)]
    L.ref_scene_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, fp, fp, fp]
    L.ref_skybox_colors.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.c_int,
                                    C.c_void_p, C.c_void_p]
    _libs[depth] = L
    return L


def parse_camera(p3f_path):
    """(cam9 = from,at,up ; dict angle/hither/aperture/focal/res) from the `v` block of a .p3f
    (grammar: SURVEY Appendix C, RT/scene.cpp:608-643) -- plain text parsing of an input asset."""
    tok = []
    for line in open(p3f_path):
        line = line.split("#", 1)[0]
        tok += line.split()
    k = tok.index("from")
    cam9 = np.array([float(tok[k + 1 + j]) for j in range(3)] +
                    [float(tok[k + 5 + j]) for j in range(3)] +
                    [float(tok[k + 9 + j]) for j in range(3)], np.float32)
    g = lambda key: float(tok[tok.index(key, k) + 1])
    r = tok.index("resolution", k)
    return cam9, dict(angle=g("angle"), hither=g("hither"), aperture=g("aperture"), focal=g("focal"),
                      res=(int(tok[r + 1]), int(tok[r + 2])))


def intersect(ptype, prim12, o, d, depth=4):
    p = np.zeros(12, np.float32)
    p[:len(prim12)] = prim12
    o = np.ascontiguousarray(o, np.float32)
    d = np.ascontiguousarray(d, np.float32)
    t = np.zeros(1, np.float32)
    n = np.zeros(3, np.float32)
    h = lib(depth).ref_intersect(int(ptype), F(p), F(o), F(d), F(t), F(n))
    return bool(h), float(t[0]), n


def prim_bbox(ptype, prim12):
    p = np.zeros(12, np.float32)
    p[:len(prim12)] = prim12
    mn = np.zeros(3, np.float32)
    mx = np.zeros(3, np.float32)
    lib().ref_prim_bbox(int(ptype), F(p), F(mn), F(mx))
    return mn, mx


class RefScene:
    """A scene made of the reference's own objects.  Geometry, materials and lights are the numbers
    the .p3f holds (taken from the oracle's loader dump, which keeps the loader form); the camera
    block is parsed from the file text."""

    def __init__(self, types, data12, material, mats12, lights6, bg3, cam9, angle, hither, res,
                 aperture, focal, depth=4):
        self.L = lib(depth)
        self.depth = depth
        self.res = (int(res[0]), int(res[1]))
        self._keep = [np.ascontiguousarray(types, np.int32), np.ascontiguousarray(data12, np.float32),
                      np.ascontiguousarray(material, np.int32), np.ascontiguousarray(mats12, np.float32),
                      np.ascontiguousarray(lights6, np.float32), np.ascontiguousarray(bg3, np.float32),
                      np.ascontiguousarray(cam9, np.float32),
                      np.array([angle, hither, res[0], res[1], aperture, focal], np.float32)]
        t, d, m, mm, li, bg, c9, c6 = self._keep
        self.h = self.L.ref_scene_new(len(t), I(t), F(d), I(m), len(mm), F(mm), len(li), F(li),
                                      F(bg), F(c9), F(c6))

    @classmethod
    def from_oracle_scene(cls, osc, p3f_path, res=None, depth=4):
        t, d, m = osc.prims()
        cam9, c = parse_camera(p3f_path)
        return cls(t, d, m, osc.materials(), osc.lights(), osc.bg(), cam9, c["angle"], c["hither"],
                   res if res is not None else (osc.res_x, osc.res_y), c["aperture"], c["focal"], depth)

    def close(self):
        if self.h:
            self.L.ref_scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, accel, spp=0, seed=12345, soft_shadow=False, fuzzy_reflection=False, y0=0, y1=0):
        W, H = self.res
        rgb8 = np.zeros((H, W, 3), np.uint8)
        f32 = np.zeros((H, W, 3), np.float32)
        hid = np.full((H, W), -2, np.int32)
        rays = C.c_ulonglong(0)
        self.L.ref_scene_render(self.h, int(accel), int(spp), int(seed), int(soft_shadow),
                                int(fuzzy_reflection), int(y0), int(y1), rgb8.ctypes.data, f32.ctypes.data,
                                hid.ctypes.data, C.byref(rays))
        return {"rgb8": rgb8, "rgb32f": f32, "hit_id": hid, "rays": int(rays.value)}

    def trace(self, accel, o, d, soft_shadow=False):
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        c = np.zeros(3, np.float32)
        self.L.ref_scene_trace(self.h, int(accel), int(soft_shadow), F(o), F(d), F(c))
        return c


def skybox_colors(faces, dirs, depth=4):
    """Scene::GetSkyboxColor (RT/scene.cpp:383-461, the reference's own object code) for n directions.
    faces: six uint8 arrays [H, W, 3 or 4] in the order right, left, top, bottom, front, back (RT/scene.cpp:337)."""
    faces = [np.ascontiguousarray(f, np.uint8) for f in faces]
    dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
    out = np.zeros_like(dirs)
    ptrs = (C.c_void_p * 6)(*[f.ctypes.data for f in faces])
    rx = (C.c_uint * 6)(*[f.shape[1] for f in faces])
    ry = (C.c_uint * 6)(*[f.shape[0] for f in faces])
    bpp = (C.c_uint * 6)(*[f.shape[2] for f in faces])
    lib(depth).ref_skybox_colors(ptrs, rx, ry, bpp, len(dirs), dirs.ctypes.data, out.ctypes.data)
    return out
