"""Synthetic scaling scene of SURVEY.md section 8d: N uniformly random spheres and triangles in a cube
seen by mount_low's camera, fixed seed.  The config scenes (12 .. 100 005 primitives) live in LDS / L2;
this one is the workload whose BVH and primitive records really stream from HBM (N = 1e5 .. 1e7).

The cube is [-1,1] x [-1,1] x [-0.6,1.4] (mount_low's extent, so its camera frames it); sphere radius
and triangle edge scale are N^(-1/3) (= 0.5 N^(-1/3) of the cube side, as in the survey).  Even scene
indices are spheres, odd ones triangles.  Four materials cycle over the primitives: two diffuse, one
mirror-like (Ks > 0 => reflection rays), one glass (T > 0 => "refraction" rays).

Note: the reference rejects triangles with |det| < 1e-3, so triangles with edges below
~0.03 are invisible to it -- and to this renderer; they still cost traversal and intersection tests.
"""
import numpy as np

CAMERA_LINES = ["v", "from -1.6 1.6 1.7", "at 0 0 0", "up 0 0 1", "angle 45", "hither 0.01",
                "resolution %d %d", "aperture 0", "focal 0.7"]
BACKGROUND = (0.078, 0.361, 0.753)
LIGHTS = np.array([[-100, -100, 100, 1.0, 1.0, 1.0],
                   [60, -20, 80, 0.6, 0.6, 0.5]], np.float32)
# diffuse rgb, Kd, specular rgb, Ks, shine, T, ior, reflection (= Ks, RT/scene.h:31)
MATERIALS = np.array([[0.80, 0.30, 0.25, 0.9, 1, 1, 1, 0.0, 50.0, 0.0, 1.0, 0.0],
                      [0.25, 0.60, 0.80, 0.8, 1, 1, 1, 0.0, 80.0, 0.0, 1.0, 0.0],
                      [0.70, 0.70, 0.70, 0.4, 1, 1, 1, 0.5, 120.0, 0.0, 1.0, 0.5],
                      [0.90, 0.90, 0.90, 0.1, 1, 1, 1, 0.2, 100.0, 0.9, 1.5, 0.2]], np.float32)
P3D_SPHERE, P3D_TRIANGLE = 0, 1


def arrays(n, seed=2024):
    """(prim_type u32[n], prim_data f32[n,12], prim_material u32[n], materials, lights, background)."""
    rng = np.random.default_rng(seed)
    r = np.float32(float(n) ** (-1.0 / 3.0))
    lo = np.array([-1.0, -1.0, -0.6], np.float32)
    centre = (lo + 2.0 * rng.random((n, 3), dtype=np.float32)).astype(np.float32)
    ptype = (np.arange(n, dtype=np.uint32) & 1).astype(np.uint32)            # even: sphere, odd: triangle
    data = np.zeros((n, 12), np.float32)
    sph = ptype == P3D_SPHERE
    data[sph, 0:3] = centre[sph]
    data[sph, 3] = r * (0.5 + rng.random(int(sph.sum()), dtype=np.float32))
    tri = ~sph
    nt = int(tri.sum())
    for k in range(3):
        data[tri, 3 * k:3 * k + 3] = centre[tri] + r * (2.0 * rng.random((nt, 3), dtype=np.float32) - 1.0)
    material = (rng.integers(0, len(MATERIALS), n)).astype(np.uint32)
    return ptype, data, material, MATERIALS.copy(), LIGHTS.copy(), BACKGROUND


def camera_p3f(path, res_x, res_y, accel=2):
    """A primitive-free .p3f holding only the camera/background/lights (for HostScene(...).camera())."""
    lines = ["accel %d" % accel, "spp 0", "bclr %g %g %g" % BACKGROUND] + CAMERA_LINES
    open(path, "w").write("\n".join(lines) % (res_x, res_y) + "\n")
    return path


def write_p3f(path, n, res_x, res_y, seed=2024, accel=2):
    """The same scene as arrays(n, seed) as a .p3f text file (%.9g round-trips every float32), so the
    oracle and the host loader read what the array path uploads.  Meant for test sizes (n <= ~1e5)."""
    ptype, data, material, mats, lights, bg = arrays(n, seed)
    g = lambda v: "%.9g" % float(v)
    out = ["accel %d" % accel, "spp 0", "bclr %g %g %g" % bg] + [s for s in CAMERA_LINES]
    out = ("\n".join(out) % (res_x, res_y)).split("\n")
    for li in lights:
        out.append("l " + " ".join(g(v) for v in li))
    cur = -1
    for i in range(n):
        if material[i] != cur:
            cur = int(material[i])
            m = mats[cur]
            out.append("f " + " ".join(g(v) for v in m[:11]))
        if ptype[i] == P3D_SPHERE:
            out.append("s " + " ".join(g(v) for v in data[i, :4]))
        else:
            out.append("p 3\n" + "\n".join(" ".join(g(v) for v in data[i, 3 * k:3 * k + 3]) for k in range(3)))
    open(path, "w").write("\n".join(out) + "\n")
    return path
