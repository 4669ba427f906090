"""Small generated .p3f scenes shared by the oracle-vs-reference and the GPU-vs-oracle tests."""


def write_scene(path, rng, n_sph, n_tri, n_box, n_pl, n_lights, accel):
    L = ["accel %d" % accel, "spp 0", "bclr 0.1 0.3 0.6", "v", "from 4.0 3.0 2.5", "at 0 0 0.3", "up 0 0 1", "angle 50",
         "hither 0.01", "resolution 64 48", "aperture 0", "focal 1"]
    for _ in range(n_lights):
        p = rng.uniform(-6, 6, 3); p[2] = abs(p[2]) + 3
        L.append("l %.4f %.4f %.4f %.3f %.3f %.3f" % (*p, *rng.uniform(0.4, 1.0, 3)))

    def material():
        kind = rng.integers(0, 3)
        c = rng.uniform(0.1, 1, 3)
        if kind == 0:
            return "f %.3f %.3f %.3f %.2f 1 1 1 0 %.1f 0 1" % (*c, rng.uniform(0.4, 1), rng.uniform(5, 200))
        if kind == 1:
            return "f %.3f %.3f %.3f %.2f %.2f %.2f %.2f %.2f %.1f 0 1" % (*c, rng.uniform(0.2, 0.8), *rng.uniform(0.5, 1, 3),
                                                                     rng.uniform(0.2, 0.9), rng.uniform(10, 100))
        return "f %.3f %.3f %.3f 0.1 1 1 1 %.2f %.1f 1 %.3f" % (*c, rng.uniform(0.05, 0.3), rng.uniform(20, 120), rng.uniform(1.1, 1.8))

    for _ in range(n_pl):
        L.append(material())
        z = rng.uniform(-0.8, -0.3)
        L.append("pl 10 10 %.3f -10 10 %.3f -10 -10 %.3f" % (z, z, z))
    for _ in range(n_sph):
        L.append(material())
        L.append("s %.4f %.4f %.4f %.4f" % (*rng.uniform(-1.5, 1.5, 3), rng.uniform(0.15, 0.6)))
    for _ in range(n_box):
        L.append(material())
        lo = rng.uniform(-1.5, 1.0, 3)
        L.append("box %.4f %.4f %.4f %.4f %.4f %.4f" % (*lo, *(lo + rng.uniform(0.2, 0.8, 3))))
    for _ in range(n_tri):
        L.append(material())
        a = rng.uniform(-2, 2, 3)
        L.append("p 3\n%.4f %.4f %.4f\n%.4f %.4f %.4f\n%.4f %.4f %.4f" % (*a, *(a + rng.uniform(-1.5, 1.5, 3)), *(a + rng.uniform(-1.5, 1.5, 3))))
    open(path, "w").write("\n".join(L) + "\n")
