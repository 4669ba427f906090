#!/usr/bin/env python3
"""Randomised parity soak (GPU box): the generator and checks of tests/test_gpu_random_scenes.py over many
more seeds than the suite runs, plus soft shadows.  usage: soak_random.py FIRST_SEED N"""
import os, sys, tempfile, traceback
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import test_gpu_random_scenes as T
from oracle import oracle_py as O
import u_4a_2s_p3d_raytracer_template2_amd as P

first, n = int(sys.argv[1]), int(sys.argv[2])
tmp = tempfile.mkdtemp(prefix="p3d_soak_")
bad = 0
worst = 0.0
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    accel = int(rng.integers(0, 3)); depth = int(rng.integers(1, 7))
    n_pl = 0 if accel == 1 else int(rng.integers(0, 2))
    path = os.path.join(tmp, "s%d.p3f" % seed)
    T.write_scene(path, rng, n_sph=int(rng.integers(0, 9)), n_tri=int(rng.integers(0, 12)), n_box=int(rng.integers(0, 4)),
                  n_pl=n_pl, n_lights=int(rng.integers(0, 4)), accel=accel)
    try:
        T.check(path, accel, depth, wavefront=True)
        T.check(path, accel, depth, tree=True, no_lds=True)
        # soft-shadow grid on the same scene
        sc = O.Scene(path)
        ref = sc.render(max_depth=depth, accel=accel, spp=0, soft_shadow=True)      # (GRID mode walks the reference's grid since round 2)
        hs = P.HostScene(path); ds = P.DeviceScene.from_host(hs)
        out = ds.render(hs.camera(), max_depth=depth, accel=accel, spp=0, soft_shadow=True, counters=True)
        ds.close()
        assert np.array_equal(out["hit_id"], ref["hit_id"])
        fin = np.isfinite(ref["rgb32f"])
        d = np.abs(out["rgb32f"][fin] - ref["rgb32f"][fin]).max() if fin.any() else 0.0
        worst = max(worst, float(d))
        assert d == 0.0 and out["counters"]["rays"] == ref["counters"]["rays"]
    except Exception:
        bad += 1
        print("seed %d FAILED (accel %d depth %d)" % (seed, accel, depth)); traceback.print_exc(limit=2)
    if (seed - first) % 50 == 49:
        print("... %d scenes, %d failures, worst soft-shadow |dRGB| %.2e" % (seed - first + 1, bad, worst), flush=True)
print("soak: %d scenes, %d failures, worst soft-shadow |dRGB| %.2e" % (n, bad, worst))
sys.exit(1 if bad else 0)
