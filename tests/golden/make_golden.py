#!/usr/bin/env python3
"""Generate the committed golden fixtures from the REFERENCE'S OWN object code (oracle/_ref).

Run in the build container (needs /root/reference):  python tests/golden/make_golden.py
Every frame is rendered by the reference's rayTracing() over the reference's Sphere/Triangle/aaBox/
Plane objects (RT/main.cpp:471-730 and RT/scene.cpp:1-331 compiled unchanged, see oracle/Makefile and
oracle/ref_harness.cpp); every known answer comes from the reference's intercepts()/getNormal().
The script also asserts that the oracle restatement reproduces each of them bit for bit, and takes
the per-kind test counters (which the reference does not keep) from the oracle.

frames.npz   one entry per case "<name>": rgb8 [H,W,3] u8 (bottom row first), rgb32f [H,W,3] f32,
             hit_id [H,W] i32, plus counters; case parameters are in cases.json.
kat.npz      per-intersector known answers: type, prim12 (loader form), origin, dir -> hit, t, normal
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

from conftest import scene_path  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
from oracle import ref_py as R  # noqa: E402

# name, scene, (W,H), accel, spp, max_depth, seed
CASES = [
    # BASELINE config 1: balls_low 512x512 depth 2, no accel struct, one sample per pixel
    ("c1_balls_low_512_d2_none", "balls_low", (512, 512), 0, 0, 2, 0),
    # BASELINE config 2 scene at 1/7.5 scale: mount_low, BVH, depth 4
    ("c2_mount_low_256x144_d4_bvh", "mount_low", (256, 144), 2, 0, 4, 0),
    ("mount_low_256x144_d4_none", "mount_low", (256, 144), 0, 0, 4, 0),
    ("mount_low_256x144_d4_grid", "mount_low", (256, 144), 1, 0, 4, 0),
    # BASELINE config 3 scene: dragon, both shadow semantics
    ("c3_dragon_96_d4_bvh", "dragon", (96, 96), 2, 0, 4, 0),
    ("dragon_96_d4_none", "dragon", (96, 96), 0, 0, 4, 0),
    # BASELINE config 4 shape: mount_low depth 6, 2x2 jittered samples + thin lens
    ("c4_mount_low_96_d6_spp2", "mount_low", (96, 96), 2, 2, 6, 12345),
    # primitives the configs do not cover: aaBox (balls_box), Plane (balls_medium), dense tris
    ("balls_box_128_d4_bvh", "balls_box", (128, 128), 2, 0, 4, 0),
    ("balls_box_128_d4_none", "balls_box", (128, 128), 0, 0, 4, 0),
    ("balls_medium_128_d4_bvh", "balls_medium", (128, 128), 2, 0, 4, 0),
    ("balls_medium_128_d4_none", "balls_medium", (128, 128), 0, 0, 4, 0),
    ("mount_high_128_d4_bvh", "mount_high", (128, 128), 2, 0, 4, 0),
    # depth of field as shipped: dof.p3f (aperture 12, spp 4)
    ("dof_64_d4_spp4", "dof", (64, 64), 2, 4, 4, 777),
    # ragged sizes: not multiples of the 16x16 tile
    ("mount_low_37x23_d4_bvh", "mount_low", (37, 23), 2, 0, 4, 0),
    ("mount_low_1x1_d1", "mount_low", (1, 1), 2, 0, 1, 0),
    # GRID mode: per-cell closest hits, planes only inside their default [-1,1]^3 box, shadow rays
    # that miss the grid count as shadowed (RT/grid.cpp:327-328)
    ("balls_medium_128_d4_grid", "balls_medium", (128, 128), 1, 0, 4, 0),
    ("balls_box_128_d3_grid", "balls_box", (128, 128), 1, 0, 3, 0),
    ("dof_64_d4_spp2_grid", "dof", (64, 64), 1, 2, 4, 4321),
]


def kat(rng, n_each=4000):
    types, prims, orgs, dirs = [], [], [], []
    for kind in (O.SPHERE, O.TRIANGLE, O.BOX, O.PLANE):
        for _ in range(n_each):
            p = np.zeros(12, np.float32)
            if kind == O.SPHERE:
                p[:3] = rng.standard_normal(3) * 2
                p[3] = abs(rng.standard_normal()) * 0.7 + 0.01
                centre = p[:3]
            elif kind == O.BOX:
                lo = rng.standard_normal(3) * 2
                p[:3] = lo
                p[3:6] = lo + np.abs(rng.standard_normal(3)) + 0.01
                centre = (p[:3] + p[3:6]) / 2
            else:
                p[:9] = rng.standard_normal(9) * (2.0 if rng.random() < 0.7 else 0.05)
                centre = (p[0:3] + p[3:6] + p[6:9]) / 3
            o = (rng.standard_normal(3) * 4).astype(np.float32)
            if rng.random() < 0.1:
                o = (centre + rng.standard_normal(3) * 0.05).astype(np.float32)   # origins inside / on
            d = (centre + rng.standard_normal(3) * 0.6 - o).astype(np.float32)
            mode = rng.random()
            if mode < 0.5:
                d = O.normalize(d)
            elif mode < 0.6:
                d[rng.integers(3)] = 0.0                                       # axis-parallel
            types.append(kind); prims.append(p); orgs.append(o); dirs.append(d.astype(np.float32))
    types = np.array(types, np.int32)
    prims = np.array(prims, np.float32)
    orgs = np.array(orgs, np.float32)
    dirs = np.array(dirs, np.float32)
    hit = np.zeros(len(types), np.int32)
    t = np.zeros(len(types), np.float32)
    nrm = np.zeros((len(types), 3), np.float32)
    for i in range(len(types)):
        h, tt, n = R.intersect(types[i], prims[i], orgs[i], dirs[i])          # the reference's objects
        ho, to, no = O.intersect(types[i], prims[i], orgs[i], dirs[i])
        assert h == ho
        hit[i] = h
        if h:
            assert np.float32(tt).view(np.uint32) == np.float32(to).view(np.uint32)
            assert np.array_equal(n.view(np.uint32), no.view(np.uint32))
            t[i] = tt
            nrm[i] = n
    return dict(type=types, prim12=prims, origin=orgs, dir=dirs, hit=hit, t=t, normal=nrm)


def main():
    O.build()
    frames = {}
    meta = {}
    for (name, scene, (w, h), accel, spp, depth, seed) in CASES:
        sc = O.Scene(scene_path(scene))
        sc.set_resolution(w, h)
        rs = R.RefScene.from_oracle_scene(sc, scene_path(scene), res=(w, h), depth=depth)
        ref = rs.render(accel, spp, seed)                                      # the reference's rayTracing()
        r = sc.render(max_depth=depth, accel=accel, spp=spp, seed=seed, threads=1)
        for k in ("rgb8", "hit_id"):
            assert np.array_equal(ref[k], r[k]), (name, k)
        assert np.array_equal(ref["rgb32f"].view(np.uint32), r["rgb32f"].view(np.uint32)), name
        assert ref["rays"] == r["counters"]["rays"], name
        frames[name + "/rgb8"] = ref["rgb8"]
        frames[name + "/rgb32f"] = ref["rgb32f"]
        frames[name + "/hit_id"] = ref["hit_id"]
        meta[name] = dict(scene=scene, res=[w, h], accel=accel, spp=spp, max_depth=depth, seed=seed,
                          counters=r["counters"])
        print(name, r["counters"]["rays"], "rays")
    np.savez_compressed(os.path.join(HERE, "frames.npz"), **frames)
    with open(os.path.join(HERE, "cases.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    k = kat(np.random.default_rng(20261004))
    np.savez_compressed(os.path.join(HERE, "kat.npz"), **k)
    print("kat:", len(k["type"]), "cases,", int(k["hit"].sum()), "hits")


if __name__ == "__main__":
    main()
