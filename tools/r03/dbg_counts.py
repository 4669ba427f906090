import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from conftest import scene_path
from oracle import oracle_py as O
import u_4a_2s_p3d_raytracer_template2_amd as P
RES=(96,64)
sc = O.Scene(scene_path("mount_low")); sc.set_resolution(*RES)
hs = P.HostScene(scene_path("mount_low")); hs.set_resolution(*RES)
ds = P.DeviceScene.from_host(hs)
for ss in (False, True):
    ref = sc.render(max_depth=4, accel=2, spp=0, soft_shadow=ss)
    print("soft", ss, "oracle", {k: ref["counters"][k] for k in ("closest_queries","shadow_queries")})
    for kw in (dict(), dict(tree=True), dict(wavefront=True), dict(tile=True), dict(no_lds=True)):
        out = ds.render(hs.camera(), max_depth=4, accel=2, spp=0, soft_shadow=ss, counters=True, **kw)
        c = out["counters"]
        print("   ", kw, ds.last_schedule(), {k: c[k] for k in ("closest_queries","shadow_queries","pixels")}, "img ok", np.array_equal(out["hit_id"], ref["hit_id"]))
