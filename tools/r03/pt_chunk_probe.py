#!/usr/bin/env python3
"""Path tracer (config 5): one launch of 256 frames per pixel strip against C concurrent launches of 256 / C frames each
(partial linear sums, added afterwards): is the launch bound by its longest strips?"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch
import u_4a_2s_p3d_raytracer_template2_amd as P

W, H, N = 1920, 1080, 256
for C in (1, 2, 4, 8, 16):
    pts = [P.PathTracer(device=0) for _ in range(C)]
    sts = [torch.cuda.Stream() for _ in range(C)]
    for pt, st in zip(pts, sts):
        pt.set_stream(st.cuda_stream)
    lins = [torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(C)]
    def run():
        for c, (pt, lin) in enumerate(zip(pts, lins)):
            pt.render_device(0, lin.data_ptr(), W, H, N // C, first_frame=c * (N // C), frame_stride=1)
        torch.cuda.synchronize()
    run()
    t0 = time.perf_counter()
    for _ in range(2):
        run()
    dt = (time.perf_counter() - t0) / 2
    tot = torch.stack(lins).sum(0)
    print("C = %2d launches of %3d frames: %.1f ms per 256-sample image  %.0f Msamples/s   mean %.6f" % (
        C, N // C, dt * 1e3, W * H * N / dt / 1e6, float(torch.nan_to_num(tot / N).mean())), flush=True)
    for pt in pts:
        pt.close()
