"""GPU tests of the Shadertoy path tracer port (BASELINE config 5).

PARITY UNPINNED against the reference: no GLSL compiler / GL driver exists here and the reference
holds no output of the shader but a screenshot.  What is checked is agreement between two
independent implementations of the same GLSL text -- the HIP kernel and oracle/pt_oracle.cpp:
the integer hash RNG bit for bit, single frames pixel by pixel (same RNG stream => same path; the
few pixels where a last-bit difference of sin/cos/pow flips a branch are counted), and the
converged image statistically."""
import numpy as np
import pytest

from oracle import oracle_py as O
import u_4a_2s_p3d_raytracer_template2_amd as P

pytestmark = pytest.mark.gpu


def test_integer_hash_is_bit_exact():
    rng = np.random.default_rng(5)
    a = rng.integers(0, 2**32, 4096, dtype=np.uint64).astype(np.uint32)
    b = rng.integers(0, 2**32, 4096, dtype=np.uint64).astype(np.uint32)
    got = P.pt_debug_hash(a, b)
    exp = np.array([O.pt_base_hash(int(x), int(y)) for x, y in zip(a, b)], np.uint32)
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("mouse", [(0.0, 0.0), (80.0, 54.0)])
def test_single_frames_follow_the_same_paths(mouse):
    W, H = 160, 90
    pt = P.PathTracer()
    bad_total = 0
    for k in (0, 1, 7):
        rgba, lin = pt.render(W, H, 1, first_frame=k, mouse=mouse)
        itime = np.float32(0.0) + np.float32(k) * np.float32(1.0 / 60.0)       # iTime of frame k, as the kernel forms it
        ref_rgba, ref_lin = O.pt_render(W, H, 1, time0=float(itime), threads=8, mouse=mouse)
        ok = np.isfinite(ref_lin).all(axis=2) & np.isfinite(lin).all(axis=2)
        assert (np.isfinite(ref_lin).all(axis=2) == np.isfinite(lin).all(axis=2)).mean() > 0.999
        d = np.abs(lin - ref_lin).max(axis=2)
        bad = (d > 1e-3) & ok
        bad_total += int(bad.sum())
        # the overwhelming majority of pixels take exactly the same path and agree to rounding
        assert np.median(d[ok]) < 1e-5
        assert bad.mean() < 0.01, "frame %d: %.2f%% of pixels took another path" % (k, 100 * bad.mean())
        assert np.array_equal(rgba[..., 3], np.ones((H, W), np.float32))
    print("pixels on a different path over 3 frames:", bad_total, "of", 3 * W * H)
    pt.close()


def test_accumulation_matches_the_shader_recurrence_and_sample_split_sums():
    W, H, N = 96, 54, 24
    pt = P.PathTracer()
    rgba, lin = pt.render(W, H, N)
    ref_rgba, ref_lin = O.pt_render(W, H, N, threads=8)
    ok = np.isfinite(ref_rgba).all(axis=2) & np.isfinite(rgba).all(axis=2)
    assert ok.mean() > 0.99
    assert np.array_equal(rgba[..., 3][ok], np.full(ok.sum(), N, np.float32))
    # converged images agree statistically (a handful of divergent paths in N*W*H)
    assert abs(float(rgba[..., :3][ok].mean()) - float(ref_rgba[..., :3][ok].mean())) < 2e-3
    assert np.mean(np.abs(rgba[..., :3][ok] - ref_rgba[..., :3][ok])) < 5e-3
    # splitting the samples over 4 "ranks" (first_frame = rank, stride = world) sums to the same linear image
    parts = [pt.render(W, H, N // 4, first_frame=r, frame_stride=4)[1] for r in range(4)]
    total = np.sum(parts, axis=0)
    assert np.allclose(total[ok], lin[ok], rtol=1e-4, atol=1e-4)
    # and the gamma-encoded running mean is the mean of the linear samples
    mean_gamma = np.power(np.clip(lin[ok] / N, 0, None), 1 / 2.2)
    assert np.mean(np.abs(mean_gamma - rgba[..., :3][ok])) < 2e-3
    pt.close()


def test_full_size_properties_1920x1080():
    """BASELINE config 5's frame size (1920x1080), 16 frames: size-independent properties of the accumulation --
    alpha == frame count in every pixel, finite colours where the samples are finite, the 8-way sample split
    (first_frame = rank, stride = world: what 8 GPUs trace) sums to the one-GPU linear image, a strip of the
    image equals the same strip of the CPU restatement on the overwhelming majority of pixels."""
    W, H, N = 1920, 1080, 16
    pt = P.PathTracer()
    rgba, lin = pt.render(W, H, N)
    fin = np.isfinite(rgba).all(axis=2)
    assert fin.mean() > 0.995
    assert np.array_equal(rgba[..., 3][fin], np.full(int(fin.sum()), N, np.float32))
    assert rgba[..., :3][fin].min() >= 0.0          # (the shader does not clamp bright samples: values above 1 occur)
    total = np.zeros_like(lin)
    for r in range(8):
        total += pt.render(W, H, N // 8, first_frame=r, frame_stride=8)[1]
    ok = np.isfinite(lin).all(axis=2) & np.isfinite(total).all(axis=2)
    assert np.allclose(total[ok], lin[ok], rtol=1e-4, atol=1e-4)
    # one frame of a 1920x24 strip against the oracle (same pixel coordinates: the strip is rows 528..551)
    one, lin1 = pt.render(W, H, 1)
    ref_rgba, ref_lin = O.pt_render(W, H, 1, threads=8)
    rows = slice(528, 552)
    d = np.abs(lin1[rows] - ref_lin[rows]).max(axis=2)
    okr = np.isfinite(d)
    assert np.median(d[okr]) < 1e-5 and (d[okr] > 1e-3).mean() < 0.01
    pt.close()


def test_config5_full_size_256_samples():
    """BASELINE config 5 AS QUOTED: 1920x1080, 256 samples per pixel in one call (0.2 s of GPU).  Parity is unpinned
    for this path (no GLSL compiler in the image), so the checks are the accumulation's size-independent properties:
    alpha == 256 wherever the colour is finite, no negative colours, the 8-way sample split of an 8-GPU run sums to
    the same linear image, and a second call returns the same bits (counter-free integer hash: deterministic)."""
    W, H, N = 1920, 1080, 256
    pt = P.PathTracer()
    rgba, lin = pt.render(W, H, N)
    fin = np.isfinite(rgba).all(axis=2)
    assert fin.mean() > 0.995
    assert np.array_equal(rgba[..., 3][fin], np.full(int(fin.sum()), N, np.float32))
    assert rgba[..., :3][fin].min() >= 0.0
    total = np.zeros_like(lin)
    for r in range(8):
        total += pt.render(W, H, N // 8, first_frame=r, frame_stride=8)[1]
    ok = np.isfinite(lin).all(axis=2) & np.isfinite(total).all(axis=2)
    assert np.allclose(total[ok], lin[ok], rtol=2e-4, atol=2e-3)        # 256 float additions in another order
    again = pt.render(W, H, N)[0]
    assert np.array_equal(again.view(np.uint32), rgba.view(np.uint32))
    pt.close()


def test_errors():
    pt = P.PathTracer()
    with pytest.raises(P.P3DError):
        pt.render(0, 10, 1)
    with pytest.raises(P.P3DError):
        pt.render(16, 16, 1, frame_stride=0)
    pt.close()
