"""The device's powf restatement (csrc/p3d_powf.h) compiled FOR THE HOST and compared with the host's libm.

The header is plain C++ apart from the bit-cast helpers, so g++ -mfma runs exactly the expressions the GPU runs (double
arithmetic, the same explicit fused multiply-adds, -ffp-contract=off).  Bit-equal to powf() here means the tables, the
polynomial order and the special cases are the ones of this image's glibc (2.35, __powf_fma); tests/test_gpu_powf.py
then shows the GPU evaluates them to the same bits.  Needs a host with FMA (the ifunc variant the port restates).
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO

CS = os.path.join(REPO, "u_4a_2s_p3d_raytracer_template2_amd", "csrc")

SRC = r"""
#define P3D_POWF_HOST_CHECK
#include "p3d_powf.h"
#include <math.h>
extern "C" void both(const float* x, const float* y, long n, float* port, float* libm) {
    for (long i = 0; i < n; ++i) { port[i] = p3d::p3d_powf(x[i], y[i]); libm[i] = powf(x[i], y[i]); }
}
"""


def powf_cases(rng, n):
    """(tag, x, y) argument sets: the shading range first, then everything a float pair can be."""
    bits = lambda: rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    sp = np.array([0, -0.0, 1, -1, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e-40, 2, 0.5, -2, 3, -3, 1e38, -1e38, 1.1754944e-38,
                   0.99999994, 1.0000001, 2.5, -2.5, 1e10, -1e10], np.float32)
    X, Y = np.meshgrid(sp, sp)
    one, zero = 0x3f800000, 0
    snan = [0x7f800001, 0xff800001, 0x7fa00000]          # kept as bit patterns: a float conversion would quiet them
    as_f32 = lambda words: np.array(words, np.uint32).view(np.float32)
    return [
        ("shading range: x in [0,1], the shine values of the scenes", rng.random(n), rng.choice([1, 2, 5, 10, 20, 30, 50, 100, 200, 1000, 0.5, 3.7, 100000], n)),
        ("x in [0,1.0001], any shine up to 300", rng.random(n) * 1.0001, rng.random(n) * 300),
        ("x one or two ulps around 1", np.nextafter(np.float32(1), np.float32(rng.choice([0, 2], n))).astype(np.float32), rng.random(n) * 1e6),
        ("random bit patterns", bits(), bits()),
        ("random x, integer y (sign rules of negative bases)", bits(), rng.integers(-40, 40, n)),
        ("special values, all pairs", X.ravel(), Y.ravel()),
        ("signalling NaNs", as_f32(snan + [one] * 3), as_f32([zero] * 3 + snan)),
        ("2^y across the overflow and underflow thresholds", np.full(n, 2), rng.random(n) * 40 - 20 + np.where(rng.random(n) < .5, 128, -150)),
        ("subnormal x", rng.random(n) * 1e-38, rng.random(n) * 4),
    ]


def differing(a, b):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    return (a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))     # NaN payloads are not compared


@pytest.fixture(scope="module")
def host_lib(tmp_path_factory):
    if "fma" not in open("/proc/cpuinfo").read():
        pytest.skip("host without FMA: glibc selects a different powf variant")
    d = tmp_path_factory.mktemp("powf")
    src = d / "h.cpp"
    src.write_text(SRC)
    so = d / "h.so"
    subprocess.check_call(["g++", "-O2", "-mfma", "-ffp-contract=off", "-shared", "-fPIC", "-I" + CS, str(src), "-o", str(so), "-lm"])
    L = C.CDLL(str(so))
    L.both.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
    return L


def test_powf_restatement_is_libm_bit_for_bit_on_the_host(host_lib):
    rng = np.random.default_rng(2024)
    total = 0
    for tag, x, y in powf_cases(rng, 1_000_000):
        x = np.ascontiguousarray(x, np.float32); y = np.ascontiguousarray(y, np.float32)
        port = np.zeros_like(x); libm = np.zeros_like(x)
        host_lib.both(x.ctypes.data, y.ctypes.data, len(x), port.ctypes.data, libm.ctypes.data)
        bad = differing(port, libm)
        assert not bad.any(), "%s: %d differ, first: x=%r y=%r port=%r libm=%r" % (
            tag, int(bad.sum()), x[bad][0], y[bad][0], port[bad][0], libm[bad][0])
        total += len(x)
    assert total > 7_000_000
