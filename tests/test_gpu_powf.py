"""The Blinn-Phong exponent on the GPU equals the host C library's powf bit for bit (csrc/p3d_powf.h).

The reference's frame holds powf(max(0, H.N), shine) unrounded (RT/main.cpp:520), so this is the one function whose
device implementation decides between "equal" and "close" for rgb32f.  Compared here: the device kernel behind
p3d_debug_powf against THIS box's libm, called from the small C harness of tests/test_powf_port.py (compiled on the box
with g++: calling powf through Python floats would quiet signalling NaNs on the way in).
"""
import numpy as np
import pytest

from test_powf_port import differing, host_lib, powf_cases      # host_lib: the C harness (libm called on float arrays in C)
import u_4a_2s_p3d_raytracer_template2_amd as P

pytestmark = pytest.mark.gpu


def test_device_powf_is_the_box_libm_bit_for_bit(host_lib):
    rng = np.random.default_rng(77)
    total = 0
    for tag, x, y in powf_cases(rng, 1_000_000):
        x = np.ascontiguousarray(x, np.float32); y = np.ascontiguousarray(y, np.float32)
        got = P.debug_powf(x, y)
        port = np.zeros_like(x); ref = np.zeros_like(x)
        host_lib.both(x.ctypes.data, y.ctypes.data, len(x), port.ctypes.data, ref.ctypes.data)
        bad = differing(got, ref)
        assert not bad.any(), "%s: %d of %d differ, first: x=%r y=%r device=%r libm=%r" % (
            tag, int(bad.sum()), len(x), x[bad][0], y[bad][0], got[bad][0], ref[bad][0])
        total += len(x)
    assert total > 7_000_000
