"""Exhaustive checks of device arithmetic that replaces a correctly rounded operation by a shorter sequence."""
import pytest

import u_4a_2s_p3d_raytracer_template2_amd as P

pytestmark = pytest.mark.gpu


def test_reciprocal_equals_the_division_for_every_float():
    """frcp(x) (hardware reciprocal + one Newton step in FMA arithmetic; the division itself outside the range where x and
    1 / x are normal) against 1.0f / x, all 2^32 bit patterns: normalize() and Triangle::intercepts divide 1.0 by a float."""
    n_bad, first_bad = P.debug_check_rcp(0, 1 << 32)
    assert n_bad == 0, "%d bit patterns differ, the lowest 0x%08x" % (n_bad, first_bad)
