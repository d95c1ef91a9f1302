"""The oracle's GLIBC mode -- the restatement of glibc 2.35's binary32 sincosf / powf that the gfx950 engine executes
(oracle/amber_oracle.cc "GLIBC mode", amber_amd/csrc/hip/dev_math.h "sin / cos / pow") -- against the LIVE libm of this
host, which is what the reference calls (sampling.h:249-250, 279, 283-284).

libm is a third-party dependency of the reference (glibc 2.35, pinned by the image): its algorithm is restated, and
the restatement is proven here over the COMPLETE argument set of the path, not sampled: phi = (2 pi) * (k * 2^-24) for all
2^24 k (the sampler's uniforms are multiples of 2^-24), and pow(k * 2^-24, 1 / (e + 1)) for all 2^24 k for the Cornell
box's Phong exponent and a few others.  Random arguments cover the rest of both functions' domains.
glibc selects its sincosf / powf variant by CPU feature (ifunc): the restatement is the FMA variant, which every
FMA-capable x86-64 selects (this container, the GPU box's EPYC 9575F).  On a host without FMA the test is skipped.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as O


def _has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read().replace("\n", " ")
    except OSError:
        return False


pytestmark = pytest.mark.skipif(not _has_fma(), reason="host CPU has no FMA: glibc runs its SSE2 sincosf/powf variant here")


def test_sincos_restatement_equals_live_libm_on_every_phi_of_the_path(oracle):
    fb = (C.c_float * 2)()
    assert oracle.oracle_math_compare(0, O.MATH_LIBM, O.MATH_GLIBC, 0, 1 << 24, 0.0, fb) == 0, list(fb)


@pytest.mark.parametrize("exponent", [256.0, 1.0, 10.0, 64.0, 1000.0])     # cornel_box.cc:112 uses 256
def test_pow_restatement_equals_live_libm_on_every_uniform(oracle, exponent):
    y = np.float32(1.0) / (np.float32(exponent) + np.float32(1.0))         # sampling.h:279: 1 / (exponent + 1) in binary32
    fb = (C.c_float * 2)()
    assert oracle.oracle_math_compare(1, O.MATH_LIBM, O.MATH_GLIBC, 0, 1 << 24, float(y), fb) == 0, list(fb)


def test_restatement_equals_live_libm_on_random_arguments(oracle):
    fb = (C.c_float * 2)()
    assert oracle.oracle_math_compare(2, O.MATH_LIBM, O.MATH_GLIBC, 1, 20_000_001, 0.0, fb) == 0, list(fb)      # sincos, (-120, 120)
    for seed, yscale in ((7, 2.0), (9, 300.0), (11, 0.01)):                                                        # pow, all positive x
        assert oracle.oracle_math_compare(3, O.MATH_LIBM, O.MATH_GLIBC, seed, seed + 10_000_000, yscale, fb) == 0, list(fb)


def test_pow_special_cases(oracle):
    inf, nan = np.float32(np.inf), np.float32(np.nan)
    cases = [(0.0, 0.5), (0.0, -0.5), (-0.0, 3.0), (-0.0, -3.0), (1.0, nan), (nan, 0.0), (inf, 0.5), (inf, -0.5), (-inf, 3.0), (-inf, 2.0),
             (-2.0, 3.0), (-2.0, 2.0), (-2.0, 0.5), (0.5, inf), (2.0, inf), (0.5, -inf), (2.0, -inf), (-1.0, inf), (1e-45, 0.5),
             (1e-40, -0.25), (3.0, 200.0), (0.3, 200.0), (0.5, 149.5), (0.5, 150.0), (-0.5, 149.0), (2.0, 127.99999), (2.0, 128.0)]
    for x, y in cases:
        a = np.float32(oracle.oracle_pow(x, y, O.MATH_LIBM)); b = np.float32(oracle.oracle_pow(x, y, O.MATH_GLIBC))
        assert (np.isnan(a) and np.isnan(b)) or a.view(np.uint32) == b.view(np.uint32), (x, y, a, b)


def test_pow_int_forms(oracle):
    """x^4 / x^5 as std::pow(float, int) (double pow): the GLIBC mode's exact-product forms equal the live pow after the
    conversion to binary32 that every use on the path applies; in the double itself the last bit may differ (glibc's pow is
    not correctly rounded: ~1e-3 of the arguments)."""
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.random(20000), rng.random(2000) * 1e-4]).astype(np.float32)
    for n in (4, 5):
        a = np.array([oracle.oracle_pow_i(float(v), n, O.MATH_LIBM) for v in xs])
        b = np.array([oracle.oracle_pow_i(float(v), n, O.MATH_GLIBC) for v in xs])
        assert np.array_equal(a.astype(np.float32).view(np.uint32), b.astype(np.float32).view(np.uint32))
        assert (a != b).mean() < 5e-3


def test_portable_mode_distance_is_what_design_md_quotes(oracle):
    """Round 1's portable forms differ from glibc in the last bit for 29 % of the path's phi and 0.4 % of its Phong pow
    arguments -- the reason the engine no longer uses them."""
    fb = (C.c_float * 2)()
    n_sc = oracle.oracle_math_compare(0, O.MATH_LIBM, O.MATH_PORTABLE, 0, 1 << 24, 0.0, fb)
    assert 0.25 < n_sc / (1 << 24) < 0.33
    y = float(np.float32(1.0) / np.float32(257.0))
    n_pw = oracle.oracle_math_compare(1, O.MATH_LIBM, O.MATH_PORTABLE, 0, 1 << 24, y, fb)
    assert 0.001 < n_pw / (1 << 24) < 0.01
