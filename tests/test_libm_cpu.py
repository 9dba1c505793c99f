"""The two places where bit parity leans on the C library of the machine:

  * `hypot(g, 0.01)` of the local TV-CSAD / NLTV-CSAD functionals (src/tvcsad_model.cpp:337, src/nltvcsad_model.cpp:357) is
    evaluated per pixel per window ON THE DEVICE as the correctly rounded double root of the exact sum, rounded once to float
    (csrc/grow_nl_kernels.hpp, hypot_g_001); the CPU checker and the reference call libm's hypot.  glibc does not promise
    correct rounding, so this test says by name when a C library disagrees with the correctly rounded value after the
    rounding to float -- instead of a flow mismatch in a growing test;
  * the golden vectors of tests/golden (Lab image, non-local weights, window weights: pow / exp / expf / hypot on the host)
    were produced with the glibc recorded in known_answers.json; tests/test_oracle_models_golden_cpu.py compares them on
    every machine, this file only reports the version next to it.
"""
import math
import os
from fractions import Fraction

import numpy as np

from util import known_answers


def correctly_rounded_hypot_g_001(g):
    """float32( RN_double( sqrt(g^2 + y^2) ) ), y = the double nearest 0.01, by exact integer arithmetic."""
    x, y = Fraction(float(g)), Fraction(0.01)
    s = x * x + y * y
    k = 60 - math.floor(math.log2(float(s))) // 2                 # sqrt(s) * 2^k has ~61 bits
    n = s * Fraction(4) ** k if k >= 0 else s / Fraction(4) ** (-k)
    r = math.isqrt(n.numerator // n.denominator)
    v = Fraction(r) + (Fraction(1, 4) if Fraction(r * r) != n else 0)   # sticky quarter: never on a tie of the 53-bit rounding
    v = v / Fraction(2) ** k if k >= 0 else v * Fraction(2) ** (-k)
    return np.float32(float(v))                                   # Fraction -> float rounds correctly; then one rounding to float


def test_libm_hypot_rounds_to_float_like_the_correctly_rounded_root():
    rng = np.random.default_rng(20261004)
    gs = np.concatenate([
        rng.random(15000).astype(np.float32) * np.float32(0.05),          # |grad I|^2 of smooth image regions: around 0.01
        np.abs(rng.standard_normal(15000)).astype(np.float32),
        (10.0 ** rng.uniform(-30, 30, 15000)).astype(np.float32),         # every exponent the float range has
        np.array([0.0, 0.01, 1e-45, 1e-20, 1.0, 3e38], np.float32)])
    bad = [(float(g), float(np.float32(math.hypot(float(g), 0.01))), float(correctly_rounded_hypot_g_001(g)))
           for g in gs if np.float32(math.hypot(float(g), 0.01)).view(np.uint32) != correctly_rounded_hypot_g_001(g).view(np.uint32)]
    assert not bad, ("this C library's hypot(g, 0.01) differs from the correctly rounded root after rounding to float for %d of %d "
                     "arguments, e.g. %s: the device's -m 4..6 window solvers (hypot_g_001) will differ from this machine's CPU path"
                     % (len(bad), len(gs), bad[:3]))


def test_glibc_of_the_golden_vectors_is_recorded():
    rec = known_answers()["_libm_of_golden_vectors"]
    assert rec["glibc"] == "2.35"
    here = os.confstr("CS_GNU_LIBC_VERSION") if hasattr(os, "confstr") else None
    if here and rec["glibc"] not in here:
        print("note: tests/golden vectors were produced with glibc %s, this machine has %s" % (rec["glibc"], here))
