"""The arithmetic claim behind the TV-L1 kernels' one-product quotients (faldoi-ipol_amd/csrc/faldoi_device.hpp: div_once), checked
on the CPU with numpy's IEEE arithmetic, independently of any device:

    RN32(a / d) == RN32( RN53( a * RN53(1/d) ) )      for every float a and every float d > 0 that is not an even integer >= 6

(the device's reciprocal is within 2^-52 of 1/d, like the correctly rounded one used here; the GPU self-test covers the device's own
instructions, faldoi_selftest_division).  Exact ties -- a = d * (2k+1) * 2^-150 representable, which needs an even-integer d -- are
the only failures, and powers of two are safe among those."""
import numpy as np


def _once(a, d):
    a = a.astype(np.float32)
    rd = np.float64(1.0) / np.float64(np.float32(d))
    with np.errstate(over="ignore", under="ignore", invalid="ignore"):
        return (a.astype(np.float64) * rd).astype(np.float32)


def _ieee(a, d):
    with np.errstate(over="ignore", under="ignore", invalid="ignore", divide="ignore"):
        return (a.astype(np.float32) / np.float32(d)).astype(np.float32)


def _numerators(rng, n):
    bits = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    bits = np.where((bits & 0x7f800000) == 0x7f800000, bits & 0xbfffffff, bits)           # finite
    spec = np.concatenate([bits[: n // 8] & np.uint32(0x807fffff),                          # subnormals
                           (bits[: n // 8] & np.uint32(0x80ffffff)) | np.uint32(0x00800000),   # smallest normals
                           np.array([0, 0x80000000, 1, 0x80000001, 0x7f7fffff, 0xff7fffff, 0x00800000], np.uint32)])
    return np.concatenate([bits, spec]).view(np.float32)


def _ties(d, ks):
    """Floats nearest to d * (2k+1) * 2^-150: exact ties of the quotient whenever that product is representable."""
    with np.errstate(under="ignore"):
        t = (np.float64(np.float32(d)) * (2.0 * ks + 1.0) * 2.0 ** -150).astype(np.float32)
    return np.concatenate([t, -t])


def test_one_product_quotient_rounds_like_the_ieee_divide():
    rng = np.random.default_rng(5)
    ks = np.arange(0, 4000, dtype=np.float64)
    divisors = [0.3, 0.1, 0.8, 0.25, 0.5, 1.0, 2.0, 4.0, 3.0, 5.0, 7.0, 1.5, 2.5, 1e-8, 3.4e-5, 0.0123, 117.25, 6.5, 6.000001]
    divisors += [float(np.float32(x)) for x in np.exp(rng.uniform(np.log(1e-9), np.log(1e6), 60))]   # (floats >= 2^24 are all even integers)
    divisors += [float(2 * k + 1) for k in rng.integers(1, 5000, 20)]                      # odd integers: safe
    a = _numerators(rng, 1 << 20)
    for d in divisors:
        d32 = np.float32(d)
        assert not (d32 >= 6 and (d32 / 2) % 1 == 0), d                                    # none of these is refused by div_once_ok
        for x in (a, _ties(d, ks)):
            got, want = _once(x, d32), _ieee(x, d32)
            bad = got.view(np.uint32) != want.view(np.uint32)
            assert not bad.any(), (d, x[bad][:4], got[bad][:4], want[bad][:4])


def test_even_integer_divisors_are_where_it_fails_and_powers_of_two_are_not():
    ks = np.arange(0, 4000, dtype=np.float64)
    failing = 0
    # (with a correctly rounded reciprocal the product often rounds back onto the tie: 6, 10, 12 ... pass here; 98, 150, 154 do not --
    # and the device's reciprocal may be one fp64 ulp off, which exposes more of them: the GPU self-test counts those)
    for d in (6.0, 10.0, 12.0, 24.0, 98.0, 150.0, 154.0, 196.0, 4094.0, 3.0 * 2 ** 20):
        x = _ties(d, ks)
        bad = _once(x, np.float32(d)).view(np.uint32) != _ieee(x, np.float32(d)).view(np.uint32)
        failing += int(bad.sum())
        # ... and only on results in the subnormal range
        assert np.all(np.abs(_ieee(x, np.float32(d))[bad]) < np.float32(2.0 ** -126))
    assert failing > 0                                                                    # the refusal of div_once_ok is needed
    rng = np.random.default_rng(6)
    a = _numerators(rng, 1 << 18)
    for d in (8.0, 16.0, 64.0, 2.0 ** 30, 2.0 ** 100):                                    # refused (conservatively), yet exact
        for x in (a, _ties(d, ks)):
            assert np.array_equal(_once(x, np.float32(d)).view(np.uint32), _ieee(x, np.float32(d)).view(np.uint32)), d
