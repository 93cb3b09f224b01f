"""The arithmetic behind the split-bf16 GEMMs (`split_rne4` in mineral_amd/csrc/gemm.hip.h), restated with numpy: an fp32 value
splits exactly into three bf16 terms, each rounded to nearest even, and the six products the kernels accumulate reproduce an
fp32 product to within fp32 rounding, without bias.  CPU only: this pins the claim the kernels rest on, not the kernels (tests/test_shac_gpu.py does that)."""
import numpy as np


def rne_bf16(x):
    """fp32 -> the fp32 value of the nearest bf16 (ties to even), what v_cvt_pk_bf16_f32 produces for finite inputs."""
    b = x.view(np.uint32).astype(np.uint64)
    b = (b + 0x7FFF + ((b >> 16) & 1)) & 0xFFFF0000
    return b.astype(np.uint32).view(np.float32)


def split3(x):
    h = rne_bf16(x)
    r1 = (x - h).astype(np.float32)
    m = rne_bf16(r1)
    l = (r1 - m).astype(np.float32)
    return h, m, l


def samples(n, seed):
    rng = np.random.default_rng(seed)
    mant = rng.uniform(1.0, 2.0, n)
    expo = rng.integers(-60, 60, n)              # magnitudes from 1e-18 to 1e18 (products stay inside fp32): bf16 keeps fp32's exponent range
    sign = rng.choice([-1.0, 1.0], n)
    return (sign * mant * np.exp2(expo)).astype(np.float32)


def test_three_term_split_is_exact():
    x = samples(200000, 1)
    h, m, l = split3(x)
    assert np.array_equal((l.view(np.uint32) & np.uint32(0xFFFF)), np.zeros_like(l.view(np.uint32))), 'the last term fits bf16'
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64), x.astype(np.float64))
    # each term is at most half a bf16 ulp of the one before it, with either sign
    assert np.all(np.abs(m) <= np.abs(x) * 2.0 ** -8) and np.all(np.abs(l) <= np.abs(x) * 2.0 ** -16)
    assert 0.4 < np.mean(m[m != 0] * np.sign(x[m != 0]) > 0) < 0.6, 'the remainder is as often above as below'
    assert np.array_equal(rne_bf16(l), l) and np.array_equal(rne_bf16(m), m)


def test_six_products_match_an_fp32_product():
    a, b = samples(200000, 2), samples(200000, 3)
    ah, am, al = (t.astype(np.float64) for t in split3(a))
    bh, bm, bl = (t.astype(np.float64) for t in split3(b))
    six = al * bh + ah * bl + am * bm + am * bh + ah * bm + ah * bh      # the order of the kernels: smallest terms first
    exact = a.astype(np.float64) * b.astype(np.float64)
    rel = np.abs(six - exact) / np.abs(exact)
    assert rel.max() <= 2.0 ** -23, rel.max()      # the three dropped terms: at most 2 * 2^-8 * 2^-16 + 2^-32 of |a*b|
    assert abs(np.mean((six - exact) / exact)) < 2.0 ** -28, 'no bias towards zero'
    # against the fp32 product itself (one rounding, 2^-24): the same order of magnitude
    f32 = (a * b).astype(np.float64)
    assert rel.mean() <= 2.0 * (np.abs(f32 - exact) / np.abs(exact)).mean()
    # a bf16 x bf16 product has at most 16 significant bits: exact in fp32, as the matrix cores accumulate it
    p = (ah * bh)
    assert np.array_equal(p.astype(np.float32).astype(np.float64), p)
