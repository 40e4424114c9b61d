"""
The exact-FFT engine (csrc/blind_rotate_xfft.h) as compiled for the host (tests/emu): its external product is the exact
integer negacyclic convolution modulo 2^32 -- the output of the reference's NTT path (nufhe/tgsw_cpu.py:82-106) -- for
EVERY input, including the all-extreme ones the plain FFT path rounds wrongly, and the values handed to the rounding
stay inside the error bound of DESIGN.md section 7 (0.037).  CPU only; the GPU tests are in test_gpu_xfft.py.
"""
import numpy
import pytest

from tests.emu import emu


def _digits(t):
    """gadget digits of int32 torus coefficients, tgsw_cpu.py:41-47 (Bg = 2^10, l = 2, balanced)"""
    offset = numpy.uint32(2**31 + 2**21)
    x = (t.astype(numpy.uint32) + offset)
    d0 = ((x >> numpy.uint32(22)) & numpy.uint32(1023)).astype(numpy.int64) - 512
    d1 = ((x >> numpy.uint32(12)) & numpy.uint32(1023)).astype(numpy.int64) - 512
    return d0, d1


def _negacyclic(d, k):
    """exact negacyclic product of two integer vectors of length 1024 (python ints via int64 matrices: |.| < 2^63)"""
    n = 1024
    idx = (numpy.arange(n)[:, None] - numpy.arange(n)[None, :])          # i - j
    sign = numpy.where(idx < 0, -1, 1).astype(numpy.int64)
    M = k.astype(numpy.int64)[idx % n] * sign                            # M[i, j] = +-k[(i - j) mod n]
    return M @ d.astype(numpy.int64)


def _exact_external_product(T, tgsw_row):
    """sum_{m,d} digit_d(T_m) (*) row[m][d][mo]  mod 2^32;  T [2,1024] int32, tgsw_row [2,2,2,1024] int32"""
    res = numpy.zeros((2, 1024), numpy.int64)
    for m in range(2):
        ds = _digits(T[m])
        for d in range(2):
            for mo in range(2):
                res[mo] += _negacyclic(ds[d], tgsw_row[m, d, mo])
    return (res & 0xFFFFFFFF).astype(numpy.uint32).view(numpy.int32)


def test_digit_helper_matches_oracle(orc):
    rs = numpy.random.RandomState(1)
    t = rs.randint(-2**31, 2**31, size=(2, 1024), dtype=numpy.int32)
    dec = orc.tgsw_decomp(t)                       # [k+1, l, N]
    for m in range(2):
        d0, d1 = _digits(t[m])
        assert (dec[m, 0] == d0).all() and (dec[m, 1] == d1).all()


def test_exact_convolution_is_what_the_ntt_path_computes(orc):
    """the claim the engine rests on: tgsw_cpu.py:82-106 with the NTT references == exact integer negacyclic
    convolution mod 2^32 (here on extreme inputs, where the integer sums reach 2^52)"""
    rs = numpy.random.RandomState(5)
    T = rs.randint(-2**31, 2**31, size=(2, 1024), dtype=numpy.int32)
    row = numpy.where(rs.rand(2, 2, 2, 1024) < 0.5, numpy.int32(2**31 - 1), numpy.int32(-2**31)).astype(numpy.int32)
    ntt = orc.tgsw_external_mul(T[None], orc.tlwe_transform_samples(row[None]), 0)[0]
    assert (ntt == _exact_external_product(T, row)).all()


def test_external_product_random_inputs_equals_exact_convolution(orc):
    rs = numpy.random.RandomState(7)
    for trial in range(3):
        T = rs.randint(-2**31, 2**31, size=(2, 1024), dtype=numpy.int32)
        row = rs.randint(-2**31, 2**31, size=(2, 2, 2, 1024), dtype=numpy.int32)
        emu.xfft_margin()
        got = emu.xfft_external_product(T, emu.bkx_from_coeffs(row))
        assert (got == _exact_external_product(T, row)).all()
        assert (got == orc.tgsw_external_mul(T[None], orc.tlwe_transform_samples(row[None]), 0)[0]).all()
        frac, mag = emu.xfft_margin()
        assert frac < 0.004 and mag < 2.0**36, (frac, mag)


@pytest.mark.parametrize("pattern", ["all_max", "alternating", "random_signs", "halves_at_their_limits"])
def test_external_product_adversarial_extremes(pattern):
    """every digit -512 / +511 and every key coefficient at +-2^31 (or with both 16-bit halves at their limits), signs
    aligned so that the convolution sums do not cancel: the magnitudes the 0.037 bound is computed for.  The plain FFT
    path is 2 LSB off on such inputs (tests/test_gpu_fft.py); this engine must be exact."""
    rs = numpy.random.RandomState(11)
    # digits: d0 = d1 = -512 <=> t + offset has both fields 0; +511 <=> both fields 1023
    def t_with_fields(f0, f1):
        x = (numpy.uint32(f0) << numpy.uint32(22)) | (numpy.uint32(f1) << numpy.uint32(12))
        return numpy.uint32((int(x) - (2**31 + 2**21)) % 2**32).view(numpy.int32)
    tmin = t_with_fields(0, 0); tmax = t_with_fields(1023, 1023)
    kmax = numpy.int32(2**31 - 1); kmin = numpy.int32(-2**31)
    n = 1024
    if pattern == "all_max":
        T = numpy.full((2, n), tmin, numpy.int32)                      # all digits -512
        row = numpy.full((2, 2, 2, n), kmin, numpy.int32)              # all coefficients -2^31: products +2^40 each
        # (coefficient 1023 sums 1024 products of one sign: |v| = 4 * 1024 * 512 * 2^15 = 2^36 in the hi half)
    elif pattern == "alternating":
        sgn = numpy.where(numpy.arange(n) % 2 == 0, 1, -1)
        T = numpy.where(sgn > 0, tmax, tmin).astype(numpy.int32)[None, :].repeat(2, 0)
        row = numpy.where(sgn > 0, kmax, kmin).astype(numpy.int32)[None, None, None, :].repeat(2, 0).repeat(2, 1).repeat(2, 2)
    elif pattern == "random_signs":
        T = numpy.where(rs.rand(2, n) < 0.5, tmax, tmin).astype(numpy.int32)
        row = numpy.where(rs.rand(2, 2, 2, n) < 0.5, kmax, kmin).astype(numpy.int32)
    else:
        # lo = -2^15 and hi = +2^15 (k = 2^31 - 2^15 does not fit: use hi = 2^15 - 1 -> k = 0x7FFF8000) / lo = 2^15 - 1, hi = -2^15
        ka = numpy.int32(0x7FFF8000); kb = numpy.uint32(0x80007FFF).view(numpy.int32)
        T = numpy.where(rs.rand(2, n) < 0.5, tmax, tmin).astype(numpy.int32)
        row = numpy.where(rs.rand(2, 2, 2, n) < 0.5, ka, kb).astype(numpy.int32)
    emu.xfft_margin()
    got = emu.xfft_external_product(T, emu.bkx_from_coeffs(row))
    exact = _exact_external_product(T, row)
    assert (got == exact).all(), int((got != exact).sum())
    frac, mag = emu.xfft_margin()
    print("pattern %s: max |v - round(v)| = %.2e, max |v| = 2^%.2f" % (pattern, frac, numpy.log2(max(mag, 1.0))))
    assert frac < 0.037 and mag <= 2.0**36


def test_worst_case_alignment_reaches_the_magnitude_bound():
    """coefficient 1023 of d (*) k sums 1024 products with the same sign when d and k are constant: with both at their
    extremes and the four (m, d) terms aligned the half sums reach 4 * 1024 * 512 * 2^15 = 2^36 exactly"""
    n = 1024
    def t_with_fields(f0, f1):
        x = (numpy.uint32(f0) << numpy.uint32(22)) | (numpy.uint32(f1) << numpy.uint32(12))
        return numpy.uint32((int(x) - (2**31 + 2**21)) % 2**32).view(numpy.int32)
    T = numpy.full((2, n), t_with_fields(0, 0), numpy.int32)           # every digit -512
    row = numpy.full((2, 2, 2, n), numpy.int32(-2**31), numpy.int32)   # hi = -2^15, lo = 0
    emu.xfft_margin()
    got = emu.xfft_external_product(T, emu.bkx_from_coeffs(row))
    assert (got == _exact_external_product(T, row)).all()
    frac, mag = emu.xfft_margin()
    assert abs(mag - 2.0**36) < 0.037 and frac < 0.037, (frac, mag)


def test_blind_rotation_reduced_key_equals_ntt_oracle(orc):
    """the whole fused body (prologue, rotation, parked accumulator, extract) on a few rows of a full-range key ==
    the NTT oracle, word for word"""
    rs = numpy.random.RandomState(63)
    n = 6
    tgsw = rs.randint(-2**31, 2**31, size=(n, 2, 2, 2, 1024), dtype=numpy.int32)
    bk_ntt = orc.tlwe_transform_samples(tgsw)
    bkx = emu.bkx_from_coeffs(tgsw)
    MU = 2**29
    a0 = rs.randint(-2**31, 2**31, size=n, dtype=numpy.int32)
    a1 = rs.randint(-2**31, 2**31, size=n, dtype=numpy.int32)
    a0[3] = 0; a1[3] = 0                                           # bara = 0 -> skipped iteration
    b0 = rs.randint(-2**31, 2**31, size=1, dtype=numpy.int32)
    b1 = rs.randint(-2**31, 2**31, size=1, dtype=numpy.int32)
    ta = (a0 + a1).astype(numpy.int32); tb = (numpy.int32(-MU) + b0 + b1).astype(numpy.int32)
    ea, eb = orc.bootstrap_extract(bk_ntt, ta[None, :], tb, MU)
    ga, gb = emu.bootstrap_bit_xfft(bkx, n, (a0, b0), 1, (a1, b1), 1, -MU, MU)
    assert (ga == ea[0]).all() and gb == eb[0]


def _exact_external_product_k2(T, tgsw_row):
    """tlwe_mask_size = 2: T [3,1024] int32, tgsw_row [3,2,3,1024] int32"""
    res = numpy.zeros((3, 1024), numpy.int64)
    for m in range(3):
        ds = _digits(T[m])
        for d in range(2):
            for mo in range(3):
                res[mo] += _negacyclic(ds[d], tgsw_row[m, d, mo])
    return (res & 0xFFFFFFFF).astype(numpy.uint32).view(numpy.int32)


def test_mask_size_2_external_product_random_and_extreme(orc):
    """six digit polynomials per sum (bound 0.055): random full-range inputs and everything at its extreme"""
    rs = numpy.random.RandomState(9)
    tmin = numpy.uint32((0 - (2**31 + 2**21)) % 2**32).view(numpy.int32)
    tmax = numpy.uint32(((1023 << 22) | (1023 << 12)) - (2**31 + 2**21)).view(numpy.int32)
    cases = [(rs.randint(-2**31, 2**31, size=(3, 1024), dtype=numpy.int32), rs.randint(-2**31, 2**31, size=(3, 2, 3, 1024), dtype=numpy.int32)),
             (numpy.where(rs.rand(3, 1024) < 0.5, tmax, tmin).astype(numpy.int32),
              numpy.where(rs.rand(3, 2, 3, 1024) < 0.5, numpy.int32(2**31 - 1), numpy.int32(-2**31)).astype(numpy.int32)),
             (numpy.full((3, 1024), tmin, numpy.int32), numpy.full((3, 2, 3, 1024), numpy.int32(-2**31), numpy.int32))]
    for T, row in cases:
        emu.xfft_margin()
        got = emu.xfft_external_product_k2(T, emu.bkx_from_coeffs(row))
        assert (got == _exact_external_product_k2(T, row)).all()
        frac, mag = emu.xfft_margin()
        assert frac < 0.055 and mag <= 1.5 * 2.0**36 + 1, (frac, mag)
    T, row = cases[0]
    ntt = orc.tgsw_external_mul(T[None], orc.tlwe_transform_samples(row[None]), 0)[0]
    assert (ntt == _exact_external_product_k2(T, row)).all()


def test_mask_size_2_blind_rotation_reduced_key_equals_ntt_oracle(orc):
    rs = numpy.random.RandomState(64)
    n = 5
    tgsw = rs.randint(-2**31, 2**31, size=(n, 3, 2, 3, 1024), dtype=numpy.int32)
    bk_ntt = orc.tlwe_transform_samples(tgsw)
    bkx = emu.bkx_from_coeffs(tgsw)
    MU = 2**29
    a0 = rs.randint(-2**31, 2**31, size=n, dtype=numpy.int32); a1 = rs.randint(-2**31, 2**31, size=n, dtype=numpy.int32)
    a0[2] = 0; a1[2] = 0
    b0 = rs.randint(-2**31, 2**31, size=1, dtype=numpy.int32); b1 = rs.randint(-2**31, 2**31, size=1, dtype=numpy.int32)
    ta = (a0 + a1).astype(numpy.int32); tb = (numpy.int32(-MU) + b0 + b1).astype(numpy.int32)
    ea, eb = orc.bootstrap_extract(bk_ntt, ta[None, :], tb, MU)
    ga, gb = emu.bootstrap_bit_xfft_k2(bkx, n, (a0, b0), 1, (a1, b1), 1, -MU, MU)
    assert (ga == ea[0]).all() and gb == eb[0]
    # six waves per bit (brxq_* with K = 2): the same words
    ha, hb = emu.bootstrap_bit_xfft_hex_k2(bkx, n, (a0, b0), 1, (a1, b1), 1, -MU, MU)
    assert (ha == ea[0]).all() and hb == eb[0]


def test_quad_kernel_body_equals_ntt_oracle_and_the_one_wave_body(orc):
    """four waves per bit (brxq_*): forward side by (polynomial, digit), product side by (output, key half), ACC through
    LDS atomics -- the same words as the oracle's prime-field path and as the one-wave body, full-range key and inputs"""
    rs = numpy.random.RandomState(65)
    n = 6
    tgsw = rs.randint(-2**31, 2**31, size=(n, 2, 2, 2, 1024), dtype=numpy.int32)
    bk_ntt = orc.tlwe_transform_samples(tgsw)
    bkx = emu.bkx_from_coeffs(tgsw)
    MU = 2**29
    a0 = rs.randint(-2**31, 2**31, size=n, dtype=numpy.int32); a1 = rs.randint(-2**31, 2**31, size=n, dtype=numpy.int32)
    a0[3] = 0; a1[3] = 0        # one step with X^0 - 1 = 0: skipped by all four waves
    b0 = rs.randint(-2**31, 2**31, size=1, dtype=numpy.int32); b1 = rs.randint(-2**31, 2**31, size=1, dtype=numpy.int32)
    ta = (a0 + a1).astype(numpy.int32); tb = (numpy.int32(-MU) + b0 + b1).astype(numpy.int32)
    ea, eb = orc.bootstrap_extract(bk_ntt, ta[None, :], tb, MU)
    for split in (False, True):      # (True: the one-team build, a second exchange buffer instead of barrier 2)
        ga, gb = emu.bootstrap_bit_xfft_quad(bkx, n, (a0, b0), 1, (a1, b1), 1, -MU, MU, split=split)
        assert (ga == ea[0]).all() and gb == eb[0]
    wa, wb = emu.bootstrap_bit_xfft(bkx, n, (a0, b0), 1, (a1, b1), 1, -MU, MU)
    assert (ga == wa).all() and gb == wb
