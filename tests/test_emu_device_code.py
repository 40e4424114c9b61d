"""
CPU checks of the DEVICE code (nufhe_amd/csrc/ff.h, ntt1024.h, ...) compiled for the host and
run lane-by-lane through the fibre emulator in tests/emu/ -- compared with the oracle.  These
tests validate the arithmetic formulations and the index math of the HIP kernels without a GPU;
the GPU parity tests proper are in tests/test_gpu_*.py.
"""

import numpy
import pytest

from tests.emu import emu

P = 2**64 - 2**32 + 1


def _edge_values():
    return [0, 1, 2, 2**31, 2**32 - 1, 2**32, 2**32 + 1, 2**63, P - 1, P - 2, P - 2**32, P - 2**32 - 1,
            2**64 - 2**33, 0xFFFFFFFE00000000, 0xFFFFFFFEFFFFFFFF, 0x00000000FFFFFFFF, 0xFFFFFFFF00000000]


def _ff_inputs(n=4000, seed=11):
    rs = numpy.random.RandomState(seed)
    a = rs.randint(0, P, size=n, dtype=numpy.uint64)
    b = rs.randint(0, P, size=n, dtype=numpy.uint64)
    ev = _edge_values()
    k = 0
    for x in ev:
        for y in ev:
            a[k] = x; b[k] = y; k += 1
    return a, b


def test_ff_add_sub_mul(orc):
    a, b = _ff_inputs()
    assert (emu.ff_binary('emu_ff_add', a, b) == orc.ff_add(a, b)).all()
    assert (emu.ff_binary('emu_ff_sub', a, b) == orc.ff_sub(a, b)).all()
    assert (emu.ff_binary('emu_ff_mul', a, b) == orc.ff_mul(a, b)).all()


def test_ff_shifts(orc):
    a, _ = _ff_inputs(1000, seed=12)
    for s in range(192):
        exp = orc.ff_lsh(a, numpy.full(a.shape, s, numpy.uint32))
        assert (emu.ff_lsh_const(a, s) == exp).all(), s
    rs = numpy.random.RandomState(13)
    s = rs.randint(0, 32, size=a.size).astype(numpy.uint32)
    s[:4] = [0, 31, 1, 30]
    assert (emu.ff_lsh_var(a, s) == orc.ff_lsh(a, s)).all()


def test_ntt_forward_inverse(orc):
    rs = numpy.random.RandomState(14)
    for case in range(3):
        if case == 0:
            x = rs.randint(0, P, size=1024, dtype=numpy.uint64)
        elif case == 1:
            x = numpy.zeros(1024, numpy.uint64); x[1] = 1
        else:
            x = orc.ff_from_i32(rs.randint(-512, 512, size=1024).astype(numpy.int32))
        f = emu.ntt_forward(x)
        assert (f == orc.ntt_forward(x[None, :], i32_conversion=False)[0]).all()
        assert (emu.ntt_inverse(f) == x).all()
        y = rs.randint(0, P, size=1024, dtype=numpy.uint64)
        assert (emu.ntt_inverse(y) == orc.ntt_inverse(y[None, :], i32_conversion=False)[0]).all()


def test_bootstrap_wave_body_reduced(orc):
    """The fused blind-rotate body (device code, emulated) == oracle bootstrap_extract, on a
    random transformed-domain key with a few rows (incl. bara = 0 and bara >= N cases)."""
    rs = numpy.random.RandomState(21)
    n = 6
    bk = rs.randint(0, P, size=(n, 2, 2, 2, 1024), dtype=numpy.uint64)
    bki = emu.bk_from_reference(bk)
    MU = 2**29
    for trial in range(2):
        a0 = rs.randint(-2**31, 2**31, size=n, dtype=numpy.int32)
        a1 = rs.randint(-2**31, 2**31, size=n, dtype=numpy.int32)
        b0 = rs.randint(-2**31, 2**31, size=1, dtype=numpy.int32)
        b1 = rs.randint(-2**31, 2**31, size=1, dtype=numpy.int32)
        if trial == 1:
            a0[2] = 0; a1[2] = 0          # bara = 0 -> skipped iteration
        ta = (-a0 - a1).astype(numpy.int32)
        tb = (numpy.int32(MU) - b0 - b1).astype(numpy.int32)
        ea, eb = orc.bootstrap_extract(bk, ta[None, :], tb, MU)
        ga, gb = emu.bootstrap_bit(bki, n, (a0, b0), -1, (a1, b1), -1, MU, MU)
        assert (ga == ea[0]).all()
        assert gb == eb[0]
        # the 4-wave team variant of the same body (small-batch path): identical bits
        ga, gb = emu.bootstrap_bit(bki, n, (a0, b0), -1, (a1, b1), -1, MU, MU, team=True)
        assert (ga == ea[0]).all()
        assert gb == eb[0]
        # the 2-wave pair variant (medium batches): identical bits
        ga, gb = emu.bootstrap_bit(bki, n, (a0, b0), -1, (a1, b1), -1, MU, MU, pair=True)
        assert (ga == ea[0]).all()
        assert gb == eb[0]
        # the 8-wave half-ring team (smallest batches; key in the half-ring layout): identical bits
        ga, gb = emu.bootstrap_bit(emu.bk_to_half(bki), n, (a0, b0), -1, (a1, b1), -1, MU, MU, team8=True)
        assert (ga == ea[0]).all()
        assert gb == eb[0]


@pytest.mark.slow
def test_bootstrap_wave_body_full_key(orc, oracle_keys):
    """Full-size key (n = 500), one bit, NAND pre-combination: device code == oracle."""
    lwe_key, tlwe_key, ck = oracle_keys
    rng = orc.DeterministicRNG(456)
    m1 = numpy.array([True]); m2 = numpy.array([True])
    c1 = orc.encrypt(rng, lwe_key, m1); c2 = orc.encrypt(rng, lwe_key, m2)
    MU = 2**29
    ta = (-c1[0] - c2[0]).astype(numpy.int32); tb = (numpy.int32(MU) - c1[1] - c2[1]).astype(numpy.int32)
    ea, eb = orc.bootstrap_extract(ck.bk, ta, tb, MU)
    bki = emu.bk_from_reference(ck.bk)
    ga, gb = emu.bootstrap_bit(bki, 500, (c1[0][0], c1[1]), -1, (c2[0][0], c2[1]), -1, MU, MU)
    assert (ga == ea[0]).all() and gb == eb[0]


def test_fft_forward_inverse_vs_numpy():
    """Device FFT-512 (emulated) vs the numpy restatement of fft_transform_ref."""
    from oracle import oracle_fft as of
    rs = numpy.random.RandomState(41)
    for case in range(3):
        if case == 0:
            a = rs.randint(-2**31, 2**31, size=1024, dtype=numpy.int32)
        elif case == 1:
            a = numpy.zeros(1024, numpy.int32); a[1] = 1; a[700] = -3
        else:
            a = rs.randint(-512, 512, size=1024).astype(numpy.int32)
        f = emu.fft_forward(a)
        ref = of.fft_forward(a)
        scale = max(1.0, numpy.abs(ref).max())
        assert numpy.abs(f - ref).max() / scale < 1e-13
        assert (emu.fft_inverse(f) == a).all()
        assert (emu.fft_inverse(ref) == of.fft_inverse(ref)).all()
    # product of a full-range polynomial with a small one is exact (test_computation.py:71-124)
    a = rs.randint(-2**31, 2**31, size=1024, dtype=numpy.int32)
    b = rs.randint(-1000, 1000, size=1024).astype(numpy.int32)
    prod = emu.fft_inverse(emu.fft_forward(a) * emu.fft_forward(b))
    from oracle import oracle as orc
    assert (prod == orc.poly_mul_schoolbook(a[None, :], b[None, :])[0]).all()


@pytest.mark.slow
def test_half_ring_transforms_vs_oracle(orc):
    """ntt512_half.h (the small-batch kernel's transforms: X^1024 + 1 = (X^512 - i)(X^512 + i), two wavefronts per
    transform): the forward half transforms of a digit polynomial give the even / odd frequencies of the oracle's
    1024-point NTT, and the two inverse halves + the join give back the coefficients of a product-sized polynomial."""
    rs = numpy.random.RandomState(31)
    for trial in range(3):
        d = rs.randint(-512, 512, size=1024).astype(numpy.int32)
        if trial == 0:
            d[:] = -512                                    # extreme digits: the limb bounds of pass 1
        ref = orc.ntt_forward(d)                           # i32 -> field -> natural-order NTT
        got = emu.nth_forward_small(d)
        assert (got == ref).all(), (trial, int((got != ref).sum()))
    for trial in range(3):
        c = rs.randint(-2**51, 2**51, size=1024).astype(numpy.int64)       # |coefficients| < 2^52 like the products
        x = orc.ntt_forward(numpy.array([int(v) % orc.P for v in c], dtype=numpy.uint64), i32_conversion=False)
        got = emu.nth_inverse_i32(x)
        assert (got == (c & 0xFFFFFFFF).astype(numpy.uint32)).all(), trial


def test_fft_final_rounding_matches_reference_semantics_up_to_2_52():
    """fft_round_to_u32 (fft512.h) == round-half-even -> int64 -> low 32 bits (transform/fft.mako:272-277) on the whole
    range the blind rotation can reach, |v| <= 2^52: both signs, ties, the binade edges 2^51 and 2^52 where a signed
    magic-number add stops working."""
    rs = numpy.random.RandomState(7)
    vals = [0.0, -0.0, 0.5, -0.5, 1.5, -1.5, 2.5, -2.5, 2.0**31, -2.0**31, 2.0**32 - 0.5, -(2.0**32) + 0.5,
            2.0**51, -(2.0**51), 2.0**51 + 1, -(2.0**51) - 1, 2.0**52 - 1, -(2.0**52) + 1, 2.0**52, -(2.0**52),
            2.0**51 - 0.5, -(2.0**51) + 0.5]
    for e in range(1, 53):
        m = rs.randint(0, 2**31, size=200).astype(numpy.float64) * 2.0**21 + rs.randint(0, 2**21, size=200)   # 52 bits
        v = numpy.ldexp(m, e - 52)
        vals.extend(v.tolist()); vals.extend((-v).tolist())
        k = numpy.floor(numpy.ldexp(rs.rand(50), min(e, 50))) + 0.5                                           # ties
        vals.extend(k.tolist()); vals.extend((-k).tolist())
    top = 2.0**51 + numpy.floor(rs.rand(2000) * 2.0**51)              # the binade [2^51, 2^52): integers only
    vals.extend(top.tolist()); vals.extend((-top).tolist())
    v = numpy.array(vals, numpy.float64)
    v = v[numpy.abs(v) <= 2.0**52]
    assert (numpy.abs(v) >= 2.0**51).sum() > 500
    expect = (numpy.rint(v).astype(numpy.int64) & 0xFFFFFFFF).astype(numpy.uint32)
    got = emu.fft_round(v)
    assert (got == expect).all(), v[got != expect][:5]


def test_bootstrap_wave_body_fft_full_key(orc, oracle_keys):
    """FFT variant of the fused body on the full-size key, one bit: equals the EXACT (NTT) oracle --
    the fp64 rounding error stays below 0.5 LSB (FFT tolerance statement)."""
    from oracle import oracle_fft as of
    lwe_key, tlwe_key, ck = oracle_keys
    rng = orc.DeterministicRNG(456)
    c1 = orc.encrypt(rng, lwe_key, numpy.array([True])); c2 = orc.encrypt(rng, lwe_key, numpy.array([False]))
    MU = 2**29
    ta = (-c1[0] - c2[0]).astype(numpy.int32); tb = (numpy.int32(MU) - c1[1] - c2[1]).astype(numpy.int32)
    ea, eb = orc.bootstrap_extract(ck.bk, ta, tb, MU)
    bkf = of.bk_from_coeffs(of.tgsw_coeffs_from_reference_bk(ck.bk))
    emu.fft_margin()
    ga, gb = emu.bootstrap_bit_fft(emu.bkf_from_reference(bkf), 500, (c1[0][0], c1[1]), -1, (c2[0][0], c2[1]), -1, MU, MU)
    assert (ga == ea[0]).all() and gb == eb[0]
    # rounding margin of the whole blind rotation (1,024,000 rounded values): the fp64 error before
    # `round` must stay far from 0.5 and the magnitudes inside the 2^51 range of the magic-number round
    # (products of real keys are ~2^44..2^46, error ~ 2^-53 * 10 * |v|)
    max_frac, max_abs = emu.fft_margin()
    print("FFT blind rotation: max |v - round(v)| = %.4f, max |v| = 2^%.1f" % (max_frac, numpy.log2(max_abs)))
    assert max_frac < 0.125 and max_abs < 2.0**49
    # the 4-wave team variant (different fp64 summation order): still equal to the exact result
    ga, gb = emu.bootstrap_bit_fft(emu.bkf_from_reference(bkf), 500, (c1[0][0], c1[1]), -1, (c2[0][0], c2[1]), -1, MU, MU,
                                   team=True)
    assert (ga == ea[0]).all() and gb == eb[0]


def test_bootstrap_fft_bodies_reduced_key(orc):
    """The four FFT bodies for k = 1 (one wave per bit, the 4-wave team, the 2-wave pair, the 4-wave quad) on a reduced number of rows of
    a full-range int32 TGSW key == the EXACT (NTT) oracle on the same key (their fp64 sums associate differently; all of
    them round to the exact integers)."""
    from oracle import oracle_fft as of
    rs = numpy.random.RandomState(63)
    n = 6
    tgsw = rs.randint(-2**31, 2**31, size=(n, 2, 2, 2, 1024), dtype=numpy.int32)
    bk_ntt = orc.tlwe_transform_samples(tgsw)
    bkf = emu.bkf_from_reference(of.fft_forward(tgsw))
    MU = 2**29
    a0 = rs.randint(-2**31, 2**31, size=n, dtype=numpy.int32)
    a1 = rs.randint(-2**31, 2**31, size=n, dtype=numpy.int32)
    a0[3] = 0; a1[3] = 0                                           # bara = 0 -> skipped iteration
    b0 = rs.randint(-2**31, 2**31, size=1, dtype=numpy.int32)
    b1 = rs.randint(-2**31, 2**31, size=1, dtype=numpy.int32)
    ta = (a0 + a1).astype(numpy.int32); tb = (numpy.int32(-MU) + b0 + b1).astype(numpy.int32)
    ea, eb = orc.bootstrap_extract(bk_ntt, ta[None, :], tb, MU)
    for kw in ({}, {'team': True}, {'pair': True}, {'quad': True}):
        ga, gb = emu.bootstrap_bit_fft(bkf, n, (a0, b0), 1, (a1, b1), 1, -MU, MU, **kw)
        assert (ga == ea[0]).all() and gb == eb[0], kw


def test_bootstrap_wave_body_mask_size_2(orc):
    """tlwe_mask_size = 2 (test/test_gates.py:96-100 of the reference): the K = 2 instantiation of
    the fused body == oracle, reduced number of rows, random transformed-domain key."""
    rs = numpy.random.RandomState(61)
    n = 5
    bk = rs.randint(0, P, size=(n, 3, 2, 3, 1024), dtype=numpy.uint64)
    bki = emu.bk_from_reference(bk)
    MU = 2**29
    a0 = rs.randint(-2**31, 2**31, size=n, dtype=numpy.int32)
    a1 = rs.randint(-2**31, 2**31, size=n, dtype=numpy.int32)
    a0[2] = 0; a1[2] = 0                                           # bara = 0 -> a step every wave of a team must skip alike
    b0 = rs.randint(-2**31, 2**31, size=1, dtype=numpy.int32)
    b1 = rs.randint(-2**31, 2**31, size=1, dtype=numpy.int32)
    ta = (a0 + a1).astype(numpy.int32); tb = (numpy.int32(-MU) + b0 + b1).astype(numpy.int32)
    ea, eb = orc.bootstrap_extract(bk, ta[None, :], tb, MU)
    ga, gb = emu.bootstrap_bit(bki, n, (a0, b0), 1, (a1, b1), 1, -MU, MU, mask_size=2)
    assert ea.shape == (1, 2048)
    assert (ga == ea[0]).all() and gb == eb[0]
    # the 3-wave team variant (small batches): same bits
    ga, gb = emu.bootstrap_bit(bki, n, (a0, b0), 1, (a1, b1), 1, -MU, MU, mask_size=2, ring=True)
    assert (ga == ea[0]).all() and gb == eb[0]        # the 3-wave ring variant (no partial-sum buffer)
    ga, gb = emu.bootstrap_bit(bki, n, (a0, b0), 1, (a1, b1), 1, -MU, MU, mask_size=2, team=True)
    assert (ga == ea[0]).all() and gb == eb[0]


def test_bootstrap_wave_body_fft_mask_size_2(orc):
    """tlwe_mask_size = 2 with the FFT transform (the reference runs this pair through its multi-kernel
    driver, test/test_gates.py:88-100): the brfk_* body on a reduced number of rows of a full-range
    int32 TGSW key == the EXACT (NTT) oracle on the same key."""
    from oracle import oracle_fft as of
    rs = numpy.random.RandomState(62)
    n = 6
    tgsw = rs.randint(-2**31, 2**31, size=(n, 3, 2, 3, 1024), dtype=numpy.int32)
    bk_ntt = orc.tlwe_transform_samples(tgsw)                 # reference format (natural order, prepared)
    bkf = emu.bkf_from_reference(of.fft_forward(tgsw))        # FFT key in the wave layout
    MU = 2**29
    a0 = rs.randint(-2**31, 2**31, size=n, dtype=numpy.int32)
    a1 = rs.randint(-2**31, 2**31, size=n, dtype=numpy.int32)
    b0 = rs.randint(-2**31, 2**31, size=1, dtype=numpy.int32)
    b1 = rs.randint(-2**31, 2**31, size=1, dtype=numpy.int32)
    ta = (a0 + a1).astype(numpy.int32); tb = (numpy.int32(-MU) + b0 + b1).astype(numpy.int32)
    ea, eb = orc.bootstrap_extract(bk_ntt, ta[None, :], tb, MU)
    ga, gb = emu.bootstrap_bit_fft(bkf, n, (a0, b0), 1, (a1, b1), 1, -MU, MU, mask_size=2)
    assert ea.shape == (1, 2048)
    assert (ga == ea[0]).all() and gb == eb[0]
    # the 3-wave team variant (small batches; different fp64 summation order): still the exact result
    ga, gb = emu.bootstrap_bit_fft(bkf, n, (a0, b0), 1, (a1, b1), 1, -MU, MU, mask_size=2, team=True)
    assert (ga == ea[0]).all() and gb == eb[0]
    # the 3-wave ring variant (no partial-sum buffer, yet another summation order)
    ga, gb = emu.bootstrap_bit_fft(bkf, n, (a0, b0), 1, (a1, b1), 1, -MU, MU, mask_size=2, ring=True)
    assert (ga == ea[0]).all() and gb == eb[0]
    # the 6-wave team (brfq_* with K = 2: one transform each way per wave, partial sums handed over)
    ga, gb = emu.bootstrap_bit_fft(bkf, n, (a0, b0), 1, (a1, b1), 1, -MU, MU, mask_size=2, quad=True)
    assert (ga == ea[0]).all() and gb == eb[0]
