"""
Pin the CPU oracle (oracle/nufhe_oracle.c) against golden vectors produced by the REFERENCE's
own CPU reference functions (tests/golden/make_golden.py) and against the known-answer
constants the reference's tests hold.  CPU only.
"""

import numpy
import pytest

import golden_inputs as gi

P = 2**64 - 2**32 + 1


def eq(a, b):
    a = numpy.asarray(a); b = numpy.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert a.dtype == b.dtype, (a.dtype, b.dtype)
    assert (a == b).all()


# ---- known-answer constants ------------------------------------------------------------------

def test_kat_constants(orc, golden):
    # root of unity: nufhe/transform/ntt_cpu.py:109, doc/source/implementation_details.rst:121
    assert orc.root_of_unity(2**32) == 0xa70dc47e4cbdf43f == 12037493425763644479
    assert orc.root_of_unity(64) == 8 == int(golden['kat_root_64'])
    # R^-1 and R mod P: test/test_transform/test_arithmetic.py:178,193
    assert int(golden['kat_rinv']) == 0xfffffffe00000001
    assert int(golden['kat_r']) == 0xffffffff
    one = numpy.array([1], numpy.uint64)
    assert int(orc.ff_mul_prepared(one, one)[0]) == 0xfffffffe00000001
    assert int(orc.ff_prepare_for_mul(one)[0]) == 0xffffffff


def test_kat_mod_boundaries(orc):
    # test/test_transform/test_arithmetic.py:152-157
    x = numpy.array([P - 1, P, P + 1, 0, 2**64 - 1], numpy.uint64)
    eq(orc.ff_mod(x), numpy.array([P - 1, 0, 1, 0, 2**32 - 2], numpy.uint64))


def test_kat_mul_regression(orc):
    # test/test_transform/test_arithmetic.py:169-174: (P-1) * 2^33
    a = numpy.array([P - 1], numpy.uint64); b = numpy.array([2**33], numpy.uint64)
    assert int(orc.ff_mul(a, b)[0]) == ((P - 1) * 2**33) % P


def test_kat_lsh_regression(orc):
    # test/test_transform/test_arithmetic.py:243-249
    a = numpy.array([11509900421665959066] * 6, numpy.uint64)
    s = numpy.array([31, 63, 95, 127, 159, 191], numpy.uint32)
    res = orc.ff_lsh(a, s)
    for r, sh in zip(res, s):
        assert int(r) == (11509900421665959066 * 2**int(sh)) % P


def test_kat_gnum_to_i32(orc):
    # test/test_transform/test_ntt_cpu.py:60-67 (the NumPy>=2-safe intended values)
    vals = numpy.array([0, 1, 2**31 - 1, 2**31, 2**32 - 1, P - 1, P - 2**31, P // 2, P // 2 + 1],
                       numpy.uint64)
    expected = []
    for v in vals:
        v = int(v)
        x = ((v & 0xffffffff) - (1 if v > P // 2 else 0)) & 0xffffffff
        expected.append(x - 2**32 if x >= 2**31 else x)
    eq(orc.ff_to_i32(vals), numpy.array(expected, numpy.int32))
    assert orc.ff_to_i32(numpy.array([P - 1], numpy.uint64))[0] == -1
    assert orc.ff_to_i32(numpy.array([P - 2**31], numpy.uint64))[0] == -2**31


def test_prepare_for_mul_matches_bigint(orc):
    # test/test_transform/test_arithmetic.py:204-208 (test_prepare_for_mul_cpu)
    rs = numpy.random.RandomState(5)
    x = rs.randint(0, P, size=200, dtype=numpy.uint64)
    exp = numpy.array([(int(v) * 2**64) % P for v in x], numpy.uint64)
    eq(orc.ff_prepare_for_mul(x), exp)


# ---- finite field vs the reference's GaloisNumber ---------------------------------------------

def test_ff_arithmetic(orc, golden):
    a, b, e, s = gi.ff_inputs()
    eq(orc.ff_add(a, b), golden['ff_add'])
    eq(orc.ff_sub(a, b), golden['ff_sub'])
    eq(orc.ff_mul(a, b), golden['ff_mul'])
    eq(orc.ff_mul_prepared(a, b), golden['ff_mul_prepared'])
    eq(orc.ff_pow(a, e), golden['ff_pow'])
    eq(orc.ff_lsh(a, s), golden['ff_lsh'])
    eq(orc.ff_to_i32(a), golden['ff_to_i32'])


def test_inv_pow2(orc):
    e = numpy.arange(0, 200, dtype=numpy.uint32)
    r = orc.ff_inv_pow2(e)
    for x, k in zip(r, e):
        assert (int(x) * 2**int(k)) % P == 1


# ---- transforms -------------------------------------------------------------------------------

def test_ntt_vs_reference(orc, golden):
    polys_i32, polys_ff = gi.ntt_inputs()
    eq(orc.ntt_forward(polys_i32, True), golden['ntt_forward_i32'])
    eq(orc.ntt_forward(polys_ff, False), golden['ntt_forward_u64'])
    eq(orc.ntt_inverse(polys_ff, True), golden['ntt_inverse_i32'])
    eq(orc.ntt_inverse(polys_ff, False), golden['ntt_inverse_u64'])
    eq(orc.ntt_forward(gi.ntt_small_inputs(), True), golden['ntt_small_forward'])


def test_ntt_roundtrip_and_product(orc):
    # test/test_transform/test_computation.py:71-124: transform-based negacyclic product of a
    # full-range i32 polynomial with a small one equals the schoolbook product mod 2^32
    rs = numpy.random.RandomState(7)
    a = rs.randint(-2**31, 2**31, size=(3, 1024), dtype=numpy.int32)
    b = rs.randint(-1000, 1000, size=(3, 1024)).astype(numpy.int32)
    eq(orc.ntt_inverse(orc.ntt_forward(a)), a)
    prod = orc.ntt_inverse(orc.ff_mul(orc.ntt_forward(a), orc.ntt_forward(b)))
    eq(prod, orc.poly_mul_schoolbook(a, b))


def test_ntt_naive_small(orc):
    # test/test_transform/test_ntt_cpu.py:24-38: fast == O(N^2) definition (here with big ints)
    N = 16
    rs = numpy.random.RandomState(8)
    a = rs.randint(-2**31, 2**31, size=(N,), dtype=numpy.int32)
    psi = orc.root_of_unity(2 * N)
    exp = [sum((int(a[j]) % P) * pow(psi, (2 * k + 1) * j, P) for j in range(N)) % P for k in range(N)]
    eq(orc.ntt_forward(a[None, :])[0], numpy.array(exp, numpy.uint64))


# ---- small ops --------------------------------------------------------------------------------

def test_t32_to_phase(orc, golden):
    eq(orc.t32_to_phase(gi.modswitch_inputs(), 2048), golden['t32_to_phase'])


def test_shift(orc, golden):
    for tag, (src, powers, N) in gi.shift_inputs().items():
        for minus_one in (False, True):
            for invert in (False, True):
                eq(orc.shift_torus_polynomial(src, powers, minus_one, invert),
                   golden['shift_%s_m%d_i%d' % (tag, minus_one, invert)])
    src, powers_arr, idx, N = gi.shift_view_inputs()
    eq(orc.shift_torus_polynomial(src, powers_arr[:, idx], True, False), golden['shift_view'])


def test_tlwe_trivial_extract(orc, golden):
    a, cv = orc.tlwe_noiseless_trivial(gi.tlwe_trivial_inputs(), 1)
    eq(a, golden['tlwe_trivial_a'])
    assert (cv == 0).all()
    ra, rb = orc.tlwe_extract_lwe_samples(gi.tlwe_extract_inputs())
    eq(ra, golden['tlwe_extract_a']); eq(rb, golden['tlwe_extract_b'])
    ra, rb = orc.tlwe_extract_lwe_samples(gi.tlwe_extract_inputs(mask_size=2))
    eq(ra, golden['tlwe_extract2_a']); eq(rb, golden['tlwe_extract2_b'])


def test_tgsw_decomp(orc, golden):
    eq(orc.tgsw_decomp(gi.decomp_inputs()), golden['tgsw_decomp'])


def test_tgsw_mac(orc, golden):
    tr_sample, bk, row = gi.mac_inputs()
    eq(orc.tlwe_transformed_add_mul(tr_sample, bk, row), golden['tgsw_mac'])


def test_tgsw_external_mul(orc, golden):
    accum, bk, row = gi.extmul_inputs()
    eq(orc.tgsw_external_mul(accum, bk, row), golden['tgsw_extmul'])
    accum, bk, row = gi.extmul_inputs(full_range=True)
    eq(orc.tgsw_external_mul(accum, bk, row), golden['tgsw_extmul_full'])


def test_keyswitch(orc, golden):
    ks_a, ks_b, ks_cv, src_a, src_b = gi.keyswitch_inputs()
    ra, rb, rcv = orc.lwe_keyswitch(ks_a, ks_b, ks_cv, src_a, src_b)
    eq(ra, golden['ks_a']); eq(rb, golden['ks_b'])
    # sequential float32 accumulation in (l, j) order is bit-identical to the reference
    eq(rcv, golden['ks_cv'])


def test_lwe_linear(orc, golden):
    res, src = gi.linear_inputs()
    for p, add in ((1, False), (-1, True), (2, True), (-2, True)):
        ra, rb, rcv = orc.lwe_linear(res, src, p, add)
        eq(ra, golden['linear_p%d_add%d_a' % (p, add)])
        eq(rb, golden['linear_p%d_add%d_b' % (p, add)])
        eq(rcv, golden['linear_p%d_add%d_cv' % (p, add)])


# ---- key generation pieces --------------------------------------------------------------------

def test_encrypt_zero(orc, golden):
    key, n1, n2 = gi.encrypt_zero_inputs()
    ra, rcv = orc.tlwe_encrypt_zero(key, n1, n2, 9e-9)
    eq(ra, golden['encrypt_zero_a']); eq(rcv, golden['encrypt_zero_cv'])


def test_add_message(orc, golden):
    tgsw_a, msgs = gi.add_message_inputs()
    eq(orc.tgsw_add_message(tgsw_a, msgs), golden['add_message'])


def test_ks_keygen(orc, golden):
    in_key, out_key, na, nb = gi.ks_keygen_inputs()
    ks_a, ks_b, ks_cv = orc.make_lwe_keyswitch_key(in_key, out_key, na, nb, 1e-3, 8, 2)
    eq(ks_a, golden['kskey_a']); eq(ks_b, golden['kskey_b']); eq(ks_cv, golden['kskey_cv'])


def test_lwe_encrypt_decrypt(orc, golden):
    msgs, key, na, nb = gi.lwe_encrypt_inputs()
    ra, rb, rcv = orc.lwe_encrypt(msgs, key, na, nb, 1e-3)
    eq(ra, golden['lwe_encrypt_a']); eq(rb, golden['lwe_encrypt_b'])
    eq(orc.lwe_decrypt(ra, rb, key), golden['lwe_decrypt'])


# ---- composition ------------------------------------------------------------------------------

def test_blind_rotate_composition(orc, golden):
    acc, bk, bara = gi.blind_rotate_inputs()
    res = orc.blind_rotate(acc, bk, bara)
    eq(res, golden['blind_rotate_acc'])
    ra, rb = orc.tlwe_extract_lwe_samples(res)
    eq(ra, golden['blind_rotate_ext_a']); eq(rb, golden['blind_rotate_ext_b'])


# ---- FFT transform path ------------------------------------------------------------------------

def test_fft_oracle_vs_reference(orc, golden):
    """numpy restatement (oracle/oracle_fft.py) vs the reference's functions run with
    transform_type='FFT'; and both equal the exact (NTT) external product on these inputs."""
    from oracle import oracle_fft as of
    polys_i32, _ = gi.ntt_inputs()
    assert numpy.allclose(of.fft_forward(polys_i32), golden['fft_forward'], rtol=1e-12, atol=1e-3)
    eq(of.fft_inverse(golden['fft_forward']), golden['fft_inverse_of_forward'])
    eq(golden['fft_inverse_of_forward'], polys_i32)
    accum, tgsw, row = gi.fft_extmul_inputs()
    res = of.external_mul(accum, of.bk_from_coeffs(tgsw), row)
    eq(res, golden['fft_extmul'])
    exact = orc.tgsw_external_mul(accum, orc.tlwe_transform_samples(tgsw), row)
    eq(golden['fft_extmul'], exact)
