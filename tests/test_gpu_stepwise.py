"""
The reference's low-level functions of the bootstrap path under their own names (polynomials.py, tlwe.py, tgsw.py,
numeric_functions.py, bootstrap.py) and its multi-kernel mode (`single_kernel_bootstrap=False`): every step against the
CPU oracle, shaped like the reference's unit tests (test/test_polynomials.py:30-58, test_tlwe.py:39-94,
test_tgsw.py:118-154, test_numeric_functions.py:27-45), and whole gates through the step-by-step driver against the
fused kernels -- two independently composed device paths that must agree on every word.
"""

import numpy
import pytest

pytestmark = pytest.mark.gpu

N = 1024


@pytest.fixture(scope='module')
def env(orc, oracle_keys):
    import gpu_helpers as H
    from nufhe_amd.device import DeviceThread
    import nufhe_amd
    thr = DeviceThread(0)
    lwe_key, tlwe_key, ck = oracle_keys
    cloud_key = H.cloud_key_from_arrays(thr, ck)
    ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(456), thread=thr)
    return dict(H=H, thr=thr, ctx=ctx, ck=ck, lwe_key=lwe_key, cloud_key=cloud_key)


def test_t32_to_phase_vs_oracle(env, orc):
    from nufhe_amd.numeric_functions import t32_to_phase, Torus32
    H = env['H']; thr = env['thr']
    rs = numpy.random.RandomState(1)
    for shape, mspace in (((10, 20), 2048), ((7,), 6), ((3, 2, 5), 2**31), ((0,), 2048)):
        phase = rs.randint(-2**31, 2**31, size=shape, dtype=numpy.int64).astype(numpy.int32)
        if phase.size:
            phase.flat[0] = -1
            phase.flat[-1] = 2**31 - 1
        result = thr.array(shape, Torus32)
        t32_to_phase(thr, result, H.dev(thr, phase), mspace)
        assert (H.host(result) == orc.t32_to_phase(phase, mspace)).all(), (shape, mspace)
    with pytest.raises(ValueError):
        t32_to_phase(thr, thr.array((4,), Torus32), thr.array((5,), Torus32), 2048)
    with pytest.raises(ValueError):
        t32_to_phase(thr, thr.array((4,), Torus32), thr.array((4,), Torus32), 0)


def test_shift_torus_polynomials_vs_oracle(env, orc):
    """X^(2N - p) * source with one power per polynomial; (X^p - 1) * source with the power taken from column
    ``power_idx`` of a 2-D array and shared by the trailing axes (test/test_polynomials.py:30-58)."""
    from nufhe_amd.polynomials import (TorusPolynomialArray, shift_tp_inverted_power,
                                       shift_tp_minus_one_power_from_array)
    H = env['H']; thr = env['thr']
    rs = numpy.random.RandomState(2)
    shape = (2, 3)
    source = rs.randint(-2**31, 2**31, size=shape + (N,), dtype=numpy.int64).astype(numpy.int32)
    powers = rs.randint(0, 2 * N, size=shape).astype(numpy.int32)
    powers[0, 0] = 0; powers[0, 1] = N; powers[1, 2] = 2 * N - 1
    src = TorusPolynomialArray(H.dev(thr, source))
    res = TorusPolynomialArray.empty(thr, N, shape)
    shift_tp_inverted_power(thr, res, H.dev(thr, powers), src)
    assert (H.host(res.coeffs) == orc.shift_torus_polynomial(source, powers, invert_powers=True)).all()

    batch, k1, rows = (4,), 2, 5
    source = rs.randint(-2**31, 2**31, size=batch + (k1, N), dtype=numpy.int64).astype(numpy.int32)
    powers = rs.randint(0, 2 * N, size=batch + (rows,)).astype(numpy.int32)
    powers[0, 3] = 0; powers[1, 3] = N
    src = TorusPolynomialArray(H.dev(thr, source))
    res = TorusPolynomialArray.empty(thr, N, batch + (k1,))
    for idx in (0, 3, rows - 1):
        shift_tp_minus_one_power_from_array(thr, res, H.dev(thr, powers), idx, src)
        assert (H.host(res.coeffs) == orc.shift_torus_polynomial(source, powers[:, idx], minus_one=True)).all(), idx
    with pytest.raises(ValueError):
        shift_tp_minus_one_power_from_array(thr, res, H.dev(thr, powers), rows, src)
    with pytest.raises(ValueError):
        shift_tp_inverted_power(thr, src, H.dev(thr, powers[:, 0]), src)          # in place
    with pytest.raises(ValueError):
        shift_tp_inverted_power(thr, res, H.dev(thr, powers[:3, 0]), src)         # powers of another batch


def test_tlwe_trivial_extract_add_copy_vs_oracle(env, orc):
    """test/test_tlwe.py:39-94 for both mask sizes"""
    from nufhe_amd.polynomials import TorusPolynomialArray
    from nufhe_amd.tlwe import (TLweParams, TLweSampleArray, tlwe_noiseless_trivial, tlwe_extract_lwe_samples,
                                tlwe_add_to, tlwe_copy)
    from nufhe_amd.lwe import LweSampleArray
    H = env['H']; thr = env['thr']
    rs = numpy.random.RandomState(3)
    for k in (1, 2):
        params = TLweParams(N, k, 1e-9, 1e-3, 'NTT')
        shape = (3, 2)
        mu = rs.randint(-2**31, 2**31, size=shape + (N,), dtype=numpy.int64).astype(numpy.int32)
        sample = TLweSampleArray.empty(thr, params, shape)
        sample.a.coeffs.fill_(7); sample.current_variances.fill_(3.0)
        tlwe_noiseless_trivial(thr, sample, TorusPolynomialArray(H.dev(thr, mu)))
        exp_a, exp_cv = orc.tlwe_noiseless_trivial(mu, k)
        assert (H.host(sample.a.coeffs) == exp_a).all() and (H.host(sample.current_variances) == exp_cv).all()

        a = rs.randint(-2**31, 2**31, size=shape + (k + 1, N), dtype=numpy.int64).astype(numpy.int32)
        sample = TLweSampleArray(params, TorusPolynomialArray(H.dev(thr, a)), thr.zeros(shape, numpy.float32))
        out = LweSampleArray.empty(thr, params.extracted_lweparams, shape)
        tlwe_extract_lwe_samples(thr, out, sample)
        exp_a, exp_b = orc.tlwe_extract_lwe_samples(a)
        assert (H.host(out.a) == exp_a).all() and (H.host(out.b) == exp_b).all()
        with pytest.raises(ValueError):
            tlwe_extract_lwe_samples(thr, LweSampleArray.empty(thr, params.extracted_lweparams, (3,)), sample)

        other = TLweSampleArray.empty(thr, params, shape)
        tlwe_copy(thr, other, sample)
        other.current_variances.fill_(0.5)
        tlwe_add_to(thr, other, sample)
        assert (H.host(other.a.coeffs) == (a.astype(numpy.int64) * 2).astype(numpy.int32)).all()     # wraps
        assert (H.host(other.current_variances) == 0.5).all()


def test_external_mul_and_blind_rotate_steps_vs_oracle(env, orc):
    """tgsw_transformed_external_mul on rows of the real key (test/test_tgsw.py:118-154), then `blind_rotate` over
    the first rows composed from mux_rotate steps: against the oracle and against the fused loop of the library."""
    from nufhe_amd import _lib
    from nufhe_amd.device import ptr
    from nufhe_amd.polynomials import TorusPolynomialArray
    from nufhe_amd.tlwe import TLweSampleArray
    from nufhe_amd.tgsw import tgsw_transformed_external_mul
    from nufhe_amd.bootstrap import blind_rotate
    H = env['H']; thr = env['thr']; ck = env['ck']; bk = env['cloud_key'].bootstrap_key
    rs = numpy.random.RandomState(4)
    shape = (3,)
    accum = rs.randint(-2**31, 2**31, size=shape + (2, N), dtype=numpy.int64).astype(numpy.int32)
    params = bk.accum_params
    for row in (0, 7, 499):
        sample = TLweSampleArray(params, TorusPolynomialArray(H.dev(thr, accum)), thr.zeros(shape, numpy.float32))
        tgsw_transformed_external_mul(thr, sample, bk.tgsw, row)
        assert (H.host(sample.a.coeffs) == orc.tgsw_external_mul(accum, ck.bk, row)).all(), row
    with pytest.raises(ValueError):
        tgsw_transformed_external_mul(thr, sample, bk.tgsw, 500)

    rows = 9                                                   # odd: the result ends in the temporary and is copied back
    bara = rs.randint(0, 2 * N, size=shape + (500,)).astype(numpy.int32)
    bara[0, 2] = 0
    sample = TLweSampleArray(params, TorusPolynomialArray(H.dev(thr, accum)), thr.zeros(shape, numpy.float32))
    blind_rotate(thr, sample, bk, H.dev(thr, bara), rows)
    exp = orc.blind_rotate(accum, ck.bk, bara, n_iter=rows)
    assert (H.host(sample.a.coeffs) == exp).all()
    fused = H.dev(thr, accum)
    _lib.call("nufhe_blind_rotate", thr.handle, bk._native.handle, ptr(fused), ptr(H.dev(thr, bara)), 500, rows, 3)
    assert (H.host(fused) == exp).all()


def _stepwise_vm(env):
    import nufhe_amd
    params = env['cloud_key'].params
    perf = nufhe_amd.PerformanceParameters(params, single_kernel_bootstrap=False)
    return env['ctx'].make_virtual_machine(env['cloud_key'], perf_params=perf)


def test_gates_step_by_step_equal_fused_and_oracle(env, orc):
    """NAND, XOR (multiplying linear forms), ANDNY (mixed signs) and MUX through the reference's multi-kernel sequence
    -- trivial constant, linear combinations, mod-switch, shift, trivial accumulator, 500 x (shift, external product,
    add), extract, keyswitch, each one launch -- against the fused gate kernels and the oracle: every word and
    variance identical."""
    H = env['H']; thr = env['thr']; ck = env['ck']; lwe_key = env['lwe_key']
    rng = orc.DeterministicRNG(31)
    shape = (2, 3)
    ms = [rng.uniform_bool(shape).astype(bool) for _ in range(3)]
    cs = [orc.encrypt(rng, lwe_key, m) for m in ms]
    ds = [H.ciphertext_from_arrays(thr, c) for c in cs]
    fused = env['ctx'].make_virtual_machine(env['cloud_key'])
    steps = _stepwise_vm(env)
    assert steps.perf_params.single_kernel_bootstrap is False
    for name in ('gate_nand', 'gate_xor', 'gate_andny'):
        got = H.ct_arrays(getattr(steps, name)(ds[0], ds[1]))
        ref = H.ct_arrays(getattr(fused, name)(ds[0], ds[1]))
        exp = orc.gate(name, ck, cs[0], cs[1])
        for g, r, e in zip(got, ref, exp):
            assert (g == r).all() and (g == e).all(), name
    got = H.ct_arrays(steps.gate_mux(ds[0], ds[1], ds[2]))
    ref = H.ct_arrays(fused.gate_mux(ds[0], ds[1], ds[2]))
    exp = orc.gate_mux(ck, cs[0], cs[1], cs[2])
    for g, r, e in zip(got, ref, exp):
        assert (g == r).all() and (g == e).all()
    # broadcasting and a preallocated destination behave as in the fused mode
    row = H.ciphertext_from_arrays(thr, tuple(x[0] for x in cs[1]))           # shape (3,)
    dest = env['ctx'].make_virtual_machine(env['cloud_key']).empty_ciphertext(shape)
    steps.gate_nand(ds[0], row, dest=dest)
    ref = H.ct_arrays(fused.gate_nand(ds[0], row))
    assert all((g == r).all() for g, r in zip(H.ct_arrays(dest), ref))


def test_bootstrap_without_keyswitch_step_by_step(env, orc):
    """`bootstrap(..., no_keyswitch=True)` through the step-by-step driver: the extracted LWE(1024) sample equals the
    fused kernel's (gates.py:633-655 building block)."""
    import nufhe_amd
    from nufhe_amd.bootstrap import bootstrap
    from nufhe_amd import lwe as L
    H = env['H']; thr = env['thr']; lwe_key = env['lwe_key']; cloud_key = env['cloud_key']
    rng = orc.DeterministicRNG(32)
    m = rng.uniform_bool((4,)).astype(bool)
    x = H.ciphertext_from_arrays(thr, orc.encrypt(rng, lwe_key, m))
    bk, ks = cloud_key.bootstrap_key, cloud_key.keyswitch_key
    perf = nufhe_amd.PerformanceParameters(cloud_key.params, single_kernel_bootstrap=False).for_device()
    a = L.LweSampleArray.empty(thr, bk.extract_params, (4,))
    b = L.LweSampleArray.empty(thr, bk.extract_params, (4,))
    bootstrap(thr, a, bk, ks, 2**29, x, perf, no_keyswitch=True)
    bootstrap(thr, b, bk, ks, 2**29, x, None, no_keyswitch=True)
    assert (H.host(a.a) == H.host(b.a)).all() and (H.host(a.b) == H.host(b.b)).all()
    # both modes broadcast their argument to the result's shape: one ciphertext (shape (4,)) into a (3, 4) result
    a2 = L.LweSampleArray.empty(thr, bk.extract_params, (3, 4))
    b2 = L.LweSampleArray.empty(thr, bk.extract_params, (3, 4))
    bootstrap(thr, a2, bk, ks, 2**29, x, perf, no_keyswitch=True)
    bootstrap(thr, b2, bk, ks, 2**29, x, None, no_keyswitch=True)
    assert (H.host(a2.a) == H.host(b2.a)).all() and (H.host(a2.b) == H.host(b2.b)).all()
    assert all((H.host(a2.a)[r] == H.host(a.a)).all() for r in range(3))
    with pytest.raises(ValueError):
        bootstrap(thr, L.LweSampleArray.empty(thr, bk.extract_params, (3, 5)), bk, ks, 2**29, x, perf, no_keyswitch=True)


def test_key_generation_steps_vs_oracle(env, orc):
    """The reference's key-generation functions under their own signatures (tgsw.py:134-161, tlwe.py:185-207;
    test/test_tlwe.py:97-141, test_tgsw.py:157-196 shapes): TLWE / TGSW encryptions of zero from the reference's random
    draws, the gadget message, and the transform into the key the external product reads."""
    import nufhe_amd
    from nufhe_amd.tlwe import TLweKey, TLweSampleArray, tlwe_encrypt_zero
    from nufhe_amd.polynomials import IntPolynomialArray
    from nufhe_amd.tgsw import (TGswKey, TGswSampleArray, TransformedTGswSampleArray, tgsw_encrypt_zero, tgsw_add_message,
                                tgsw_encrypt_int, tgsw_transform_samples, tgsw_transformed_external_mul)
    from nufhe_amd.tlwe import TLweSampleArray as Acc
    from nufhe_amd.polynomials import TorusPolynomialArray
    H = env['H']; thr = env['thr']
    params = nufhe_amd.NuFHEParameters()
    tgsw_params = params.tgsw_params
    tlwe_params = tgsw_params.tlwe_params
    noise = tlwe_params.min_noise
    key_bits = numpy.random.RandomState(9).randint(0, 2, size=(1, N)).astype(numpy.int32)
    tlwe_key = TLweKey(tlwe_params, IntPolynomialArray(H.dev(thr, key_bits)))

    # TLWE encryptions of zero: the same draws through the oracle's RNG (uniform mask, then Gaussian noise)
    shape = (3, 2)
    seed = 77
    sample = TLweSampleArray.empty(thr, tlwe_params, shape)
    tlwe_encrypt_zero(thr, nufhe_amd.DeterministicRNG(seed), sample, noise, tlwe_key)
    orng = orc.DeterministicRNG(seed)
    n1 = orng.uniform_torus32(shape + (1, N))
    n2 = orc.rand_gaussian_torus32(orng, 0, noise, shape + (N,))
    exp_a, exp_cv = orc.tlwe_encrypt_zero(key_bits, n1, n2, noise)
    assert (H.host(sample.a.coeffs) == exp_a).all() and (H.host(sample.current_variances) == exp_cv).all()

    # TGSW(message): zero encryptions + gadget; then the transform, used by one external product
    n = 5
    messages = numpy.array([1, 0, 1, 1, 0], numpy.int32)
    tgsw_key = TGswKey(tgsw_params, tlwe_key)
    tgsw = TGswSampleArray.empty(thr, tgsw_params, (n,))
    assert tgsw.shape == (n,) and tuple(tgsw.samples.a.coeffs.shape) == (n, 2, 2, 2, N)
    tgsw_encrypt_int(thr, nufhe_amd.DeterministicRNG(seed + 1), tgsw, H.dev(thr, messages), noise, tgsw_key)
    orng = orc.DeterministicRNG(seed + 1)
    n1 = orng.uniform_torus32((n, 2, 2, 1, N))
    n2 = orc.rand_gaussian_torus32(orng, 0, noise, (n, 2, 2, N))
    zero_a, _ = orc.tlwe_encrypt_zero(key_bits, n1, n2, noise)
    exp = orc.tgsw_add_message(zero_a, messages)
    assert (H.host(tgsw.samples.a.coeffs) == exp).all()
    again = TGswSampleArray.empty(thr, tgsw_params, (n,))
    tgsw_encrypt_zero(thr, nufhe_amd.DeterministicRNG(seed + 1), again, noise, tgsw_key)
    assert (H.host(again.samples.a.coeffs) == zero_a).all()
    tgsw_add_message(thr, again, H.dev(thr, messages))
    assert (H.host(again.samples.a.coeffs) == exp).all()
    with pytest.raises(ValueError):
        tgsw_add_message(thr, again, H.dev(thr, messages[:3]))

    transformed = TransformedTGswSampleArray.empty(thr, tgsw_params, (n,))
    tgsw_transform_samples(thr, transformed, tgsw)
    bk = orc.tlwe_transform_samples(exp)                      # reference format: forward NTT + Montgomery preparation
    accum = numpy.random.RandomState(10).randint(-2**31, 2**31, size=(2, 2, N), dtype=numpy.int64).astype(numpy.int32)
    acc = Acc(tlwe_params, TorusPolynomialArray(H.dev(thr, accum)), thr.zeros((2,), numpy.float32))
    tgsw_transformed_external_mul(thr, acc, transformed, 3)
    assert (H.host(acc.a.coeffs) == orc.tgsw_external_mul(accum, bk, 3)).all()


def test_tlwe_transform_samples_reference_format(env, orc):
    """`tlwe_transform_samples` into a TransformedTLweSampleArray: the reference's storage form of transformed samples
    (natural order; NTT values Montgomery-prepared) -- equal to the oracle's, to the rows of the reference-format
    bootstrapping key the library exports, and stable through dump / load (tlwe.py:115-153,199-207)."""
    import io
    import nufhe_amd
    from oracle import oracle_fft
    from nufhe_amd.polynomials import TorusPolynomialArray, TransformedPolynomialArray
    from nufhe_amd.tlwe import TLweSampleArray, TransformedTLweSampleArray, tlwe_transform_samples
    H = env['H']; thr = env['thr']
    rs = numpy.random.RandomState(12)
    shape = (2, 3)
    a = rs.randint(-2**31, 2**31, size=shape + (2, N), dtype=numpy.int64).astype(numpy.int32)
    cv = rs.rand(*shape).astype(numpy.float32)
    for transform in ('NTT', 'FFT'):
        params = nufhe_amd.NuFHEParameters(transform_type=transform).tgsw_params.tlwe_params
        source = TLweSampleArray(params, TorusPolynomialArray(H.dev(thr, a)), H.dev(thr, cv))
        result = TransformedTLweSampleArray.empty(thr, params, shape)
        assert tuple(result.a.coeffs.shape) == shape + (2, N if transform == 'NTT' else N // 2)
        tlwe_transform_samples(thr, result, source)
        got = result.a._host()
        if transform == 'NTT':
            assert (got == orc.tlwe_transform_samples(a)).all()
        else:
            exp = oracle_fft.fft_forward(a)
            assert numpy.abs(got - exp).max() <= 1e-12 * numpy.abs(exp).max()
        assert (H.host(result.current_variances) == cv).all()
        buf = io.BytesIO()
        result.dump(buf)
        buf.seek(0)
        back = TransformedTLweSampleArray.load(buf, thr)
        assert back == result and back.a.transform_type == transform
        with pytest.raises(TypeError):
            TransformedPolynomialArray(transform, H.dev(thr, a))
    # the key the library holds, exported in the reference's format, is the same transform of the same TGSW samples
    ck = env['ck']
    tgsw = oracle_fft.tgsw_coeffs_from_reference_bk(ck.bk[:2])          # int32 [2, 2, 2, 2, N]
    params = nufhe_amd.NuFHEParameters().tgsw_params.tlwe_params
    source = TLweSampleArray(params, TorusPolynomialArray(H.dev(thr, tgsw)), thr.zeros((2, 2, 2), numpy.float32))
    result = TransformedTLweSampleArray.empty(thr, params, (2, 2, 2))
    tlwe_transform_samples(thr, result, source)
    assert (result.a._host() == numpy.asarray(ck.bk[:2], numpy.uint64)).all()


def test_step_by_step_driver_with_fft_and_mask_size_2(env):
    """The step-by-step driver is transform- and k-agnostic (every step goes through the per-kernel entry points, which
    take both from the key): NAND and MUX on keys generated here, against the fused kernels -- exact for the NTT with
    k = 2, within the FFT tolerance (2^4 LSB, tests/test_gpu_fft.py) for the FFT -- and correct after decryption."""
    import nufhe_amd
    H = env['H']; thr = env['thr']
    ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(77), thread=thr)
    bits = [numpy.array([True, False, True]), numpy.array([True, True, False]), numpy.array([False, True, True])]
    for kwds, tol in ((dict(transform_type='FFT'), 16), (dict(tlwe_mask_size=2), 0)):
        sk, ck = ctx.make_key_pair(**kwds)
        cts = [ctx.encrypt(sk, b) for b in bits]
        fused = ctx.make_virtual_machine(ck)
        steps = ctx.make_virtual_machine(ck, perf_params=nufhe_amd.PerformanceParameters(ck.params, single_kernel_bootstrap=False))
        for got, ref, truth in ((steps.gate_nand(cts[0], cts[1]), fused.gate_nand(cts[0], cts[1]), ~(bits[0] & bits[1])),
                                (steps.gate_mux(*cts), fused.gate_mux(*cts), numpy.where(bits[0], bits[1], bits[2]))):
            assert (ctx.decrypt(sk, got) == truth).all(), kwds
            for g, r in zip(H.ct_arrays(got)[:2], H.ct_arrays(ref)[:2]):
                diff = (g.astype(numpy.int64) - r.astype(numpy.int64) + 2**31) % 2**32 - 2**31
                assert numpy.abs(diff).max() <= tol, (kwds, int(numpy.abs(diff).max()))
