"""
GPU tests of the FFT transform path (BASELINE config 5).

Stated tolerance: fp64 rounding may move a coefficient of an external product by
+-1 torus LSB (2^-32) when the rounding error of the complex arithmetic reaches 0.5 before
`round`; the contract is
  * transforms: relative error <= 1e-12 vs numpy.fft (the reference asserts `allclose`,
    test/test_transform/test_computation.py:67-68),
  * product of a full-range int32 polynomial with a small one: EXACT (test_computation.py:106-124),
  * one external product: max |delta| <= 1 LSB vs the exact (NTT) result on <= 1e-4 of coefficients,
  * a whole gate (500 external products + keyswitch): decrypted bits identical; every output word
    within 2^4 LSB (2^-28 torus) of the exact path.
Why so tight: the values handed to `round` are the exact integers plus the fp64 error of the
transforms, about 10 * 2^-53 * |v| with |v| <= 2^47 on real keys, i.e. < 0.06 -- an eighth of the 0.5 at
which a rounding decision could flip (measured over a whole blind rotation on the host build of the
same code: tests/test_emu_device_code.py::test_bootstrap_wave_body_fft_full_key, max 0.055).  Every
external product therefore rounds to the exact result and the observed deviation is ZERO (bit-identical
to the exact path), which the tests also record; the 2^4 allowance covers keys / inputs whose products
approach the 2^52 worst case.
"""

import ctypes

import numpy
import pytest

FFT_TOLERANCE_LSB = 2**4     # per output word of a whole gate vs the exact (NTT) path

import golden_inputs as gi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def thr():
    from nufhe_amd.device import DeviceThread
    return DeviceThread(0)


@pytest.fixture(scope='module')
def H():
    import gpu_helpers
    return gpu_helpers


def call(name, *args):
    from nufhe_amd import _lib
    _lib.call(name, *args)


def ptr(t):
    from nufhe_amd.device import ptr as p
    return p(t)


def _fft_forward(thr, H, data):
    d = H.dev(thr, data)
    out = thr.array(data.shape[:-1] + (1024,), numpy.float64)       # 512 complex = 1024 doubles
    call("nufhe_fft_forward_i32", thr.handle, ptr(out), ptr(d), data.size // 1024)
    return H.host(out).view(numpy.complex128)


def _fft_inverse(thr, H, data):
    d = H.dev(thr, numpy.ascontiguousarray(data).view(numpy.float64))
    out = thr.array(data.shape[:-1] + (1024,), numpy.int32)
    call("nufhe_fft_inverse_i32", thr.handle, ptr(out), ptr(d), data.size // 512)
    return H.host(out)


def test_fft_transform_vs_reference_golden(thr, H, golden, orc):
    polys_i32, _ = gi.ntt_inputs()
    f = _fft_forward(thr, H, polys_i32)
    ref = golden['fft_forward']
    assert numpy.abs(f - ref).max() / numpy.abs(ref).max() < 1e-12
    assert (_fft_inverse(thr, H, f) == polys_i32).all()
    assert (_fft_inverse(thr, H, ref) == golden['fft_inverse_of_forward']).all()
    # ragged batch + exact product with a small polynomial
    rs = numpy.random.RandomState(51)
    a = rs.randint(-2**31, 2**31, size=(37, 1024), dtype=numpy.int32)
    b = rs.randint(-1000, 1000, size=(37, 1024)).astype(numpy.int32)
    prod = _fft_inverse(thr, H, _fft_forward(thr, H, a) * _fft_forward(thr, H, b))
    assert (prod == orc.poly_mul_schoolbook(a, b)).all()


def _fft_key(thr, bkf):
    from nufhe_amd.bootstrap import NativeCloudKey
    native = NativeCloudKey(thr, bkf.shape[0], 'FFT')
    arr = numpy.ascontiguousarray(bkf, numpy.complex128)
    call("nufhe_bk_upload_reference", native.handle, arr.ctypes.data_as(ctypes.c_void_p))
    return native


def test_fft_external_mul_vs_reference_golden(thr, H, golden, orc):
    from oracle import oracle_fft as of
    accum, tgsw, row = gi.fft_extmul_inputs()
    bkf = of.bk_from_coeffs(tgsw)
    native = _fft_key(thr, bkf)
    back = numpy.empty_like(bkf)
    call("nufhe_bk_download_reference", native.handle, back.ctypes.data_as(ctypes.c_void_p))
    assert (back == bkf).all()
    acc = H.dev(thr, accum)
    call("nufhe_external_mul", thr.handle, native.handle, ptr(acc), row, 6)
    got = H.host(acc)
    ref = golden['fft_extmul']          # reference functions with transform_type='FFT' (== exact here)
    delta = got.astype(numpy.int64) - ref.astype(numpy.int64)
    assert numpy.abs(delta).max() <= 1 and (delta != 0).mean() <= 1e-4
    print("FFT external product: coefficients differing from the reference:", int((delta != 0).sum()))
    # the key transformed on the device (nufhe_bk_from_coeffs) gives the same product
    native2 = __import__('nufhe_amd.bootstrap', fromlist=['NativeCloudKey']).NativeCloudKey(thr, 3, 'FFT')
    d_tgsw = H.dev(thr, tgsw)
    call("nufhe_bk_from_coeffs", native2.handle, ptr(d_tgsw))
    acc2 = H.dev(thr, accum)
    call("nufhe_external_mul", thr.handle, native2.handle, ptr(acc2), row, 6)
    d2 = H.host(acc2).astype(numpy.int64) - ref.astype(numpy.int64)
    assert numpy.abs(d2).max() <= 1 and (d2 != 0).mean() <= 1e-4


def _exact_external_mul(accum, tgsw, row):
    """exact negacyclic external product mod 2^32 in integer arithmetic (tgsw_cpu.py:26-106 semantics)"""
    a = accum.astype(numpy.int64)
    t = (a + 0x80200000) & 0xFFFFFFFF
    out = numpy.zeros_like(a)
    idx = numpy.arange(1024)
    for b in range(a.shape[0]):
        for m in range(2):
            for d in range(2):
                dg = ((t[b, m] >> (32 - 10 * (d + 1))) & 1023) - 512
                for mo in range(2):
                    key = tgsw[row, m, d, mo].astype(numpy.int64)
                    full = numpy.convolve(dg, key)                   # |terms| <= 2^40, 1024 of them: fits int64
                    neg = full[:1024].copy()
                    neg[:1023] -= full[1024:]
                    out[b, mo] += neg
    return out


def test_fft_external_mul_at_the_largest_reachable_magnitude(thr, H):
    """The final rounding of the FFT path at the edge of its domain: a TGSW row whose
    coefficients are all +-2^31-ish, sign-aligned with accumulators whose gadget digits are all -512 or +511, drives
    the values handed to `round` up to 4 x 1024 x 2^9 x 2^31 = 2^52 -- beyond 2^51, where a signed magic-number
    add leaves its binade.  fft_round_to_u32 rounds the MAGNITUDE, so it follows the reference's
    round -> int64 -> truncate (transform/fft.mako:272-277) on the whole reachable range; what is left is the fp64
    error of the transforms themselves, ~2^-50 relative: at |v| ~ 2^52 that is a few units, and the tolerance
    stated for the path (2^4 LSB) is asserted here exactly where it is hardest to meet."""
    from oracle import oracle_fft as of
    rs = numpy.random.RandomState(77)
    B = 4
    # digits: t + OFFSET has all decomposition bits 0 (digits -512, -512) or all 1 (digits 511, 511)
    lo_t = numpy.uint32(0x100000000 - 0x80200000)                 # t + 0x80200000 = 0 (mod 2^32)
    hi_t = numpy.uint32((0xFFFFF000 - 0x80200000) & 0xFFFFFFFF)   # t + 0x80200000 = 0xFFFFF000
    accum = numpy.empty((B, 2, 1024), numpy.uint32)
    tgsw = numpy.empty((1, 2, 2, 2, 1024), numpy.int32)
    # bit 0: everything at the extreme, constant sign: coefficient 1023 of every output reaches -/+ 2^52
    accum[0] = lo_t
    # bit 1: digits +511
    accum[1] = hi_t
    # bits 2, 3: random choice of the two extremes per coefficient
    accum[2:] = numpy.where(rs.randint(0, 2, size=(2, 2, 1024)).astype(bool), lo_t, hi_t)
    accum += rs.randint(0, 0x1000, size=accum.shape).astype(numpy.uint32)      # low 12 bits do not reach a digit
    tgsw[:] = -2**31
    tgsw[0, :, :, 1] = 2**31 - 1
    tgsw[0, 1, 1, :, ::2] = rs.choice([-2**31, 2**31 - 1], size=(2, 512))
    accum = accum.view(numpy.int32)
    exact = _exact_external_mul(accum, tgsw, 0)
    assert numpy.abs(exact).max() >= 1.5 * 2**51 and (numpy.abs(exact) >= 2**51).sum() > 1000
    expect = (exact & 0xFFFFFFFF).astype(numpy.uint32)

    from nufhe_amd.bootstrap import NativeCloudKey
    native = NativeCloudKey(thr, 1, 'FFT')
    d_tgsw = H.dev(thr, tgsw)
    call("nufhe_bk_from_coeffs", native.handle, ptr(d_tgsw))
    acc = H.dev(thr, accum.reshape(B, 1, 2, 1024).copy())
    call("nufhe_external_mul", thr.handle, native.handle, ptr(acc), 0, B)      # the accumulator becomes the product
    got = H.host(acc).reshape(B, 2, 1024).view(numpy.uint32)
    delta = (got.astype(numpy.int64) - expect.astype(numpy.int64) + 2**31) % 2**32 - 2**31
    big = numpy.abs(exact) >= 2**51
    print("FFT external product at |v| up to 2^%.2f: max |delta| %d LSB overall, %d LSB where |v| >= 2^51 (%d values), "
          "%d of %d words differ" % (numpy.log2(float(numpy.abs(exact).max())), numpy.abs(delta).max(),
                                     numpy.abs(delta[big]).max(), int(big.sum()), int((delta != 0).sum()), delta.size))
    assert numpy.abs(delta).max() <= FFT_TOLERANCE_LSB
    # the reference's own FFT formulation (numpy fft, round -> int64 -> truncate) on the same inputs stays within
    # the same tolerance of the exact product: the deviation is the transforms' rounding, not the final conversion
    bkf = of.bk_from_coeffs(tgsw)
    ref = numpy.array([of.external_mul(accum[b].copy(), bkf, 0) for b in range(B)]).astype(numpy.int32).view(numpy.uint32)
    dref = (ref.astype(numpy.int64) - expect.astype(numpy.int64) + 2**31) % 2**32 - 2**31
    print("oracle_fft (numpy) on the same inputs: max |delta| %d LSB" % numpy.abs(dref).max())
    assert numpy.abs(dref).max() <= 4 * FFT_TOLERANCE_LSB


def test_fft_blind_rotate_step_at_the_largest_reachable_magnitude(thr, H):
    """The same edge through the path a gate takes: one blind-rotate STEP (`nufhe_blind_rotate`, one row, bara = 1024 so
    that (X^a - 1) ACC = -2 ACC) with accumulators chosen so that -2 ACC has the extreme digit patterns above.  The step's
    rounding is fused with the accumulation (fft_round_add_u32: |v| + 2^52, sign mask, one subtraction and one v_xad_u32 on
    the device) and the inverse transform hands it the imaginary parts negated -- another instruction sequence than
    nufhe_external_mul's, which rounds without accumulating -- so it gets its own check where |v| reaches 1.5 x 2^51."""
    rs = numpy.random.RandomState(78)
    B = 4
    lo_t = 0x100000000 - 0x80200000
    hi_t = (0xFFFFF000 - 0x80200000) & 0xFFFFFFFF
    T = numpy.empty((B, 2, 1024), numpy.int64)
    T[0] = lo_t
    T[1] = hi_t
    T[2:] = numpy.where(rs.randint(0, 2, size=(2, 2, 1024)).astype(bool), lo_t, hi_t)
    T += 2 * rs.randint(0, 0x800, size=T.shape)                     # low 12 bits do not reach a digit; T stays even
    # ACC with -2 ACC = T (mod 2^32): ACC = -(T / 2), plus 2^31 on a random half (both solutions)
    accum = ((-(T // 2)) + (rs.randint(0, 2, size=T.shape).astype(numpy.int64) << 31)) & 0xFFFFFFFF
    assert (((-2 * accum) & 0xFFFFFFFF) == (T & 0xFFFFFFFF)).all()
    tgsw = numpy.empty((1, 2, 2, 2, 1024), numpy.int32)
    tgsw[:] = -2**31
    tgsw[0, :, :, 1] = 2**31 - 1
    tgsw[0, 1, 1, :, ::2] = rs.choice([-2**31, 2**31 - 1], size=(2, 512))
    exact = _exact_external_mul((T & 0xFFFFFFFF).astype(numpy.uint32).view(numpy.int32), tgsw, 0)
    assert numpy.abs(exact).max() >= 1.5 * 2**51 and (numpy.abs(exact) >= 2**51).sum() > 1000
    expect = ((accum + exact) & 0xFFFFFFFF).astype(numpy.uint32)

    from nufhe_amd.bootstrap import NativeCloudKey
    native = NativeCloudKey(thr, 1, 'FFT')
    d_tgsw = H.dev(thr, tgsw)
    call("nufhe_bk_from_coeffs", native.handle, ptr(d_tgsw))
    acc = H.dev(thr, accum.astype(numpy.uint32).view(numpy.int32).reshape(B, 1, 2, 1024).copy())
    d_bara = H.dev(thr, numpy.full((B, 1), 1024, numpy.int32))
    call("nufhe_blind_rotate", thr.handle, native.handle, ptr(acc), ptr(d_bara), 1, 1, B)
    got = H.host(acc).reshape(B, 2, 1024).view(numpy.uint32)
    delta = (got.astype(numpy.int64) - expect.astype(numpy.int64) + 2**31) % 2**32 - 2**31
    big = numpy.abs(exact) >= 2**51
    print("FFT blind-rotate step at |v| up to 2^%.2f: max |delta| %d LSB overall, %d LSB where |v| >= 2^51 (%d values)"
          % (numpy.log2(float(numpy.abs(exact).max())), numpy.abs(delta).max(), numpy.abs(delta[big]).max(), int(big.sum())))
    assert numpy.abs(delta).max() <= FFT_TOLERANCE_LSB


@pytest.fixture(scope='module')
def fft_env(thr, H, orc, oracle_keys):
    import nufhe_amd
    from oracle import oracle_fft as of
    lwe_key, tlwe_key, ck = oracle_keys
    bkf = of.bk_from_coeffs(of.tgsw_coeffs_from_reference_bk(ck.bk))
    params = nufhe_amd.NuFHEParameters(transform_type='FFT')
    ckf = orc.CloudKeyArrays(bkf, ck.ks_a, ck.ks_b, ck.ks_cv)
    cloud_key = H.cloud_key_from_arrays(thr, ckf, params)
    secret_key = H.secret_key_from_array(thr, lwe_key, params)
    ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(456), thread=thr)
    return dict(ctx=ctx, vm=ctx.make_virtual_machine(cloud_key), sk=secret_key, ck=ck, lwe_key=lwe_key, params=params)


def test_fft_heterogeneous_gate_batch(fft_env, thr, H, orc):
    """nufhe_gate_batch on the FFT path: AND | MUX | XNOR of different sizes in one launch write exactly the words of the
    individual FFT gate calls (same kernels, same per-bit arithmetic) and stay within the FFT tolerance of the exact
    (NTT) oracle; sizes chosen so that the total crosses from the pair kernel's range into the wave kernel's."""
    import torch
    vm = fft_env['vm']; ck = fft_env['ck']; lwe_key = fft_env['lwe_key']; params = fft_env['params']
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    rng = orc.DeterministicRNG(515)

    def make(n):
        m = rng.uniform_bool((n,)).astype(bool)
        c = orc.encrypt(rng, lwe_key, m)
        return m, c, H.ciphertext_from_arrays(thr, c, params)
    for sizes in ((7, 3, 12), (2 * cus, cus, cus + 5)):
        x = [make(sizes[0]) for _ in range(2)]; y = [make(sizes[1]) for _ in range(3)]; z = [make(sizes[2]) for _ in range(2)]
        rx, ry, rz = vm.gate_batch([('gate_and', x[0][2], x[1][2]), ('gate_mux', y[0][2], y[1][2], y[2][2]),
                                    ('gate_xnor', z[0][2], z[1][2])])
        assert rx == vm.gate_and(x[0][2], x[1][2]) and ry == vm.gate_mux(y[0][2], y[1][2], y[2][2])
        assert rz == vm.gate_xnor(z[0][2], z[1][2])
        n = min(16, sizes[1])
        exp = orc.gate_mux(ck, *[tuple(v[:n] for v in q[1]) for q in y])
        ra, rb, rcv = H.ct_arrays(ry)
        da = (ra[:n].astype(numpy.int64) - exp[0].astype(numpy.int64) + 2**31) % 2**32 - 2**31
        assert abs(da).max() <= 16 and (rcv[:n] == exp[2]).all()
        assert (fft_env['ctx'].decrypt(fft_env['sk'], ry) == numpy.where(y[0][0], y[1][0], y[2][0])).all()
        assert (fft_env['ctx'].decrypt(fft_env['sk'], rz) == ~(z[0][0] ^ z[1][0])).all()


def test_config5_fft_gates_vs_exact_path(fft_env, thr, H, orc):
    """NAND and MUX with the FFT transform vs the exact (NTT) oracle: tolerance contract + record."""
    vm = fft_env['vm']; ctx = fft_env['ctx']; sk = fft_env['sk']; ck = fft_env['ck']; lwe_key = fft_env['lwe_key']
    rng = orc.DeterministicRNG(456)
    B = 64
    ms = [rng.uniform_bool((B,)).astype(bool) for _ in range(3)]
    cs = [orc.encrypt(rng, lwe_key, m) for m in ms]
    ds = [H.ciphertext_from_arrays(thr, c, fft_env['params']) for c in cs]
    for name, got, exp, truth in (
            ('nand', vm.gate_nand(ds[0], ds[1]), orc.gate('gate_nand', ck, cs[0], cs[1]), ~(ms[0] & ms[1])),
            ('mux', vm.gate_mux(ds[0], ds[1], ds[2]), orc.gate_mux(ck, cs[0], cs[1], cs[2]),
             numpy.where(ms[0], ms[1], ms[2]))):
        assert (ctx.decrypt(sk, got) == truth).all(), name
        ra, rb, rcv = H.ct_arrays(got)
        da = (ra.astype(numpy.int64) - exp[0].astype(numpy.int64) + 2**31) % 2**32 - 2**31
        db = (rb.astype(numpy.int64) - exp[1].astype(numpy.int64) + 2**31) % 2**32 - 2**31
        assert numpy.abs(da).max() <= FFT_TOLERANCE_LSB and numpy.abs(db).max() <= FFT_TOLERANCE_LSB, name
        print("FFT %s: output words differing from the exact path: %d of %d" % (
            name, int((da != 0).sum() + (db != 0).sum()), da.size + db.size))


def test_config5_fft_context_end_to_end_4096(thr):
    """transform_type='FFT' through the public API: GPU key generation, 4096-bit NAND, every
    decrypted bit equals the truth table; the FFT key encrypts the same TGSW samples as the NTT key
    generated from the same seed (key generation is transform independent)."""
    import nufhe_amd
    ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(123), thread=thr)
    sk, cloud = ctx.make_key_pair(transform_type='FFT')
    vm = ctx.make_virtual_machine(cloud)
    rs = numpy.random.RandomState(9)
    m1 = rs.randint(0, 2, size=4096).astype(bool); m2 = rs.randint(0, 2, size=4096).astype(bool)
    c1 = ctx.encrypt(sk, m1); c2 = ctx.encrypt(sk, m2)
    r = vm.gate_nand(c1, c2)
    assert (ctx.decrypt(sk, r) == ~(m1 & m2)).all()
    # same seed, NTT parameters: identical ciphertext out of the two transforms?  (recorded)
    ctx2 = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(123), thread=thr)
    sk2, cloud2 = ctx2.make_key_pair(transform_type='NTT')
    vm2 = ctx2.make_virtual_machine(cloud2)
    r2 = vm2.gate_nand(c1, c2)
    import gpu_helpers as H
    a1, b1, _ = H.ct_arrays(r); a2, b2, _ = H.ct_arrays(r2)
    d = (a1.astype(numpy.int64) - a2.astype(numpy.int64) + 2**31) % 2**32 - 2**31
    assert numpy.abs(d).max() <= FFT_TOLERANCE_LSB
    print("FFT vs NTT, 4096-bit NAND: differing words:", int((d != 0).sum()), "of", d.size)
    # serialization round trip of an FFT key
    cloud3 = ctx.load_cloud_key(cloud.dumps())
    assert cloud3 == cloud


def test_fft_small_batch_team_kernel(fft_env, thr, H, orc):
    """The four FFT kernels on a small batch -- one wave per bit, the 4-wave team (taken when the pair kernel is
    switched off), the 2-wave pair (team switch at 0), the 4-wave quad (default up to 1 x CUs bits: k_bootstrap_fft_quad, the
    MUX = 140 rotations puts its job boundary inside the launch) -- vs the exact (NTT) oracle: within the path's tolerance;
    observed: identical words."""
    from nufhe_amd import _lib
    vm = fft_env['vm']; ck = fft_env['ck']; lwe_key = fft_env['lwe_key']
    rng = orc.DeterministicRNG(31337)
    B = 70
    ms = [rng.uniform_bool((B,)).astype(bool) for _ in range(3)]
    cs = [orc.encrypt(rng, lwe_key, m) for m in ms]
    ds = [H.ciphertext_from_arrays(thr, c, fft_env['params']) for c in cs]
    exp = {'nand': orc.gate('gate_nand', ck, cs[0], cs[1]), 'mux': orc.gate_mux(ck, cs[0], cs[1], cs[2])}
    got = {}
    try:
        # (team limit, pair limit): wave-per-bit kernel; team kernel; pair kernel; default switches = quad kernel
        for limit in ((0, 0), (-1, 0), (0, -1), (-1, -1)):
            _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, limit[0])
            _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, limit[1])
            got[limit] = {'nand': H.ct_arrays(vm.gate_nand(ds[0], ds[1])), 'mux': H.ct_arrays(vm.gate_mux(ds[0], ds[1], ds[2]))}
    finally:
        _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, -1)
        _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, -1)
    for name in ('nand', 'mux'):
        for limit in ((0, 0), (-1, 0), (0, -1), (-1, -1)):
            ra, rb, rcv = got[limit][name]
            da = (ra.astype(numpy.int64) - exp[name][0].astype(numpy.int64) + 2**31) % 2**32 - 2**31
            db = (rb.astype(numpy.int64) - exp[name][1].astype(numpy.int64) + 2**31) % 2**32 - 2**31
            assert numpy.abs(da).max() <= FFT_TOLERANCE_LSB and numpy.abs(db).max() <= FFT_TOLERANCE_LSB, (name, limit)
            assert (rcv == exp[name][2]).all()
        same = all((x == y).all() for k in ((-1, 0), (0, -1), (-1, -1)) for x, y in zip(got[(0, 0)][name], got[k][name]))
        print("FFT %s: team, pair and quad kernels == wave kernel: %s; words differing from the exact path: %d" % (
            name, same, int((got[(-1, -1)][name][0] != exp[name][0]).sum())))


def test_fft_medium_batch_pair_kernel(fft_env, thr, H, orc):
    """The 2-waves-per-bit FFT kernel (bits <= 3 x CUs) with 1, 2 and 3 pairs per work-group (ragged last group)
    vs the wave-per-bit FFT kernel on the same ciphertexts and vs the exact (NTT) oracle on the first 24 bits: within
    the path's tolerance; observed: identical words."""
    import torch
    from nufhe_amd import _lib
    vm = fft_env['vm']; ck = fft_env['ck']; lwe_key = fft_env['lwe_key']
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    rng = orc.DeterministicRNG(909)
    sizes = [cus - 7, 2 * cus - 5, 3 * cus - 1]
    B = max(sizes)
    ms = [rng.uniform_bool((B,)).astype(bool) for _ in range(2)]
    cs = [orc.encrypt(rng, lwe_key, m) for m in ms]
    ds = [H.ciphertext_from_arrays(thr, c, fft_env['params']) for c in cs]
    exp = orc.gate('gate_nand', ck, tuple(x[:24] for x in cs[0]), tuple(x[:24] for x in cs[1]))

    def dev(x, y):
        return numpy.abs((x.astype(numpy.int64) - y.astype(numpy.int64) + 2**31) % 2**32 - 2**31).max()
    try:
        for size in sizes:
            a, b = ds[0][:size], ds[1][:size]
            _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, 0)
            _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, 0)
            wave = H.ct_arrays(vm.gate_nand(a, b))
            _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, -1)
            pair = H.ct_arrays(vm.gate_nand(a, b))
            assert dev(wave[0], pair[0]) <= FFT_TOLERANCE_LSB and dev(wave[1], pair[1]) <= FFT_TOLERANCE_LSB, size
            assert (wave[2] == pair[2]).all()
            assert dev(pair[0][:24], exp[0]) <= FFT_TOLERANCE_LSB and dev(pair[1][:24], exp[1]) <= FFT_TOLERANCE_LSB
            print("FFT pair vs wave kernel, %d bits: differing words: %d" % (size, int((wave[0] != pair[0]).sum() + (wave[1] != pair[1]).sum())))
    finally:
        _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, -1)
        _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, -1)


def test_fft_mux_job_boundary_inside_a_multi_pair_group(fft_env, thr, H, orc):
    """FFT MUX on 301 bits = 602 bootstraps: pair kernel with 3 pairs per work-group, one group straddling the two
    blind rotations; vs the wave kernel (tolerance; observed identical)."""
    from nufhe_amd import _lib
    vm = fft_env['vm']; lwe_key = fft_env['lwe_key']
    rng = orc.DeterministicRNG(607)
    B = 301
    ms = [rng.uniform_bool((B,)).astype(bool) for _ in range(3)]
    ds = [H.ciphertext_from_arrays(thr, orc.encrypt(rng, lwe_key, m), fft_env['params']) for m in ms]
    try:
        pair = H.ct_arrays(vm.gate_mux(ds[0], ds[1], ds[2]))
        _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, 0)
        _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, 0)
        wave = H.ct_arrays(vm.gate_mux(ds[0], ds[1], ds[2]))
    finally:
        _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, -1)
        _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, -1)
    for x, y in zip(pair[:2], wave[:2]):
        d = numpy.abs((x.astype(numpy.int64) - y.astype(numpy.int64) + 2**31) % 2**32 - 2**31).max()
        assert d <= FFT_TOLERANCE_LSB
    assert (pair[2] == wave[2]).all()


def test_fft_ragged_large_batch_head_and_tail(fft_env, thr, H, orc):
    """8 x CUs + 5 bits: whole rounds on the one-wave kernel + a 5-bit tail on the quad kernel (a second launch); 9 x CUs + 1:
    tail on the pair kernel; 10 x CUs + 1: one launch.  Against the one-wave kernel alone (both switches at 0) on every
    word: within the path's tolerance; observed: identical."""
    import torch
    from nufhe_amd import _lib
    vm = fft_env['vm']; lwe_key = fft_env['lwe_key']
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    rng = orc.DeterministicRNG(2024)
    B = 10 * cus + 1
    ms = [rng.uniform_bool((B,)).astype(bool) for _ in range(2)]
    ds = [H.ciphertext_from_arrays(thr, orc.encrypt(rng, lwe_key, m), fft_env['params']) for m in ms]

    def dev(x, y):
        return numpy.abs((x.astype(numpy.int64) - y.astype(numpy.int64) + 2**31) % 2**32 - 2**31).max()
    try:
        _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, 0)
        _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, 0)
        wave = H.ct_arrays(vm.gate_nand(ds[0], ds[1]))
    finally:
        _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, -1)
        _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, -1)
    for size in (8 * cus + 5, 9 * cus + 1, 10 * cus + 1):
        got = H.ct_arrays(vm.gate_nand(ds[0][:size], ds[1][:size]))
        assert dev(got[0], wave[0][:size]) <= FFT_TOLERANCE_LSB and dev(got[1], wave[1][:size]) <= FFT_TOLERANCE_LSB, size
        assert (got[2] == wave[2][:size]).all()
        print("FFT ragged batch of %d bits vs the one-wave kernel: differing words: %d" % (
            size, int((got[0] != wave[0][:size]).sum() + (got[1] != wave[1][:size]).sum())))
