"""
The exact-FFT engine for NTT-parameter keys (nufhe_cloudkey_set_engine, csrc/blind_rotate_xfft.h, kernels_xfft.hip) on
the GPU, through the C ABI.  Its claim is bit-identity with the reference's NTT path (nufhe/tgsw_cpu.py:82-106,
bootstrap.py:96-142, gates.py:81-121 / 600-664) for EVERY key made of int32 torus polynomials and EVERY input, so the
checks are equalities of all output words: against the reference-made gate golden, against the oracle, against the
native prime-field kernels, and against exact integer convolutions on the all-extreme inputs where the plain FFT path
(tests/test_gpu_fft.py) is 2 LSB off.  The 4096-bit every-word run against the oracle lives in test_gpu_gates.py
(test_config2_config3_every_word_vs_oracle[exact-fft]).
"""
import ctypes
import os

import numpy
import pytest

pytestmark = pytest.mark.gpu

import make_golden_gate
from test_emu_xfft import _exact_external_product

P = 2**64 - 2**32 + 1


@pytest.fixture(scope='module')
def thr():
    from nufhe_amd.device import DeviceThread
    return DeviceThread(0)


@pytest.fixture(scope='module')
def H():
    import gpu_helpers
    return gpu_helpers


def _native_key(thr, orc, tgsw, engine):
    """device key from int32 TGSW coefficient polynomials [n,2,2,2,1024] through the reference-format upload"""
    from nufhe_amd.bootstrap import NativeCloudKey
    from nufhe_amd import _lib
    native = NativeCloudKey(thr, tgsw.shape[0])
    bk = numpy.ascontiguousarray(orc.tlwe_transform_samples(tgsw), numpy.uint64)
    _lib.call("nufhe_bk_upload_reference", native.handle, bk.ctypes.data_as(ctypes.c_void_p))
    native.set_engine(engine)
    return native, bk


def test_external_product_random_and_extreme_inputs(thr, H, orc):
    from nufhe_amd import _lib
    from nufhe_amd.device import ptr
    rs = numpy.random.RandomState(21)
    n = 3
    tmin = numpy.uint32((0 - (2**31 + 2**21)) % 2**32).view(numpy.int32)                     # both digits -512
    tmax = numpy.uint32(((1023 << 22) | (1023 << 12)) - (2**31 + 2**21)).view(numpy.int32)   # both digits +511
    cases = []
    # random full-range
    cases.append((rs.randint(-2**31, 2**31, size=(5, 2, 1024), dtype=numpy.int32),
                  rs.randint(-2**31, 2**31, size=(n, 2, 2, 2, 1024), dtype=numpy.int32)))
    # everything at its extreme, signs random / aligned (|v| up to 2^36 in each half, 2^52 in the integer result)
    cases.append((numpy.where(rs.rand(5, 2, 1024) < 0.5, tmax, tmin).astype(numpy.int32),
                  numpy.where(rs.rand(n, 2, 2, 2, 1024) < 0.5, numpy.int32(2**31 - 1), numpy.int32(-2**31)).astype(numpy.int32)))
    cases.append((numpy.full((5, 2, 1024), tmin, numpy.int32), numpy.full((n, 2, 2, 2, 1024), numpy.int32(-2**31), numpy.int32)))
    ka = numpy.int32(0x7FFF8000); kb = numpy.uint32(0x80007FFF).view(numpy.int32)         # both halves at their limits
    cases.append((numpy.where(rs.rand(5, 2, 1024) < 0.5, tmax, tmin).astype(numpy.int32),
                  numpy.where(rs.rand(n, 2, 2, 2, 1024) < 0.5, ka, kb).astype(numpy.int32)))
    for accum, tgsw in cases:
        native, bk = _native_key(thr, orc, tgsw, 'exact-fft')
        for row in (0, n - 1):
            acc = H.dev(thr, accum)
            _lib.call("nufhe_external_mul", thr.handle, native.handle, ptr(acc), row, accum.shape[0])
            got = H.host(acc)
            assert (got == orc.tgsw_external_mul(accum, bk, row)).all()
            assert (got[1] == _exact_external_product(accum[1], tgsw[row])).all()
        native.destroy()


def test_blind_rotate_ragged_batch_vs_oracle_and_native(thr, H, orc):
    from nufhe_amd import _lib
    from nufhe_amd.device import ptr
    rs = numpy.random.RandomState(34)
    B, n = 21, 9
    tgsw = rs.randint(-2**31, 2**31, size=(n, 2, 2, 2, 1024), dtype=numpy.int32)
    acc0 = rs.randint(-2**31, 2**31, size=(B, 2, 1024), dtype=numpy.int32)
    bara = rs.randint(0, 2048, size=(B, n)).astype(numpy.int32)
    bara[:, 4] = 0                                            # skipped iterations
    native, bk = _native_key(thr, orc, tgsw, 'exact-fft')
    out = {}
    for engine in ('exact-fft', 'native'):
        native.set_engine(engine)
        assert native.get_engine() == engine
        acc = H.dev(thr, acc0); d_bara = H.dev(thr, bara)
        _lib.call("nufhe_blind_rotate", thr.handle, native.handle, ptr(acc), ptr(d_bara), n, n, B)
        out[engine] = H.host(acc)
    assert (out['exact-fft'] == orc.blind_rotate(acc0, bk, bara)).all()
    assert (out['exact-fft'] == out['native']).all()


@pytest.fixture(scope='module')
def env(orc, oracle_keys, thr, H):
    import nufhe_amd
    lwe_key, tlwe_key, ck = oracle_keys
    cloud_key = H.cloud_key_from_arrays(thr, ck)
    cloud_key.set_engine('exact-fft')
    secret_key = H.secret_key_from_array(thr, lwe_key)
    ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(456), thread=thr)
    vm = ctx.make_virtual_machine(cloud_key)
    return dict(ctx=ctx, vm=vm, ck=ck, lwe_key=lwe_key, cloud_key=cloud_key, secret_key=secret_key)


def test_reference_made_gate_golden(thr, H, orc):
    """NAND and MUX composed from the reference's own CPU functions at full size (tests/golden/make_golden_gate.py)"""
    import nufhe_amd
    g = numpy.load(os.path.join(os.path.dirname(make_golden_gate.__file__), 'reference_gate_outputs.npz'))
    lwe_key, tlwe_key, ck, cts, ms = make_golden_gate.gate_inputs()
    cloud_key = H.cloud_key_from_arrays(thr, ck).set_engine('exact-fft')
    assert cloud_key.engine == 'exact-fft'
    ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(1), thread=thr)
    vm = ctx.make_virtual_machine(cloud_key)
    d = [H.ciphertext_from_arrays(thr, ct) for ct in cts]
    ra, rb, rcv = H.ct_arrays(vm.gate_nand(d[0], d[1]))
    assert (ra == g['nand_a']).all() and (rb == g['nand_b']).all() and (rcv == g['nand_cv']).all()
    ra, rb, rcv = H.ct_arrays(vm.gate_mux(d[0], d[1], d[2]))
    assert (ra == g['mux_a']).all() and (rb == g['mux_b']).all() and (rcv == g['mux_cv']).all()


def test_gates_every_word_vs_oracle_ragged_batch(env, H, thr, orc):
    vm = env['vm']; ctx = env['ctx']; sk = env['secret_key']; ck = env['ck']
    rng = numpy.random.RandomState(77)
    B = 70                                                   # not a multiple of anything
    ms = [rng.randint(0, 2, size=(B,)).astype(bool) for _ in range(3)]
    cs = [ctx.encrypt(sk, m) for m in ms]
    host = [H.ct_arrays(c) for c in cs]
    for name in ('gate_nand', 'gate_xor', 'gate_andyn'):
        ra, rb, rcv = H.ct_arrays(getattr(vm, name)(cs[0], cs[1]))
        exp = orc.gate(name, ck, host[0][:2], host[1][:2])
        assert (ra == exp[0]).all() and (rb == exp[1]).all() and (rcv == exp[2]).all(), name
    rm = vm.gate_mux(cs[0], cs[1], cs[2])
    assert (ctx.decrypt(sk, rm) == numpy.where(ms[0], ms[1], ms[2])).all()
    ma, mb, mcv = H.ct_arrays(rm)
    expm = orc.gate_mux(ck, *[tuple(h[:2]) for h in host])
    assert (ma == expm[0]).all() and (mb == expm[1]).all() and (mcv == expm[2]).all()


@pytest.mark.parametrize("B", [1, 257, 511, 2048, 4096, 4100])
def test_engines_agree_word_for_word_at_every_batch_size(env, H, B):
    """the native engine switches kernel families with the batch size (team8 / pair / wave + tail), the exact engine
    between four waves per bit (up to 2 x CUs bits: 1, 257 = one and two teams per work-group with a ragged last group;
    MUX on 257 and 511 bits = 514 / 1022 rotations: the wave kernel, and the job boundary inside a two-team group at 1 bit)
    and one wave per bit: the same ciphertexts everywhere"""
    vm = env['vm']; ctx = env['ctx']; sk = env['secret_key']; key = env['cloud_key']
    rng = numpy.random.RandomState(B)
    ms = [rng.randint(0, 2, size=(B,)).astype(bool) for _ in range(3)]
    cs = [ctx.encrypt(sk, m) for m in ms]
    out = {}
    try:
        for engine in ('exact-fft', 'native'):
            key.set_engine(engine)
            out[engine] = (H.ct_arrays(vm.gate_nand(cs[0], cs[1])), H.ct_arrays(vm.gate_mux(cs[0], cs[1], cs[2])))
    finally:
        key.set_engine('exact-fft')
    for g in range(2):
        for x, y in zip(out['exact-fft'][g], out['native'][g]):
            assert int((x != y).sum()) == 0
    assert (ctx.decrypt(sk, vm.gate_nand(cs[0], cs[1])) == ~(ms[0] & ms[1])).all()


def test_quad_kernel_equals_wave_kernel_and_oracle(env, H, thr, orc):
    """k_bootstrap_xfft_quad (four waves per bit, brxq_*) against k_bootstrap_xfft (team switch at 0) on every word of a
    2 x CUs - 5 bit NAND (two teams per work-group, ragged last group) and a 301-bit MUX (602 rotations > 2 x CUs on the
    wave kernel either way; 100 bits = 200 rotations: job boundary inside the quad launch); oracle on the first bits"""
    import torch
    from nufhe_amd import _lib
    vm = env['vm']; ctx = env['ctx']; sk = env['secret_key']; ck = env['ck']
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    rng = numpy.random.RandomState(4242)
    B = 2 * cus - 5
    ms = [rng.randint(0, 2, size=(B,)).astype(bool) for _ in range(3)]
    cs = [ctx.encrypt(sk, m) for m in ms]
    quad = (H.ct_arrays(vm.gate_nand(cs[0], cs[1])), H.ct_arrays(vm.gate_mux(cs[0][:100], cs[1][:100], cs[2][:100])))
    try:
        _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, 0)
        wave = (H.ct_arrays(vm.gate_nand(cs[0], cs[1])), H.ct_arrays(vm.gate_mux(cs[0][:100], cs[1][:100], cs[2][:100])))
    finally:
        _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, -1)
    for g in range(2):
        for x, y in zip(quad[g], wave[g]):
            assert int((x != y).sum()) == 0
    host = [H.ct_arrays(c[:6]) for c in cs]
    exp = orc.gate('gate_nand', ck, host[0][:2], host[1][:2])
    assert all((q[:6] == e).all() for q, e in zip(quad[0], exp))
    expm = orc.gate_mux(ck, *[tuple(h[:2]) for h in host])
    assert all((q[:6] == e).all() for q, e in zip(quad[1], expm))
    assert (ctx.decrypt(sk, vm.gate_nand(cs[0], cs[1])) == ~(ms[0] & ms[1])).all()


def test_exact_engine_switch_points_plus_minus_one_bit(env, H):
    """the exact engine's dispatch: one quad team per work-group up to 1 x CUs rotations, two up to 2 x CUs, two launches
    of the quad kernel (head of 2 x CUs + tail) up to 4 x CUs, then one wave per bit -- whole rounds of 8 x CUs bits, a tail of
    up to 2 x CUs bits back on the quad kernel; tlwe_mask_size = 1.  NAND at each
    switch point and one bit beyond == the native engine, every word; a MUX whose second job starts inside the head launch."""
    import torch
    vm = env['vm']; ctx = env['ctx']; sk = env['secret_key']; key = env['cloud_key']
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    rng = numpy.random.RandomState(99)
    B = 10 * cus + 1
    ms = [rng.randint(0, 2, size=(B,)).astype(bool) for _ in range(3)]
    cs = [ctx.encrypt(sk, m) for m in ms]
    M = cus + 44                                        # MUX: 2 M rotations, job boundary at M < 2 x CUs (inside the head)
    try:
        key.set_engine('native')
        ref = H.ct_arrays(vm.gate_nand(cs[0], cs[1]))
        ref_mux = H.ct_arrays(vm.gate_mux(cs[0][:M], cs[1][:M], cs[2][:M]))
        key.set_engine('exact-fft')
        got_mux = H.ct_arrays(vm.gate_mux(cs[0][:M], cs[1][:M], cs[2][:M]))
        assert all((g == r).all() for g, r in zip(got_mux, ref_mux))
        for size in (cus, cus + 1, 2 * cus, 2 * cus + 1, 3 * cus - 1, 4 * cus, 4 * cus + 1, 8 * cus + 1, 10 * cus, 10 * cus + 1):
            got = H.ct_arrays(vm.gate_nand(cs[0][:size], cs[1][:size]))
            assert all((g == r[:size]).all() for g, r in zip(got, ref)), size
    finally:
        key.set_engine('exact-fft')


def test_gate_batch_and_stepwise_driver_on_the_exact_engine(env, H, orc):
    import nufhe_amd
    vm = env['vm']; ctx = env['ctx']; sk = env['secret_key']; ck = env['ck']
    rng = numpy.random.RandomState(5)
    ms = [rng.randint(0, 2, size=(n,)).astype(bool) for n in (3, 3, 9, 9, 9)]
    cs = [ctx.encrypt(sk, m) for m in ms]
    r0, r1 = vm.gate_batch([('gate_or', cs[0], cs[1]), ('gate_mux', cs[2], cs[3], cs[4])])
    assert r0 == vm.gate_or(cs[0], cs[1]) and r1 == vm.gate_mux(cs[2], cs[3], cs[4])
    h = [H.ct_arrays(c) for c in cs]
    exp = orc.gate('gate_or', ck, h[0][:2], h[1][:2])
    ra, rb, rcv = H.ct_arrays(r0)
    assert (ra == exp[0]).all() and (rb == exp[1]).all() and (rcv == exp[2]).all()
    # a batch whose 300 + 2 x 200 + 2100 rotations go through every launch of the engine's dispatch: rounds on the one-wave
    # kernel + a quad tail (2800 = 2048 + 752 -> 752 alone: two quad launches) -- every word equal to the single gates
    big = [ctx.encrypt(sk, rng.randint(0, 2, size=(n,)).astype(bool)) for n in (300, 300, 200, 200, 200, 2100, 2100)]
    b0, b1, b2 = vm.gate_batch([('gate_xnor', big[0], big[1]), ('gate_mux', big[2], big[3], big[4]), ('gate_nand', big[5], big[6])])
    assert b0 == vm.gate_xnor(big[0], big[1]) and b1 == vm.gate_mux(big[2], big[3], big[4]) and b2 == vm.gate_nand(big[5], big[6])
    # the reference's multi-kernel mode (bootstrap.py:96-142) drives nufhe_blind_rotate step by step: same engine, same words
    pp = nufhe_amd.PerformanceParameters(env['cloud_key'].params, single_kernel_bootstrap=False)
    vm2 = ctx.make_virtual_machine(env['cloud_key'], perf_params=pp)
    assert vm2.gate_or(cs[0], cs[1]) == r0


def test_engine_refusals_and_key_changes(thr, H, orc, oracle_keys):
    from nufhe_amd import _lib
    from nufhe_amd.bootstrap import NativeCloudKey
    from nufhe_amd.device import ptr
    # not an NTT key (tlwe_mask_size 2 is served: tests/test_mask_size_2.py)
    native = NativeCloudKey(thr, 500, transform_type='FFT')
    with pytest.raises(ValueError, match='exact-FFT engine serves NTT keys'):
        native.set_engine('exact-fft')
    native.destroy()
    with pytest.raises(ValueError, match='unknown engine'):
        NativeCloudKey(thr, 4).set_engine('fast')
    # a synthetic key of random field elements is not the transform of int32 polynomials: refused at first use, by name
    rs = numpy.random.RandomState(1)
    native = NativeCloudKey(thr, 4)
    bk = rs.randint(0, P, size=(4, 2, 2, 2, 1024), dtype=numpy.uint64)
    _lib.call("nufhe_bk_upload_reference", native.handle, bk.ctypes.data_as(ctypes.c_void_p))
    native.set_engine('exact-fft')
    acc = H.dev(thr, rs.randint(-2**31, 2**31, size=(1, 2, 1024), dtype=numpy.int32))
    with pytest.raises(ValueError, match='not the transform of int32 polynomials'):
        _lib.call("nufhe_external_mul", thr.handle, native.handle, ptr(acc), 0, 1)
    # a new upload into the same holder replaces the derived image
    accum = rs.randint(-2**31, 2**31, size=(2, 2, 1024), dtype=numpy.int32)
    for seed in (2, 3):
        tgsw = numpy.random.RandomState(seed).randint(-2**31, 2**31, size=(4, 2, 2, 2, 1024), dtype=numpy.int32)
        bk = numpy.ascontiguousarray(orc.tlwe_transform_samples(tgsw), numpy.uint64)
        _lib.call("nufhe_bk_upload_reference", native.handle, bk.ctypes.data_as(ctypes.c_void_p))
        acc = H.dev(thr, accum)
        _lib.call("nufhe_external_mul", thr.handle, native.handle, ptr(acc), 1, 2)
        assert (H.host(acc) == orc.tgsw_external_mul(accum, bk, 1)).all()
    native.destroy()


def test_device_image_round_trip_keeps_the_engine_working(env, thr, H):
    from nufhe_amd.api_low_level import NuFHECloudKey
    ctx = env['ctx']; sk = env['secret_key']; vm = env['vm']
    params, image = env['cloud_key'].device_image()
    clone = NuFHECloudKey.from_device_image(thr, params, image).set_engine('exact-fft')
    vm2 = ctx.make_virtual_machine(clone)
    rng = numpy.random.RandomState(8)
    m1 = rng.randint(0, 2, size=(16,)).astype(bool); m2 = rng.randint(0, 2, size=(16,)).astype(bool)
    c1 = ctx.encrypt(sk, m1); c2 = ctx.encrypt(sk, m2)
    assert vm2.gate_nand(c1, c2) == vm.gate_nand(c1, c2)
