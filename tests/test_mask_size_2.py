"""
tlwe_mask_size = 2 (reference: test/test_gates.py:96-100 runs every gate with
NuFHEParameters(tlwe_mask_size=2), where the reference falls back to its multi-kernel blind rotate,
blind_rotate.py:53-58).  CPU part: the oracle vs the reference's k = 2 outputs
(tests/golden/make_golden_k2.py).  GPU part: the fused gfx950 kernel k_bootstrap<2>, the k = 2
keyswitch (input size 2048) and key generation through the C ABI / the public API.
"""

import ctypes
import os

import numpy
import pytest

import golden_inputs as gi

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def golden_k2():
    return numpy.load(os.path.join(HERE, 'golden', 'reference_outputs_k2.npz'))


@pytest.fixture(scope='module')
def k2_inputs(orc):
    params = orc.Params(mask_size=2)
    lwe_key, tlwe_key, ck = orc.make_key_pair(orc.DeterministicRNG(123), params)
    rng = orc.DeterministicRNG(456)
    ms = [numpy.array([True, False]), numpy.array([True, True])]
    cts = [orc.encrypt(rng, lwe_key, m, params) for m in ms]
    return params, lwe_key, tlwe_key, ck, cts, ms


# ---- CPU: oracle pinned to the reference for k = 2 ----------------------------------------------

def test_oracle_k2_pieces_vs_reference(orc, golden_k2):
    for full in (False, True):
        accum, bk, row = gi.extmul_inputs_k2(full_range=full)
        assert (orc.tgsw_external_mul(accum, bk, row) == golden_k2['tgsw_extmul_k2' + ('_full' if full else '')]).all()
    key, n1, n2 = gi.encrypt_zero_inputs_k2()
    ra, rcv = orc.tlwe_encrypt_zero(key, n1, n2, 9e-9)
    assert (ra == golden_k2['encrypt_zero_k2_a']).all() and (rcv == golden_k2['encrypt_zero_k2_cv']).all()


def test_oracle_k2_nand_vs_reference_full_size(orc, golden_k2, k2_inputs):
    params, lwe_key, tlwe_key, ck, cts, ms = k2_inputs
    assert ck.bk.shape == (500, 3, 2, 3, 1024) and ck.ks_a.shape == (2048, 8, 4, 500)
    MU = 2**29
    ta = (-cts[0][0] - cts[1][0]).astype(numpy.int32)
    tb = (numpy.int32(MU) - cts[0][1] - cts[1][1]).astype(numpy.int32)
    ea, eb = orc.bootstrap_extract(ck.bk, ta, tb, MU)
    assert (ea == golden_k2['nand_k2_ext_a']).all() and (eb == golden_k2['nand_k2_ext_b']).all()
    ra, rb, rcv = orc.gate('gate_nand', ck, cts[0], cts[1])
    assert (ra == golden_k2['nand_k2_a']).all() and (rb == golden_k2['nand_k2_b']).all()
    assert (rcv == golden_k2['nand_k2_cv']).all()
    assert (orc.decrypt(lwe_key, (ra, rb, rcv)) == ~(ms[0] & ms[1])).all()


def test_parameters_k2_host_side():
    import nufhe_amd
    p = nufhe_amd.NuFHEParameters(tlwe_mask_size=2)
    assert p.tgsw_params.tlwe_params.mask_size == 2
    assert p.tgsw_params.tlwe_params.extracted_lweparams.size == 2048
    pf = nufhe_amd.NuFHEParameters(transform_type='FFT', tlwe_mask_size=2)
    assert pf.tgsw_params.tlwe_params.mask_size == 2 and pf.tgsw_params.tlwe_params.transform_type == 'FFT'
    with pytest.raises(NotImplementedError):
        nufhe_amd.NuFHEParameters(tlwe_mask_size=3)


# ---- GPU ------------------------------------------------------------------------------------------

@pytest.fixture(scope='module')
def thr():
    from nufhe_amd.device import DeviceThread
    return DeviceThread(0)


@pytest.fixture(scope='module')
def H():
    import gpu_helpers
    return gpu_helpers


def _call(name, *args):
    from nufhe_amd import _lib
    _lib.call(name, *args)


@pytest.mark.gpu
def test_gpu_k2_external_mul_and_encrypt_zero_vs_reference(thr, H, golden_k2):
    from nufhe_amd.bootstrap import NativeCloudKey
    from nufhe_amd.device import ptr
    for full in (False, True):
        accum, bk, row = gi.extmul_inputs_k2(full_range=full)
        native = NativeCloudKey(thr, bk.shape[0], 'NTT', 2)
        _call("nufhe_bk_upload_reference", native.handle, bk.ctypes.data_as(ctypes.c_void_p))
        back = numpy.empty_like(bk)
        _call("nufhe_bk_download_reference", native.handle, back.ctypes.data_as(ctypes.c_void_p))
        assert (back == bk).all()
        acc = H.dev(thr, accum)
        _call("nufhe_external_mul", thr.handle, native.handle, ptr(acc), row, accum.size // (3 * 1024))
        assert (H.host(acc) == golden_k2['tgsw_extmul_k2' + ('_full' if full else '')]).all()
    key, n1, n2 = gi.encrypt_zero_inputs_k2()
    dk, d1, d2 = H.dev(thr, key), H.dev(thr, n1), H.dev(thr, n2)
    res = thr.array(n2.shape[:-1] + (3, 1024), numpy.int32)
    _call("nufhe_tlwe_encrypt_zero", thr.handle, ptr(res), ptr(dk), ptr(d1), ptr(d2), 6, 2)
    assert (H.host(res) == golden_k2['encrypt_zero_k2_a']).all()


@pytest.fixture(scope='module')
def k2_env(thr, H, orc, k2_inputs):
    import nufhe_amd
    oparams, lwe_key, tlwe_key, ck, cts, ms = k2_inputs
    params = nufhe_amd.NuFHEParameters(tlwe_mask_size=2)
    cloud_key = H.cloud_key_from_arrays(thr, ck, params)
    secret_key = H.secret_key_from_array(thr, lwe_key, params)
    ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(789), thread=thr)
    return dict(ctx=ctx, vm=ctx.make_virtual_machine(cloud_key), sk=secret_key, params=params, ck=ck,
                lwe_key=lwe_key, oparams=oparams, cloud_key=cloud_key)


@pytest.mark.gpu
def test_gpu_k2_nand_vs_reference_golden(k2_env, thr, H, golden_k2, k2_inputs):
    """The reference's own k = 2 NAND outputs (extracted LWE(2048) and keyswitched LWE(500))."""
    from nufhe_amd.bootstrap import bootstrap
    from nufhe_amd.lwe import LweSampleArray
    _, lwe_key, _, ck, cts, ms = k2_inputs
    vm = k2_env['vm']; params = k2_env['params']
    d = [H.ciphertext_from_arrays(thr, c, params) for c in cts]
    r = vm.gate_nand(d[0], d[1])
    ra, rb, rcv = H.ct_arrays(r)
    assert (ra == golden_k2['nand_k2_a']).all() and (rb == golden_k2['nand_k2_b']).all()
    assert (rcv == golden_k2['nand_k2_cv']).all()
    # the bootstrap without the keyswitch: LWE(2048)
    MU = 2**29
    ta = (-cts[0][0] - cts[1][0]).astype(numpy.int32)
    tb = (numpy.int32(MU) - cts[0][1] - cts[1][1]).astype(numpy.int32)
    src = LweSampleArray(params.in_out_params, H.dev(thr, ta), H.dev(thr, tb), H.dev(thr, numpy.zeros(2, numpy.float32)))
    ext = LweSampleArray.empty(thr, params.tgsw_params.tlwe_params.extracted_lweparams, (2,))
    ck_dev = k2_env['cloud_key']
    bootstrap(thr, ext, ck_dev.bootstrap_key, ck_dev.keyswitch_key, MU, src, no_keyswitch=True)
    ea, eb, _ = H.ct_arrays(ext)
    assert ea.shape == (2, 2048)
    assert (ea == golden_k2['nand_k2_ext_a']).all() and (eb == golden_k2['nand_k2_ext_b']).all()


@pytest.mark.gpu
def test_gpu_k2_gates_vs_oracle_ragged(k2_env, thr, H, orc):
    """NAND/XOR/MUX on a ragged 37-bit batch: bit-exact vs the oracle run with the same k = 2 keys."""
    vm = k2_env['vm']; ck = k2_env['ck']; lwe_key = k2_env['lwe_key']; params = k2_env['params']
    rng = orc.DeterministicRNG(4567)
    B = 37
    ms = [rng.uniform_bool((B,)).astype(bool) for _ in range(3)]
    cs = [orc.encrypt(rng, lwe_key, m, k2_env['oparams']) for m in ms]
    ds = [H.ciphertext_from_arrays(thr, c, params) for c in cs]
    for name, got, exp, truth in (
            ('nand', vm.gate_nand(ds[0], ds[1]), orc.gate('gate_nand', ck, cs[0], cs[1]), ~(ms[0] & ms[1])),
            ('xor', vm.gate_xor(ds[0], ds[1]), orc.gate('gate_xor', ck, cs[0], cs[1]), ms[0] ^ ms[1]),
            ('mux', vm.gate_mux(ds[0], ds[1], ds[2]), orc.gate_mux(ck, cs[0], cs[1], cs[2]),
             numpy.where(ms[0], ms[1], ms[2]))):
        ra, rb, rcv = H.ct_arrays(got)
        assert (ra == exp[0]).all() and (rb == exp[1]).all() and (rcv == exp[2]).all(), name
        assert (k2_env['ctx'].decrypt(k2_env['sk'], got) == truth).all(), name


@pytest.mark.gpu
def test_gpu_k2_heterogeneous_gate_batch(k2_env, thr, H, orc):
    """nufhe_gate_batch with tlwe_mask_size = 2 (extracted samples of 2048 coefficients through the MUX fold and the
    keyswitch): XOR | MUX | NAND of different sizes in one launch, every word vs the k = 2 oracle."""
    vm = k2_env['vm']; ck = k2_env['ck']; lwe_key = k2_env['lwe_key']; params = k2_env['params']
    rng = orc.DeterministicRNG(8910)

    def make(n):
        m = rng.uniform_bool((n,)).astype(bool)
        c = orc.encrypt(rng, lwe_key, m, k2_env['oparams'])
        return m, c, H.ciphertext_from_arrays(thr, c, params)
    x = [make(9) for _ in range(2)]; y = [make(4) for _ in range(3)]; z = [make(21) for _ in range(2)]
    rx, ry, rz = vm.gate_batch([('gate_xor', x[0][2], x[1][2]), ('gate_mux', y[0][2], y[1][2], y[2][2]),
                                ('gate_nand', z[0][2], z[1][2])])
    for got, exp in ((rx, orc.gate('gate_xor', ck, x[0][1], x[1][1])), (ry, orc.gate_mux(ck, y[0][1], y[1][1], y[2][1])),
                     (rz, orc.gate('gate_nand', ck, z[0][1], z[1][1]))):
        ra, rb, rcv = H.ct_arrays(got)
        assert (ra == exp[0]).all() and (rb == exp[1]).all() and (rcv == exp[2]).all()
    assert (k2_env['ctx'].decrypt(k2_env['sk'], ry) == numpy.where(y[0][0], y[1][0], y[2][0])).all()


@pytest.mark.gpu
def test_gpu_k2_large_batch_kernels(k2_env, thr, H, orc):
    """k = 2, NTT, batches beyond the team kernel.  Default: the ring kernel (3 waves per bit, 2 teams per work-group,
    ragged last group).  With the pair switch at 0: between 4 x and 6 x CUs bits the 6-waves-per-CU build
    (k_bootstrap<2>), otherwise the one-wave-per-SIMD build (k_bootstrap_k2_roomy).  A 5 x CUs batch through the ring
    kernel and through k_bootstrap<2> must equal the same ciphertexts processed as two halves by the roomy kernel on
    every word, and the oracle on its first and last 8 bits."""
    import torch
    from nufhe_amd import _lib
    vm = k2_env['vm']; ck = k2_env['ck']; lwe_key = k2_env['lwe_key']; params = k2_env['params']
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    rng = orc.DeterministicRNG(31)
    B = 5 * cus - 3
    ms = [rng.uniform_bool((B,)).astype(bool) for _ in range(2)]
    cs = [orc.encrypt(rng, lwe_key, m, k2_env['oparams']) for m in ms]
    ds = [H.ciphertext_from_arrays(thr, c, params) for c in cs]
    h = 3 * cus        # 3 x CUs and 2 x CUs - 3 bits: one round of the roomy kernel each
    try:
        ring = H.ct_arrays(vm.gate_nand(ds[0], ds[1]))
        _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, 0)
        whole = H.ct_arrays(vm.gate_nand(ds[0], ds[1]))
        first = H.ct_arrays(vm.gate_nand(ds[0][:h], ds[1][:h]))
        second = H.ct_arrays(vm.gate_nand(ds[0][h:], ds[1][h:]))
    finally:
        _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, -1)
    for r, w, a, b in zip(ring, whole, first, second):
        assert (w[:h] == a).all() and (w[h:] == b).all() and (r == w).all()
    idx = numpy.r_[0:8, B - 8:B]
    exp = orc.gate('gate_nand', ck, tuple(x[idx] for x in cs[0]), tuple(x[idx] for x in cs[1]))
    for w, e in zip(ring, exp):
        assert (w[idx] == e).all()


@pytest.mark.gpu
def test_gpu_k2_mux_job_boundary_inside_a_two_team_group(k2_env, thr, H, orc):
    """k = 2 MUX on 301 bits = 602 bootstraps through the ring kernel with 2 teams per work-group: one group holds the
    last bit of the first blind rotation and the first of the second.  Bit-identical to the wave kernels."""
    from nufhe_amd import _lib
    vm = k2_env['vm']; lwe_key = k2_env['lwe_key']; params = k2_env['params']
    rng = orc.DeterministicRNG(608)
    B = 301
    ms = [rng.uniform_bool((B,)).astype(bool) for _ in range(3)]
    ds = [H.ciphertext_from_arrays(thr, orc.encrypt(rng, lwe_key, m, k2_env['oparams']), params) for m in ms]
    try:
        ring = H.ct_arrays(vm.gate_mux(ds[0], ds[1], ds[2]))
        _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, 0)
        _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, 0)
        wave = H.ct_arrays(vm.gate_mux(ds[0], ds[1], ds[2]))
    finally:
        _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, -1)
        _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, -1)
    for x, y in zip(ring, wave):
        assert (x == y).all()
    assert (k2_env['ctx'].decrypt(k2_env['sk'], vm.gate_mux(ds[0], ds[1], ds[2])) == numpy.where(ms[0], ms[1], ms[2])).all()


@pytest.mark.gpu
def test_gpu_k2_keyswitch_on_the_matrix_cores(k2_env, thr, H, orc):
    """k = 2 (keyswitch input LWE(2048)): matrix-core keyswitch == LDS-window keyswitch == oracle on a ragged batch."""
    from nufhe_amd import _lib
    vm = k2_env['vm']; ck = k2_env['ck']; lwe_key = k2_env['lwe_key']; params = k2_env['params']
    rng = orc.DeterministicRNG(4243)
    B = 70
    ms = [rng.uniform_bool((B,)).astype(bool) for _ in range(2)]
    cs = [orc.encrypt(rng, lwe_key, m, k2_env['oparams']) for m in ms]
    ds = [H.ciphertext_from_arrays(thr, c, params) for c in cs]
    got = {}
    try:
        for mode in (0, 2):
            _lib.call("nufhe_ctx_set_keyswitch_mfma", thr.handle, mode)
            got[mode] = H.ct_arrays(vm.gate_nand(ds[0], ds[1]))
    finally:
        _lib.call("nufhe_ctx_set_keyswitch_mfma", thr.handle, 1)
    exp = orc.gate('gate_nand', ck, cs[0], cs[1])
    for x, y, e in zip(got[0], got[2], exp):
        assert (x == y).all() and (y == e).all()


@pytest.mark.gpu
def test_gpu_k2_exact_engine_external_mul_vs_oracle_and_native(thr, H, orc):
    """tlwe_mask_size = 2 on the exact-FFT engine (brxk_* in csrc/blind_rotate_xfft.h: six digit polynomials per sum,
    bound 0.055): random full-range and all-extreme inputs against the oracle's prime-field result and the native kernels."""
    from nufhe_amd.bootstrap import NativeCloudKey
    from nufhe_amd.device import ptr
    # (the reference-made external-product fixture multiplies by random field elements, which no int32 key transforms
    # to: the engine refuses it by name -- tests/test_gpu_xfft.py; the reference-made NAND golden is in the next test)
    rs = numpy.random.RandomState(76)
    accum = rs.randint(-2**31, 2**31, size=(7, 3, 1024), dtype=numpy.int32)
    tgsw = rs.randint(-2**31, 2**31, size=(3, 3, 2, 3, 1024), dtype=numpy.int32)
    bk = numpy.ascontiguousarray(orc.tlwe_transform_samples(tgsw), numpy.uint64)
    native = NativeCloudKey(thr, 3, 'NTT', 2)
    _call("nufhe_bk_upload_reference", native.handle, bk.ctypes.data_as(ctypes.c_void_p))
    native.set_engine('exact-fft')
    for row in (0, 2):
        acc = H.dev(thr, accum)
        _call("nufhe_external_mul", thr.handle, native.handle, ptr(acc), row, 7)
        assert (H.host(acc) == orc.tgsw_external_mul(accum, bk, row)).all()
    native.destroy()
    # every digit at +-512, every key coefficient at +-2^31: the sums reach their largest magnitude
    rs = numpy.random.RandomState(77)
    tmin = numpy.uint32((0 - (2**31 + 2**21)) % 2**32).view(numpy.int32)
    tmax = numpy.uint32(((1023 << 22) | (1023 << 12)) - (2**31 + 2**21)).view(numpy.int32)
    accum = numpy.where(rs.rand(5, 3, 1024) < 0.5, tmax, tmin).astype(numpy.int32)
    tgsw = numpy.where(rs.rand(2, 3, 2, 3, 1024) < 0.5, numpy.int32(2**31 - 1), numpy.int32(-2**31)).astype(numpy.int32)
    bk = numpy.ascontiguousarray(orc.tlwe_transform_samples(tgsw), numpy.uint64)
    out = {}
    for engine in ('native', 'exact-fft'):
        native = NativeCloudKey(thr, 2, 'NTT', 2)
        _call("nufhe_bk_upload_reference", native.handle, bk.ctypes.data_as(ctypes.c_void_p))
        native.set_engine(engine)
        acc = H.dev(thr, accum)
        _call("nufhe_external_mul", thr.handle, native.handle, ptr(acc), 1, 5)
        out[engine] = H.host(acc)
        native.destroy()
    assert (out['exact-fft'] == out['native']).all()
    assert (out['native'] == orc.tgsw_external_mul(accum, bk, 1)).all()


@pytest.mark.gpu
def test_gpu_k2_exact_engine_gates_vs_reference_oracle_and_native(k2_env, thr, H, orc, golden_k2, k2_inputs):
    """k = 2 gates on the exact engine: the reference's NAND golden, NAND / XOR / MUX on a ragged batch vs the oracle,
    a heterogeneous gate batch, the step-by-step driver, and a 5 x CUs - 3 bit batch against the native ring kernel on
    every word."""
    import torch
    import nufhe_amd
    _, lwe_key, _, ck, cts, _ = k2_inputs
    vm = k2_env['vm']; params = k2_env['params']; key = k2_env['cloud_key']
    rng = orc.DeterministicRNG(1213)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    key.set_engine('exact-fft')
    try:
        assert key.engine == 'exact-fft'
        d = [H.ciphertext_from_arrays(thr, c, params) for c in cts]
        ra, rb, rcv = H.ct_arrays(vm.gate_nand(d[0], d[1]))
        assert (ra == golden_k2['nand_k2_a']).all() and (rb == golden_k2['nand_k2_b']).all()
        assert (rcv == golden_k2['nand_k2_cv']).all()
        B = 37
        ms = [rng.uniform_bool((B,)).astype(bool) for _ in range(3)]
        cs = [orc.encrypt(rng, lwe_key, m, k2_env['oparams']) for m in ms]
        ds = [H.ciphertext_from_arrays(thr, c, params) for c in cs]
        for name, got, exp, truth in (
                ('nand', vm.gate_nand(ds[0], ds[1]), orc.gate('gate_nand', ck, cs[0], cs[1]), ~(ms[0] & ms[1])),
                ('xor', vm.gate_xor(ds[0], ds[1]), orc.gate('gate_xor', ck, cs[0], cs[1]), ms[0] ^ ms[1]),
                ('mux', vm.gate_mux(ds[0], ds[1], ds[2]), orc.gate_mux(ck, cs[0], cs[1], cs[2]),
                 numpy.where(ms[0], ms[1], ms[2]))):
            ga = H.ct_arrays(got)
            assert all((g == e).all() for g, e in zip(ga, exp)), name
            assert (k2_env['ctx'].decrypt(k2_env['sk'], got) == truth).all(), name
        rx, ry = vm.gate_batch([('gate_xor', ds[0][:9], ds[1][:9]), ('gate_mux', ds[0][9:13], ds[1][9:13], ds[2][9:13])])
        ex = orc.gate('gate_xor', ck, tuple(x[:9] for x in cs[0]), tuple(x[:9] for x in cs[1]))
        ey = orc.gate_mux(ck, *(tuple(x[9:13] for x in c) for c in cs))
        assert all((g == e).all() for g, e in zip(H.ct_arrays(rx), ex))
        assert all((g == e).all() for g, e in zip(H.ct_arrays(ry), ey))
        pp = nufhe_amd.PerformanceParameters(params, single_kernel_bootstrap=False)
        vm2 = k2_env['ctx'].make_virtual_machine(key, perf_params=pp)
        assert all((g == e).all() for g, e in zip(H.ct_arrays(vm2.gate_xor(ds[0][:9], ds[1][:9])), ex))
        # up to 1 x CUs bits: six waves per bit (k_bootstrap_xfft_hex_k2); the same gate with the team switch at 0 runs
        # the one-wave kernel: equal on every word (the 37-bit gates above ran on the six-wave kernel)
        from nufhe_amd import _lib
        six = H.ct_arrays(vm.gate_mux(ds[0], ds[1], ds[2]))
        try:
            _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, 0)
            one = H.ct_arrays(vm.gate_mux(ds[0], ds[1], ds[2]))
        finally:
            _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, -1)
        assert all((x == y).all() for x, y in zip(six, one))
        Bl = 5 * cus - 3
        ml = [rng.uniform_bool((Bl,)).astype(bool) for _ in range(2)]
        dl = [H.ciphertext_from_arrays(thr, orc.encrypt(rng, lwe_key, m, k2_env['oparams']), params) for m in ml]
        exact = H.ct_arrays(vm.gate_nand(dl[0], dl[1]))
        key.set_engine('native')
        native = H.ct_arrays(vm.gate_nand(dl[0], dl[1]))
        assert all((x == y).all() for x, y in zip(exact, native))
    finally:
        key.set_engine('native')


@pytest.mark.gpu
def test_gpu_k2_context_end_to_end(thr):
    """Public API with tlwe_mask_size=2 (test/test_gates.py:96-100): GPU key generation, all binary
    gates + MUX on 64 bits, serialization round trip of the k = 2 cloud key."""
    import nufhe_amd
    ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(31), thread=thr)
    sk, cloud = ctx.make_key_pair(tlwe_mask_size=2)
    vm = ctx.make_virtual_machine(cloud)
    rs = numpy.random.RandomState(2)
    m = [rs.randint(0, 2, size=64).astype(bool) for _ in range(3)]
    c = [ctx.encrypt(sk, x) for x in m]
    assert (ctx.decrypt(sk, vm.gate_nand(c[0], c[1])) == ~(m[0] & m[1])).all()
    assert (ctx.decrypt(sk, vm.gate_or(c[0], c[1])) == (m[0] | m[1])).all()
    assert (ctx.decrypt(sk, vm.gate_xnor(c[0], c[1])) == ~(m[0] ^ m[1])).all()
    assert (ctx.decrypt(sk, vm.gate_mux(c[0], c[1], c[2])) == numpy.where(m[0], m[1], m[2])).all()
    cloud2 = ctx.load_cloud_key(cloud.dumps())
    assert cloud2 == cloud
    vm2 = ctx.make_virtual_machine(cloud2)
    assert (ctx.decrypt(sk, vm2.gate_and(c[0], c[1])) == (m[0] & m[1])).all()


@pytest.mark.gpu
def test_gpu_k2_small_batch_team_kernel_equals_wave_kernel(k2_env, thr, H, orc):
    """k = 2 on a small batch: the 3-waves-per-bit kernels (k_bootstrap_team_k2 with its partial-sum buffer, batches <=
    CUs bits by default; k_bootstrap_ring_k2 without it) give exactly the ciphertexts of the wave-per-bit kernel and of
    the oracle (NAND and MUX, 70 bits)."""
    from nufhe_amd import _lib
    vm = k2_env['vm']; ck = k2_env['ck']; lwe_key = k2_env['lwe_key']; params = k2_env['params']
    rng = orc.DeterministicRNG(777)
    B = 70
    ms = [rng.uniform_bool((B,)).astype(bool) for _ in range(3)]
    cs = [orc.encrypt(rng, lwe_key, m, k2_env['oparams']) for m in ms]
    ds = [H.ciphertext_from_arrays(thr, c, params) for c in cs]
    exp = {'nand': orc.gate('gate_nand', ck, cs[0], cs[1]), 'mux': orc.gate_mux(ck, cs[0], cs[1], cs[2])}
    got = {}
    try:
        # (team limit, pair limit): (0, 0) wave-per-bit kernel; (0, -1) ring kernel; (-1, -1) default = team kernel
        for limit in ((0, 0), (0, -1), (-1, -1)):
            _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, limit[0])
            _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, limit[1])
            got[limit] = {'nand': H.ct_arrays(vm.gate_nand(ds[0], ds[1])),
                          'mux': H.ct_arrays(vm.gate_mux(ds[0], ds[1], ds[2]))}
    finally:
        _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, -1)
        _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, -1)
    for name in ('nand', 'mux'):
        for limit in ((0, 0), (0, -1), (-1, -1)):
            ra, rb, rcv = got[limit][name]
            assert (ra == exp[name][0]).all() and (rb == exp[name][1]).all() and (rcv == exp[name][2]).all(), (name, limit)


# ---- tlwe_mask_size = 2 with the FFT transform (test/test_gates.py:88-100: every (transform, k) pair) ----

FFT_TOLERANCE_LSB = 2**4      # per output word of a gate vs the exact path (tests/test_gpu_fft.py)


@pytest.mark.gpu
def test_gpu_k2_fft_external_mul_vs_exact(thr, H, golden_k2, orc):
    """TGswTransformedExternalMul with k = 2 and the FFT transform.  The reference-made k = 2 golden uses a
    key that is random in the NTT domain (not the transform of int32 polynomials), so the FFT case takes
    int32 TGSW samples, transforms them both ways, and compares with the C oracle's exact product -- the
    function the k = 2 goldens pin (test_oracle_k2_external_mul...).  +-1 LSB on <= 1e-4 of the
    coefficients is the contract; observed: 0."""
    from nufhe_amd.bootstrap import NativeCloudKey
    from nufhe_amd.device import ptr
    from oracle import oracle_fft as of
    rs = numpy.random.RandomState(2718)
    tgsw = rs.randint(-2**31, 2**31, size=(3, 3, 2, 3, 1024), dtype=numpy.int32)
    bk_ntt = orc.tlwe_transform_samples(tgsw)
    bkf = numpy.ascontiguousarray(of.fft_forward(tgsw), numpy.complex128)
    native = NativeCloudKey(thr, 3, 'FFT', 2)
    _call("nufhe_bk_upload_reference", native.handle, bkf.ctypes.data_as(ctypes.c_void_p))
    back = numpy.empty_like(bkf)
    _call("nufhe_bk_download_reference", native.handle, back.ctypes.data_as(ctypes.c_void_p))
    assert (back == bkf).all()
    # the key transformed on the device from the coefficients gives the same polynomials (to fp64 accuracy)
    native2 = NativeCloudKey(thr, 3, 'FFT', 2)
    _call("nufhe_bk_from_coeffs", native2.handle, ptr(H.dev(thr, tgsw)))
    back2 = numpy.empty_like(bkf)
    _call("nufhe_bk_download_reference", native2.handle, back2.ctypes.data_as(ctypes.c_void_p))
    assert numpy.abs(back2 - bkf).max() / numpy.abs(bkf).max() < 1e-12
    for full in (False, True):
        accum = (rs.randint(-2**31, 2**31, size=(7, 3, 1024), dtype=numpy.int32) if full
                 else rs.randint(-1000, 1000, size=(7, 3, 1024)).astype(numpy.int32))
        for row in (0, 2):
            ref = orc.tgsw_external_mul(accum, bk_ntt, row)
            for nat in (native, native2):
                acc = H.dev(thr, accum)
                _call("nufhe_external_mul", thr.handle, nat.handle, ptr(acc), row, 7)
                delta = H.host(acc).astype(numpy.int64) - ref.astype(numpy.int64)
                assert numpy.abs(delta).max() <= 1 and (delta != 0).mean() <= 1e-4, (full, row)


@pytest.fixture(scope='module')
def k2_fft_env(thr, H, orc, k2_inputs):
    import nufhe_amd
    from oracle import oracle_fft as of
    oparams, lwe_key, tlwe_key, ck, cts, ms = k2_inputs
    params = nufhe_amd.NuFHEParameters(transform_type='FFT', tlwe_mask_size=2)
    bkf = of.fft_forward(of.tgsw_coeffs_from_reference_bk(ck.bk))
    ckf = orc.CloudKeyArrays(bkf, ck.ks_a, ck.ks_b, ck.ks_cv)
    cloud_key = H.cloud_key_from_arrays(thr, ckf, params)
    secret_key = H.secret_key_from_array(thr, lwe_key, params)
    ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(790), thread=thr)
    return dict(ctx=ctx, vm=ctx.make_virtual_machine(cloud_key), sk=secret_key, params=params, ck=ck,
                lwe_key=lwe_key, oparams=oparams, cloud_key=cloud_key)


@pytest.mark.gpu
def test_gpu_k2_fft_gates_vs_exact_path(k2_fft_env, thr, H, orc, golden_k2, k2_inputs):
    """k = 2, FFT: the reference-made NAND golden (B = 2) and NAND / XOR / MUX on a ragged 37-bit batch vs
    the exact (NTT) oracle: decrypted bits identical, every word within the FFT tolerance (observed: 0)."""
    env = k2_fft_env
    vm = env['vm']; params = env['params']; ck = env['ck']; lwe_key = env['lwe_key']
    _, _, _, _, cts, ms = k2_inputs
    d = [H.ciphertext_from_arrays(thr, c, params) for c in cts]
    ra, rb, rcv = H.ct_arrays(vm.gate_nand(d[0], d[1]))

    def dev(x, y):
        return numpy.abs((x.astype(numpy.int64) - y.astype(numpy.int64) + 2**31) % 2**32 - 2**31).max()
    assert dev(ra, golden_k2['nand_k2_a']) <= FFT_TOLERANCE_LSB and dev(rb, golden_k2['nand_k2_b']) <= FFT_TOLERANCE_LSB
    assert (rcv == golden_k2['nand_k2_cv']).all()
    rng = orc.DeterministicRNG(4568)
    B = 37
    msg = [rng.uniform_bool((B,)).astype(bool) for _ in range(3)]
    cs = [orc.encrypt(rng, lwe_key, m, env['oparams']) for m in msg]
    ds = [H.ciphertext_from_arrays(thr, c, params) for c in cs]
    from nufhe_amd import _lib
    exp_all = {'nand': orc.gate('gate_nand', ck, cs[0], cs[1]), 'xor': orc.gate('gate_xor', ck, cs[0], cs[1]),
               'mux': orc.gate_mux(ck, cs[0], cs[1], cs[2])}
    truth_all = {'nand': ~(msg[0] & msg[1]), 'xor': msg[0] ^ msg[1], 'mux': numpy.where(msg[0], msg[1], msg[2])}
    try:
        # (team, pair) limits: (0, 0) one wave per bit (k_bootstrap_fft_k2); (0, -1) ring kernel
        # (k_bootstrap_fft_ring_k2); (-1, -1) default switches = team kernel for 37 / 74 bits (k_bootstrap_fft_team_k2)
        for limit in ((0, 0), (0, -1), (-1, -1)):
            _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, limit[0])
            _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, limit[1])
            for name, got in (('nand', vm.gate_nand(ds[0], ds[1])), ('xor', vm.gate_xor(ds[0], ds[1])),
                              ('mux', vm.gate_mux(ds[0], ds[1], ds[2]))):
                exp = exp_all[name]
                ga, gb, gcv = H.ct_arrays(got)
                assert dev(ga, exp[0]) <= FFT_TOLERANCE_LSB and dev(gb, exp[1]) <= FFT_TOLERANCE_LSB, (name, limit)
                assert (gcv == exp[2]).all(), (name, limit)
                assert (env['ctx'].decrypt(env['sk'], got) == truth_all[name]).all(), (name, limit)
                print("k=2 FFT %s (team, pair limits %s): words differing from the exact path: %d"
                      % (name, limit, int((ga != exp[0]).sum() + (gb != exp[1]).sum())))
    finally:
        _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, -1)
        _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, -1)


@pytest.mark.gpu
def test_gpu_k2_fft_ring_kernel_two_teams(k2_fft_env, thr, H, orc):
    """k = 2, FFT, a batch of 2 x CUs - 5 bits: the ring kernel (2 teams per work-group, ragged last group) vs the
    one-wave-per-bit kernel on every word and vs the exact oracle on the first 16 bits (tolerance; observed: 0)."""
    import torch
    from nufhe_amd import _lib
    env = k2_fft_env
    vm = env['vm']; params = env['params']; ck = env['ck']; lwe_key = env['lwe_key']
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    rng = orc.DeterministicRNG(515)
    B = 2 * cus - 5
    msg = [rng.uniform_bool((B,)).astype(bool) for _ in range(2)]
    cs = [orc.encrypt(rng, lwe_key, m, env['oparams']) for m in msg]
    ds = [H.ciphertext_from_arrays(thr, c, params) for c in cs]

    def dev(x, y):
        return numpy.abs((x.astype(numpy.int64) - y.astype(numpy.int64) + 2**31) % 2**32 - 2**31).max()
    try:
        ring = H.ct_arrays(vm.gate_nand(ds[0], ds[1]))
        _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, 0)
        _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, 0)
        wave = H.ct_arrays(vm.gate_nand(ds[0], ds[1]))
    finally:
        _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, -1)
        _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, -1)
    assert dev(ring[0], wave[0]) <= FFT_TOLERANCE_LSB and dev(ring[1], wave[1]) <= FFT_TOLERANCE_LSB
    assert (ring[2] == wave[2]).all()
    exp = orc.gate('gate_nand', ck, tuple(x[:16] for x in cs[0]), tuple(x[:16] for x in cs[1]))
    assert dev(ring[0][:16], exp[0]) <= FFT_TOLERANCE_LSB and dev(ring[1][:16], exp[1]) <= FFT_TOLERANCE_LSB
    print("k=2 FFT ring vs wave kernel, %d bits: differing words: %d" % (B, int((ring[0] != wave[0]).sum() + (ring[1] != wave[1]).sum())))


@pytest.mark.gpu
def test_gpu_k2_fft_six_wave_kernel(k2_fft_env, thr, H, orc):
    """k = 2, FFT, up to 1 x CUs bits: six waves per bit (k_bootstrap_fft_hex_k2, brfq_* with K = 2) -- NAND on 37 bits and a MUX
    on 21 bits (two jobs in one launch) vs the team kernel (team switch at -1 with the ring switch at 0 does NOT reach it: the
    comparison runs the one-wave kernel, both switches at 0) on every word, and vs the exact oracle (tolerance; observed: 0)."""
    from nufhe_amd import _lib
    env = k2_fft_env
    vm = env['vm']; params = env['params']; ck = env['ck']; lwe_key = env['lwe_key']
    rng = orc.DeterministicRNG(616)
    B = 37
    msg = [rng.uniform_bool((B,)).astype(bool) for _ in range(3)]
    cs = [orc.encrypt(rng, lwe_key, m, env['oparams']) for m in msg]
    ds = [H.ciphertext_from_arrays(thr, c, params) for c in cs]

    def dev(x, y):
        return numpy.abs((x.astype(numpy.int64) - y.astype(numpy.int64) + 2**31) % 2**32 - 2**31).max()

    def gates():
        return (H.ct_arrays(vm.gate_nand(ds[0], ds[1])), H.ct_arrays(vm.gate_mux(ds[0][:21], ds[1][:21], ds[2][:21])))
    try:
        six = gates()
        _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, 0)
        _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, 0)
        wave = gates()
    finally:
        _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, -1)
        _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, -1)
    exp = (orc.gate('gate_nand', ck, cs[0], cs[1]), orc.gate_mux(ck, *(tuple(x[:21] for x in c) for c in cs)))
    for g in range(2):
        assert dev(six[g][0], wave[g][0]) <= FFT_TOLERANCE_LSB and dev(six[g][1], wave[g][1]) <= FFT_TOLERANCE_LSB
        assert (six[g][2] == wave[g][2]).all()
        assert dev(six[g][0], exp[g][0]) <= FFT_TOLERANCE_LSB and dev(six[g][1], exp[g][1]) <= FFT_TOLERANCE_LSB
        print("k=2 FFT six-wave kernel vs one-wave kernel, gate %d: differing words: %d; vs the exact path: %d" % (
            g, int((six[g][0] != wave[g][0]).sum() + (six[g][1] != wave[g][1]).sum()), int((six[g][0] != exp[g][0]).sum())))


@pytest.mark.gpu
def test_gpu_k2_fft_context_end_to_end(thr):
    """Public API: Context.make_key_pair(transform_type='FFT', tlwe_mask_size=2) on the GPU, gates on 200
    bits, serialization round trip of the key."""
    import nufhe_amd
    ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(32), thread=thr)
    sk, cloud = ctx.make_key_pair(transform_type='FFT', tlwe_mask_size=2)
    vm = ctx.make_virtual_machine(cloud)
    rs = numpy.random.RandomState(3)
    m = [rs.randint(0, 2, size=200).astype(bool) for _ in range(3)]
    c = [ctx.encrypt(sk, x) for x in m]
    assert (ctx.decrypt(sk, vm.gate_nand(c[0], c[1])) == ~(m[0] & m[1])).all()
    assert (ctx.decrypt(sk, vm.gate_xor(c[0], c[1])) == (m[0] ^ m[1])).all()
    assert (ctx.decrypt(sk, vm.gate_mux(c[0], c[1], c[2])) == numpy.where(m[0], m[1], m[2])).all()
    cloud2 = ctx.load_cloud_key(cloud.dumps())
    assert cloud2 == cloud
