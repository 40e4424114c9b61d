"""
GPU parity tests, whole-gate granularity (BASELINE configs 1-3), through the nufhe-style API:
 * ciphertext-level equality (a, b, current_variances) with golden outputs of the REFERENCE's own
   CPU functions at full size (B=2 NAND/MUX) and with the oracle on larger batches,
 * decrypt == truth table for all 14 gates (test/test_gates.py:178-245 of the reference),
 * GPU key generation == oracle key generation from the same seed (RNG order),
 * the 4096-bit configurations through size-independent properties.
"""

import os

import numpy
import pytest

pytestmark = pytest.mark.gpu

import make_golden_gate


@pytest.fixture(scope='module')
def env(orc, oracle_keys):
    import gpu_helpers as H
    from nufhe_amd.device import DeviceThread
    import nufhe_amd
    thr = DeviceThread(0)
    lwe_key, tlwe_key, ck = oracle_keys
    cloud_key = H.cloud_key_from_arrays(thr, ck)
    secret_key = H.secret_key_from_array(thr, lwe_key)
    ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(456), thread=thr)
    vm = ctx.make_virtual_machine(cloud_key)
    return dict(H=H, thr=thr, ctx=ctx, vm=vm, ck=ck, lwe_key=lwe_key, cloud_key=cloud_key, secret_key=secret_key)


def test_nand_mux_vs_reference_full_size_golden(env, orc):
    H = env['H']; thr = env['thr']; vm = env['vm']
    g = numpy.load(os.path.join(os.path.dirname(make_golden_gate.__file__), 'reference_gate_outputs.npz'))
    lwe_key, tlwe_key, ck, cts, ms = make_golden_gate.gate_inputs()
    d = [H.ciphertext_from_arrays(thr, ct) for ct in cts]
    ra, rb, rcv = H.ct_arrays(vm.gate_nand(d[0], d[1]))
    assert (ra == g['nand_a']).all() and (rb == g['nand_b']).all() and (rcv == g['nand_cv']).all()
    ra, rb, rcv = H.ct_arrays(vm.gate_mux(d[0], d[1], d[2]))
    assert (ra == g['mux_a']).all() and (rb == g['mux_b']).all() and (rcv == g['mux_cv']).all()
    # bootstrap without keyswitch -> extracted sample (gates.py:633-655 building block)
    from nufhe_amd.bootstrap import bootstrap
    from nufhe_amd import lwe as L
    params = env['cloud_key'].params
    MU = 2**29
    tmp = L.LweSampleArray.empty(thr, params.in_out_params, (2,))
    L.lwe_noiseless_trivial_constant(thr, tmp, MU)
    L.lwe_sub_to(thr, tmp, d[0]); L.lwe_sub_to(thr, tmp, d[1])
    ext = L.LweSampleArray.empty(thr, params.tgsw_params.tlwe_params.extracted_lweparams, (2,))
    bootstrap(thr, ext, env['cloud_key'].bootstrap_key, env['cloud_key'].keyswitch_key, MU, tmp, no_keyswitch=True)
    assert (H.host(ext.a) == g['nand_ext_a']).all() and (H.host(ext.b) == g['nand_ext_b']).all()


def test_config1_nand_32_bits_vs_oracle(env, orc):
    """BASELINE config 1/2 shape at 32 bits: every output word equals the CPU path."""
    H = env['H']; thr = env['thr']; vm = env['vm']; ck = env['ck']; lwe_key = env['lwe_key']
    rng = orc.DeterministicRNG(456)
    m1 = rng.uniform_bool((32,)).astype(bool); m2 = rng.uniform_bool((32,)).astype(bool)
    c1 = orc.encrypt(rng, lwe_key, m1); c2 = orc.encrypt(rng, lwe_key, m2)
    exp = orc.gate('gate_nand', ck, c1, c2)
    r = vm.gate_nand(H.ciphertext_from_arrays(thr, c1), H.ciphertext_from_arrays(thr, c2))
    ra, rb, rcv = H.ct_arrays(r)
    assert (ra == exp[0]).all() and (rb == exp[1]).all() and (rcv == exp[2]).all()
    assert (env['ctx'].decrypt(env['secret_key'], r) == ~(m1 & m2)).all()


def test_mux_vs_oracle_odd_batch(env, orc):
    H = env['H']; thr = env['thr']; vm = env['vm']; ck = env['ck']; lwe_key = env['lwe_key']
    rng = orc.DeterministicRNG(789)
    B = 19
    ms = [rng.uniform_bool((B,)).astype(bool) for _ in range(3)]
    cs = [orc.encrypt(rng, lwe_key, m) for m in ms]
    exp = orc.gate_mux(ck, *cs)
    r = vm.gate_mux(*[H.ciphertext_from_arrays(thr, c) for c in cs])
    ra, rb, rcv = H.ct_arrays(r)
    assert (ra == exp[0]).all() and (rb == exp[1]).all() and (rcv == exp[2]).all()
    assert (env['ctx'].decrypt(env['secret_key'], r) == numpy.where(ms[0], ms[1], ms[2])).all()


TRUTH = {
    'gate_nand': lambda a, b: ~(a & b), 'gate_or': lambda a, b: a | b, 'gate_and': lambda a, b: a & b,
    'gate_nor': lambda a, b: ~(a | b), 'gate_xor': lambda a, b: a ^ b, 'gate_xnor': lambda a, b: ~(a ^ b),
    'gate_andny': lambda a, b: ~a & b, 'gate_andyn': lambda a, b: a & ~b,
    'gate_orny': lambda a, b: ~a | b, 'gate_oryn': lambda a, b: a | ~b,
}


def test_all_gates_truth_tables_and_parity(env, orc):
    """test_gates.py:178-245 of the reference (32 bits), plus ciphertext equality with the oracle."""
    H = env['H']; thr = env['thr']; vm = env['vm']; ctx = env['ctx']; sk = env['secret_key']
    ck = env['ck']
    rng = numpy.random.RandomState(5)
    m1 = rng.randint(0, 2, size=(4, 8)).astype(bool); m2 = rng.randint(0, 2, size=(4, 8)).astype(bool)
    c1 = ctx.encrypt(sk, m1); c2 = ctx.encrypt(sk, m2)
    for name, fn in TRUTH.items():
        r = getattr(vm, name)(c1, c2)
        assert r.shape == (4, 8)
        assert (ctx.decrypt(sk, r) == fn(m1, m2)).all(), name
        exp = orc.gate(name, ck, H.ct_arrays(c1)[:2], H.ct_arrays(c2)[:2])
        ra, rb, rcv = H.ct_arrays(r)
        assert (ra == exp[0]).all() and (rb == exp[1]).all(), name
    assert (ctx.decrypt(sk, vm.gate_not(c1)) == ~m1).all()
    assert (ctx.decrypt(sk, vm.gate_copy(c1)) == m1).all()
    assert (ctx.decrypt(sk, vm.gate_constant(m2)) == m2).all()


def test_broadcast_and_views(env, orc):
    """Shape derivation / broadcasting (test_api_high_level.py:135-172) and strided views as
    inputs and output (test_gates.py:514-559)."""
    vm = env['vm']; ctx = env['ctx']; sk = env['secret_key']
    rng = numpy.random.RandomState(6)
    m1 = rng.randint(0, 2, size=(3, 4)).astype(bool); m2 = rng.randint(0, 2, size=(4,)).astype(bool)
    c1 = ctx.encrypt(sk, m1); c2 = ctx.encrypt(sk, m2)
    r = vm.gate_and(c1, c2)
    assert r.shape == (3, 4) and (ctx.decrypt(sk, r) == (m1 & m2)).all()
    big = ctx.encrypt(sk, numpy.zeros((6, 4), bool))
    vm.gate_or(c1[:, ::2], c2[::2], dest=big[1::2, 1:3])
    dec = ctx.decrypt(sk, big)
    assert (dec[1::2, 1:3] == (m1[:, ::2] | m2[::2])).all()
    mask = numpy.ones((6, 4), bool); mask[1::2, 1:3] = False
    assert (dec[mask] == False).all()
    with pytest.raises(ValueError):
        vm.gate_and(c1, ctx.encrypt(sk, numpy.zeros((5,), bool)))


def test_gpu_keygen_matches_oracle_keygen(env, orc, oracle_keys):
    """Context.make_key_pair on the GPU == oracle.make_key_pair from the same seed: same RNG
    consumption order, same TLWE-encrypt-zero products, same transformed key (reference format)."""
    import nufhe_amd
    thr = env['thr']; H = env['H']
    lwe_key, tlwe_key, ck = oracle_keys
    ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(123), thread=thr)
    sk, cloud = ctx.make_key_pair()
    assert (H.host(sk.lwe_key.key) == lwe_key).all()
    assert (cloud.bootstrap_key.transformed_reference_format() == ck.bk).all()
    assert (cloud.keyswitch_key.lwe.a == ck.ks_a).all()
    assert (cloud.keyswitch_key.lwe.b == ck.ks_b).all()
    assert (cloud.keyswitch_key.lwe.current_variances == ck.ks_cv).all()
    # serialization round trip (test_api_high_level.py:58-110)
    cloud2 = ctx.load_cloud_key(cloud.dumps())
    assert cloud2 == cloud
    sk2 = ctx.load_secret_key(sk.dumps())
    assert sk2 == sk
    ct = ctx.encrypt(sk, [True, False, True])
    ct2 = ctx.load_ciphertext(ct.dumps())
    assert ct2 == ct
    vm = ctx.make_virtual_machine(cloud2)
    assert (ctx.decrypt(sk2, vm.gate_nand(ct2, ct2)) == [False, True, False]).all()


def test_config2_config3_full_size_properties(env, orc):
    """BASELINE configs 2 and 3 (4096 bits): decrypted outputs equal the truth tables for every bit;
    the first 32 ciphertexts equal the CPU path word for word; re-running gives identical output
    (determinism of the atomics-based keyswitch)."""
    H = env['H']; thr = env['thr']; vm = env['vm']; ctx = env['ctx']; sk = env['secret_key']; ck = env['ck']
    rng = numpy.random.RandomState(7)
    B = 4096
    ms = [rng.randint(0, 2, size=(B,)).astype(bool) for _ in range(3)]
    cs = [ctx.encrypt(sk, m) for m in ms]
    r = vm.gate_nand(cs[0], cs[1])
    assert (ctx.decrypt(sk, r) == ~(ms[0] & ms[1])).all()
    exp = orc.gate('gate_nand', ck, tuple(x[:32] for x in H.ct_arrays(cs[0])[:2]),
                   tuple(x[:32] for x in H.ct_arrays(cs[1])[:2]))
    ra, rb, rcv = H.ct_arrays(r)
    assert (ra[:32] == exp[0]).all() and (rb[:32] == exp[1]).all() and (rcv[:32] == exp[2]).all()
    r2 = vm.gate_nand(cs[0], cs[1])
    assert r2 == r
    rm = vm.gate_mux(cs[0], cs[1], cs[2])
    assert (ctx.decrypt(sk, rm) == numpy.where(ms[0], ms[1], ms[2])).all()
    expm = orc.gate_mux(ck, *[tuple(x[:16] for x in H.ct_arrays(c)[:2]) for c in cs])
    ma, mb, mcv = H.ct_arrays(rm)
    assert (ma[:16] == expm[0]).all() and (mb[:16] == expm[1]).all()


def test_edge_shapes_empty_scalar_and_odd(env, orc):
    """Empty batch, 0-dim ciphertexts and batch sizes that are not multiples of the wave/tile
    granularities (8 bits per work-group, 32 bits per keyswitch tile)."""
    H = env['H']; thr = env['thr']; vm = env['vm']; ctx = env['ctx']; sk = env['secret_key']; ck = env['ck']
    # empty
    e1 = ctx.encrypt(sk, numpy.zeros((0,), bool)); e2 = ctx.encrypt(sk, numpy.zeros((0,), bool))
    r = vm.gate_nand(e1, e2)
    assert r.shape == (0,) and ctx.decrypt(sk, r).shape == (0,)
    r = vm.gate_mux(e1, e2, e1)
    assert r.shape == (0,)
    # 0-dim
    s1 = ctx.encrypt(sk, numpy.array(True)); s2 = ctx.encrypt(sk, numpy.array(False))
    r = vm.gate_or(s1, s2)
    assert r.shape == () and bool(ctx.decrypt(sk, r)) is True
    # odd sizes, ciphertext-level parity with the oracle
    for B in (1, 9, 33):
        rng = numpy.random.RandomState(B)
        m1 = rng.randint(0, 2, size=B).astype(bool); m2 = rng.randint(0, 2, size=B).astype(bool)
        c1 = ctx.encrypt(sk, m1); c2 = ctx.encrypt(sk, m2)
        r = vm.gate_xor(c1, c2)
        exp = orc.gate('gate_xor', ck, H.ct_arrays(c1)[:2], H.ct_arrays(c2)[:2])
        ra, rb, rcv = H.ct_arrays(r)
        assert (ra == exp[0]).all() and (rb == exp[1]).all() and (rcv == exp[2]).all()
        assert (ctx.decrypt(sk, r) == (m1 ^ m2)).all()


def test_error_behaviour(env):
    """Shape / key errors surface as the reference's exception types."""
    import nufhe_amd
    from nufhe_amd import lwe as L
    vm = env['vm']; ctx = env['ctx']; sk = env['secret_key']; thr = env['thr']
    c1 = ctx.encrypt(sk, numpy.zeros((4,), bool)); c2 = ctx.encrypt(sk, numpy.zeros((3,), bool))
    with pytest.raises(ValueError):
        vm.gate_nand(c1, c2)                                    # gates.py:57-58
    with pytest.raises(ValueError):
        vm.gate_nand(c1, c1, dest=ctx.encrypt(sk, numpy.zeros((5,), bool)))   # gates.py:74-78
    with pytest.raises(ValueError):
        L.LweSampleArray(c1.params, c1.a, c2.b, c1.current_variances)         # lwe.py:113-117
    with pytest.raises(AttributeError):
        vm.not_a_gate


def test_small_batch_team_kernel_equals_wave_kernel(env, orc):
    """The 8-waves-per-bit half-ring kernel (batches <= CUs by default), the 4-waves-per-bit kernel (the same batches
    with nufhe_ctx_set_team8(0)), the 2-waves-per-bit kernel (<= 4 x CUs) and the wave-per-bit kernel give bit-identical
    ciphertexts, all equal to the oracle; the switches are exercised on both sides of their boundaries
    (nufhe_ctx_set_team_max_bits / nufhe_ctx_set_pair_max_bits / nufhe_ctx_set_team8)."""
    from nufhe_amd import _lib
    H = env['H']; thr = env['thr']; vm = env['vm']; ck = env['ck']; lwe_key = env['lwe_key']
    rng = orc.DeterministicRNG(2024)
    B = 70
    ms = [rng.uniform_bool((B,)).astype(bool) for _ in range(3)]
    cs = [orc.encrypt(rng, lwe_key, m) for m in ms]
    ds = [H.ciphertext_from_arrays(thr, c) for c in cs]
    exp_nand = orc.gate('gate_nand', ck, cs[0], cs[1])
    exp_mux = orc.gate_mux(ck, cs[0], cs[1], cs[2])
    try:
        # MUX launches 2 B = 140 bits in one bootstrap.  (team limit, pair limit): (0, 0) = wave kernel only,
        # (0, -1) = pair kernel, (69, 139): NAND pair / MUX wave, (70, 140): NAND team / MUX pair, ...
        for team8 in (1, 0):
            _lib.call("nufhe_ctx_set_team8", thr.handle, team8)
            for team, pair in ((0, 0), (0, -1), (69, 139), (69, 140), (70, 140), (139, 0), (140, -1), (-1, -1)):
                _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, team)
                _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, pair)
                for got, exp in ((vm.gate_nand(ds[0], ds[1]), exp_nand), (vm.gate_mux(ds[0], ds[1], ds[2]), exp_mux)):
                    ra, rb, rcv = H.ct_arrays(got)
                    assert (ra == exp[0]).all() and (rb == exp[1]).all() and (rcv == exp[2]).all(), (team8, team, pair)
    finally:
        _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, -1)
        _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, -1)
        _lib.call("nufhe_ctx_set_team8", thr.handle, 1)


def test_every_switch_point_plus_minus_one_bit(env, orc):
    """The kernel-family switch points come from the device (nufhe_ctx_get_tuning: a table keyed by architecture name +
    CU count); at every one of them, and one bit either side, the gate's words equal the oracle's (first and last 6 bits)
    -- NAND and, where the MUX's doubled rotation count crosses the boundary instead, MUX.  Then the same with every
    switch point moved (nufhe_ctx_set_tuning): results do not depend on where the switches sit."""
    import ctypes
    from nufhe_amd import _lib
    H = env['H']; thr = env['thr']; vm = env['vm']; ck = env['ck']; lwe_key = env['lwe_key']
    t = thr.tuning()
    assert t['measured'] == 1 and t['arch_name'].startswith('gfx950') and t['num_cus'] > 0
    cus = t['num_cus']
    assert t['team_max_bits'] == cus and t['pair_max_bits_ntt'] == 4 * cus and t['ks_mfma_min_bits'] == 2 * cus
    points = sorted({t['team_max_bits'], t['pair_max_bits_ntt'], t['ks_mfma_min_bits'], 8 * cus, 8 * cus + t['team_max_bits']})
    top = max(points) + 1
    rng = orc.DeterministicRNG(31)
    ms = [rng.uniform_bool((top,)).astype(bool) for _ in range(3)]
    cs = [orc.encrypt(rng, lwe_key, m) for m in ms]
    ds = [H.ciphertext_from_arrays(thr, c) for c in cs]

    def ends(c, size):
        idx = numpy.r_[0:6, size - 6:size]
        return tuple(x[idx] for x in c)

    def check(sizes, mux_sizes):
        for size in sizes:
            got = H.ct_arrays(vm.gate_nand(ds[0][:size], ds[1][:size]))
            exp = orc.gate('gate_nand', ck, ends(cs[0], size), ends(cs[1], size))
            for g, e in zip(ends(got, size), exp):
                assert (g == e).all(), ('nand', size)
        for size in mux_sizes:
            got = H.ct_arrays(vm.gate_mux(ds[0][:size], ds[1][:size], ds[2][:size]))
            exp = orc.gate_mux(ck, ends(cs[0], size), ends(cs[1], size), ends(cs[2], size))
            for g, e in zip(ends(got, size), exp):
                assert (g == e).all(), ('mux', size)

    sizes = sorted({p + d for p in points for d in (-1, 0, 1)})
    mux_sizes = sorted({p // 2 + d for p in (t['team_max_bits'], t['pair_max_bits_ntt']) for d in (0, 1)})
    check(sizes, mux_sizes)
    moved = _lib.NufheTuning()
    _lib.check(_lib.lib().nufhe_ctx_get_tuning(thr.handle, ctypes.byref(moved)))
    moved.team_max_bits = 37; moved.pair_max_bits_ntt = 201; moved.ks_mfma_min_bits = 100
    try:
        _lib.check(_lib.lib().nufhe_ctx_set_tuning(thr.handle, ctypes.byref(moved)))
        assert thr.tuning()['measured'] == 0 and thr.tuning()['team_max_bits'] == 37
        check([36, 37, 38, 100, 101, 200, 201, 202], [18, 19, 100, 101])
    finally:
        _lib.check(_lib.lib().nufhe_ctx_set_tuning(thr.handle, None))
    assert thr.tuning() == t


def test_replacing_the_key_drops_the_half_ring_copy(env, orc):
    """The 8-waves-per-bit kernel reads a second, lazily converted copy of the bootstrapping key; uploading other key
    material into the same handle must invalidate it.  A small gate (makes the copy), a different key uploaded, the same
    gate again: identical to the 4-waves-per-bit kernel, which reads the primary copy, and different from before."""
    import ctypes
    from nufhe_amd import _lib
    H = env['H']; thr = env['thr']; ck = env['ck']; lwe_key = env['lwe_key']
    import nufhe_amd
    cloud_key = H.cloud_key_from_arrays(thr, ck)          # a handle of its own: the module fixture stays untouched
    vm = env['ctx'].make_virtual_machine(cloud_key)
    rng = orc.DeterministicRNG(5)
    ms = [rng.uniform_bool((9,)).astype(bool) for _ in range(2)]
    ds = [H.ciphertext_from_arrays(thr, orc.encrypt(rng, lwe_key, m)) for m in ms]
    first = H.ct_arrays(vm.gate_nand(ds[0], ds[1]))
    other = numpy.ascontiguousarray(numpy.roll(numpy.asarray(ck.bk, numpy.uint64), 1, axis=0))   # rows rotated by one
    _lib.call("nufhe_bk_upload_reference", cloud_key.bootstrap_key._native.handle, other.ctypes.data_as(ctypes.c_void_p))
    try:
        with_copy = H.ct_arrays(vm.gate_nand(ds[0], ds[1]))
        _lib.call("nufhe_ctx_set_team8", thr.handle, 0)
        primary = H.ct_arrays(vm.gate_nand(ds[0], ds[1]))
    finally:
        _lib.call("nufhe_ctx_set_team8", thr.handle, 1)
    assert all((x == y).all() for x, y in zip(with_copy, primary))
    assert (first[0] != primary[0]).any()


def test_medium_batch_pair_kernel_every_group_size(env, orc):
    """The 2-waves-per-bit kernel with 1, 2, 3 and 4 pairs per work-group (ragged last group included) against the
    wave-per-bit kernel on the same ciphertexts: bit-identical; the first 24 bits also against the oracle."""
    import torch
    from nufhe_amd import _lib
    H = env['H']; thr = env['thr']; vm = env['vm']; ck = env['ck']; lwe_key = env['lwe_key']
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    rng = orc.DeterministicRNG(77)
    sizes = [cus - 3, 2 * cus - 5, 3 * cus - 1, 4 * cus]
    B = max(sizes)
    ms = [rng.uniform_bool((B,)).astype(bool) for _ in range(2)]
    cs = [orc.encrypt(rng, lwe_key, m) for m in ms]
    ds = [H.ciphertext_from_arrays(thr, c) for c in cs]
    exp = orc.gate('gate_nand', ck, tuple(x[:24] for x in cs[0]), tuple(x[:24] for x in cs[1]))
    try:
        for size in sizes:
            a, b = ds[0][:size], ds[1][:size]
            _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, 0)
            _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, 0)
            wave = H.ct_arrays(vm.gate_nand(a, b))
            _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, -1)
            pair = H.ct_arrays(vm.gate_nand(a, b))
            for x, y in zip(wave, pair):
                assert (x == y).all(), size
            assert (pair[0][:24] == exp[0]).all() and (pair[1][:24] == exp[1]).all() and (pair[2][:24] == exp[2]).all()
    finally:
        _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, -1)
        _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, -1)


def test_mux_job_boundary_inside_a_multi_pair_group(env, orc):
    """MUX on 301 bits = 602 bootstraps in one launch of the pair kernel with 3 pairs per work-group: one group holds
    the last bit of the first blind rotation and the first two of the second.  Bit-identical to the wave kernel; first
    and last 8 bits vs the oracle."""
    from nufhe_amd import _lib
    H = env['H']; thr = env['thr']; vm = env['vm']; ck = env['ck']; lwe_key = env['lwe_key']
    rng = orc.DeterministicRNG(606)
    B = 301
    ms = [rng.uniform_bool((B,)).astype(bool) for _ in range(3)]
    cs = [orc.encrypt(rng, lwe_key, m) for m in ms]
    ds = [H.ciphertext_from_arrays(thr, c) for c in cs]
    try:
        pair = H.ct_arrays(vm.gate_mux(ds[0], ds[1], ds[2]))
        _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, 0)
        _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, 0)
        wave = H.ct_arrays(vm.gate_mux(ds[0], ds[1], ds[2]))
    finally:
        _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, -1)
        _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, -1)
    for x, y in zip(pair, wave):
        assert (x == y).all()
    idx = numpy.r_[0:8, B - 8:B]
    exp = orc.gate_mux(ck, *[tuple(x[idx] for x in c) for c in cs])
    for g, e in zip(pair, exp):
        assert (g[idx] == e).all()


def test_keyswitch_on_the_matrix_cores_equals_lds_kernel(env, orc):
    """The two keyswitch kernels (one-hot int8 MFMA product over byte planes of the key / LDS row gather) give the same
    ciphertexts word for word, for NAND and for MUX (second source added on the fly), on ragged batches that exercise
    partly filled 64-bit tiles and 16-bit row groups; both equal the oracle."""
    from nufhe_amd import _lib
    H = env['H']; thr = env['thr']; vm = env['vm']; ck = env['ck']; lwe_key = env['lwe_key']
    rng = orc.DeterministicRNG(4242)
    B = 333
    ms = [rng.uniform_bool((B,)).astype(bool) for _ in range(3)]
    cs = [orc.encrypt(rng, lwe_key, m) for m in ms]
    ds = [H.ciphertext_from_arrays(thr, c) for c in cs]
    got = {}
    try:
        for mode in (0, 2):
            _lib.call("nufhe_ctx_set_keyswitch_mfma", thr.handle, mode)
            got[mode] = [H.ct_arrays(vm.gate_nand(ds[0][:n], ds[1][:n])) for n in (1, 17, 64, 65, B)]
            got[mode].append(H.ct_arrays(vm.gate_mux(ds[0], ds[1], ds[2])))
    finally:
        _lib.call("nufhe_ctx_set_keyswitch_mfma", thr.handle, 1)
    for a, b in zip(got[0], got[2]):
        for x, y in zip(a, b):
            assert (x == y).all()
    exp = orc.gate('gate_nand', ck, tuple(x[:65] for x in cs[0]), tuple(x[:65] for x in cs[1]))
    for g, e in zip(got[2][3], exp):
        assert (g == e).all()
    exp = orc.gate_mux(ck, *[tuple(x[:40] for x in c) for c in cs])
    for g, e in zip(got[2][5], exp):
        assert (g[:40] == e).all()


def test_profile_clock_of_the_wave_kernel(env, orc):
    """nufhe_profile_clock: the shader clock measured inside a profiled wave-per-bit launch is a plausible MI355X
    clock and the wave's life time is shorter than the launch; small batches (team kernel) report no measurement."""
    import ctypes
    import torch
    from nufhe_amd import _lib
    H = env['H']; thr = env['thr']; vm = env['vm']; lwe_key = env['lwe_key']
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    rng = orc.DeterministicRNG(5)
    B = 8 * cus
    m = [rng.uniform_bool((B,)).astype(bool) for _ in range(2)]
    ds = [H.ciphertext_from_arrays(thr, orc.encrypt(rng, lwe_key, x)) for x in m]
    lib = _lib.lib()
    lib.nufhe_profile_enable(thr.handle, 1)
    try:
        vm.gate_nand(ds[0], ds[1])
        br = ctypes.c_float(); ks = ctypes.c_float(); ghz = ctypes.c_double(); wave = ctypes.c_double()
        _lib.check(lib.nufhe_profile_last(thr.handle, ctypes.byref(br), ctypes.byref(ks)))
        _lib.check(lib.nufhe_profile_clock(thr.handle, ctypes.byref(ghz), ctypes.byref(wave)))
        assert 1.0 < ghz.value < 3.0
        assert 0.5 * br.value < wave.value <= br.value
        vm.gate_nand(ds[0][:16], ds[1][:16])
        assert lib.nufhe_profile_clock(thr.handle, ctypes.byref(ghz), ctypes.byref(wave)) != 0
        # nufhe_profile_history: the durations of EVERY gate since the last read, oldest first, without a wait per gate
        lib.nufhe_profile_enable(thr.handle, 1)
        sizes = (B, 16, B, 300)
        for n in sizes:
            vm.gate_nand(ds[0][:n], ds[1][:n])                       # nothing synchronises between these
        hb = (ctypes.c_float * 8)(); hk = (ctypes.c_float * 8)(); got = ctypes.c_int(-1)
        _lib.check(lib.nufhe_profile_history(thr.handle, hb, hk, 8, ctypes.byref(got)))
        assert got.value == 4 and all(hb[i] > 0 and hk[i] > 0 for i in range(4))
        assert hb[0] > 2 * hb[1] and hb[2] > 2 * hb[1] and abs(hb[0] - hb[2]) < 0.1 * hb[0]     # big, small, big
        _lib.check(lib.nufhe_profile_last(thr.handle, ctypes.byref(br), ctypes.byref(ks)))
        assert br.value == hb[3]                                      # "last" is the newest entry of the history
        _lib.check(lib.nufhe_profile_history(thr.handle, hb, hk, 8, ctypes.byref(got)))
        assert got.value == 0                                         # reading resets it
        for _ in range(3):
            vm.gate_nand(ds[0][:16], ds[1][:16])
        _lib.check(lib.nufhe_profile_history(thr.handle, hb, hk, 2, ctypes.byref(got)))
        assert got.value == 2                                         # capacity-limited: the two most recent
    finally:
        lib.nufhe_profile_enable(thr.handle, 0)


def test_large_batch_tail_goes_to_the_small_batch_kernels(env, orc):
    """A batch of whole rounds + a short last round is split into two launches (wave kernel for the rounds, team /
    pair kernel for the tail, for MUX across the boundary of its two blind rotations): bit-identical to the single
    wave-kernel launch on every bit, and to the oracle on the first and the last 12 bits."""
    import torch
    from nufhe_amd import _lib
    H = env['H']; thr = env['thr']; vm = env['vm']; ck = env['ck']; lwe_key = env['lwe_key']
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    rng = orc.DeterministicRNG(99)
    B = 10 * cus - 3
    ms = [rng.uniform_bool((B,)).astype(bool) for _ in range(3)]
    cs = [orc.encrypt(rng, lwe_key, m) for m in ms]
    ds = [H.ciphertext_from_arrays(thr, c) for c in cs]

    def both(fn):
        _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, 0)
        _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, 0)
        single = H.ct_arrays(fn())
        _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, -1)
        _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, -1)
        split = H.ct_arrays(fn())
        for x, y in zip(single, split):
            assert (x == y).all()
        return split

    def ends(c, size):
        idx = numpy.r_[0:12, size - 12:size]
        return tuple(x[idx] for x in c)

    try:
        for size in (8 * cus + 70, 10 * cus - 3):             # tails: team kernel, pair kernel
            got = both(lambda: vm.gate_nand(ds[0][:size], ds[1][:size]))
            exp = orc.gate('gate_nand', ck, ends(cs[0], size), ends(cs[1], size))
            for g, e in zip(ends(got, size), exp):
                assert (g == e).all(), size
        for size in (4 * cus + 35, 5 * cus - 2):               # MUX: 2 x size bootstraps, the tail lies in the second job
            got = both(lambda: vm.gate_mux(ds[0][:size], ds[1][:size], ds[2][:size]))
            exp = orc.gate_mux(ck, ends(cs[0], size), ends(cs[1], size), ends(cs[2], size))
            for g, e in zip(ends(got, size), exp):
                assert (g == e).all(), size
    finally:
        _lib.call("nufhe_ctx_set_team_max_bits", thr.handle, -1)
        _lib.call("nufhe_ctx_set_pair_max_bits", thr.handle, -1)


@pytest.mark.parametrize('script', ['gate_nand.py', 'gate_nand_low_level.py', 'serialization.py', 'gate_batch.py'])
def test_examples_run(script):
    """The counterparts of the reference's examples/ run end to end on the GPU."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run([sys.executable, os.path.join(root, 'examples', script), '--bits', '16'],
                          capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    assert 'OK' in proc.stdout


def test_bench_two_ranks_on_one_gpu():
    """bench.py's N > 1 path end to end (sharded bits, replicated keys, result gather, max-over-ranks
    timing), with two ranks sharing this box's GPU over the gloo test route; the RCCL route differs
    only in the backend string."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NUFHE_BENCH_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', '29533', os.path.join(root, 'bench.py'),
           '--gpus', '2', '--steps', '1', '--warmup', '1', '--bits', '256', '--no-extra']
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['correct'] and d['scaling'] == 'weak' and d['gather_ms'] > 0
    assert d['gather']['dst'] == 0 and d['gather']['verified'] is True and len(d['per_rank_ms_per_step']) == 2
    assert d['gather']['collectives_per_step'] == 1 and d['gather']['overlapped'] is True
    assert abs(d['value'] - 2 * 256 * 1000.0 / d['ms_per_step']) < 1e-6 * d['value']
    # an N > 1 line carries parity from EVERY rank (first 256 ciphertexts of each shard vs the oracle) and the roofline
    assert d['parity']['ranks_reporting'] == 2 and d['parity']['differing'] == 0 and d['parity']['words'] == 2 * 256 * 501
    assert d['parity']['variances_differing'] == 0 and d['roofline']['kernel_ms'] > 0


@pytest.mark.parametrize('hook', ['init', 'gather'])
def test_bench_survives_an_rccl_failure_with_an_error_field(hook):
    """RCCL has never run with more than one rank before the driver's own 8-GPU run.  If its process
    group or its first gather fails, the line must still come out -- over the host-staged gloo gather, with the reason
    under "error" -- instead of a traceback (the failure is injected: NUFHE_BENCH_FAIL_RCCL)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != 'NUFHE_BENCH_BACKEND'}
    env['NUFHE_BENCH_FAIL_RCCL'] = hook
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
           '--master-addr', '127.0.0.1', '--master-port', '29561', os.path.join(root, 'bench.py'),
           '--gpus', '1', '--steps', '2', '--warmup', '1', '--bits', '256', '--no-extra', '--no-cpu-baseline']
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert 'NUFHE_BENCH_FAIL_RCCL=%s' % hook in d['error'] and d['gather']['backend'] == 'gloo' and d['gather']['requested'] == 'rccl'
    assert d['correct'] and d['gather']['verified'] is True and d['roofline']['kernel_ms'] > 0 and d['value'] > 0
    assert d['gather']['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' and d['gather']['env']['NCCL_DEBUG']


def test_bench_peer_route_one_process():
    """`--gather-backend peer`: the RCCL-free route of the N > 1 line (one process, a DeviceThread per GPU, slices
    collected by nufhe_gather) -- with the one GPU of this box"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--gather-backend', 'peer', '--steps', '2',
           '--warmup', '1', '--bits', '300']
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
    d = json.loads([l for l in proc.stdout.splitlines() if l.startswith('{')][0])
    assert d['correct'] and d['n_gpus'] == 1 and d['gather']['backend'].startswith('peer') and d['value'] > 0


@pytest.mark.parametrize('backend,nproc,bits', [('gloo', 2, 70), ('gloo', 2, 37), ('nccl', 1, 64)])
def test_multi_gpu_example_under_torchrun(backend, nproc, bits):
    """examples/multi_gpu.py (the counterpart of the reference's examples/multi_gpu.py:46-114) launched the way
    config 4 launches it: two ranks sharing this GPU over the gloo test route (even and ragged split: every rank
    other than 0 must come through the gather with None and exit cleanly), and one rank over RCCL."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != 'NUFHE_BENCH_BACKEND'}
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc),
           '--master-addr', '127.0.0.1', '--master-port', '29547', os.path.join(root, 'examples', 'multi_gpu.py'),
           '--bits', str(bits), '--backend', backend]
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-3000:]
    assert 'multi-GPU gate_nand OK: %d bits over %d GPU(s), gathered to rank 0 over %s' % (bits, nproc, backend) in proc.stdout


_every_word_cache = {}


@pytest.mark.parametrize("engine", ["native", "exact-fft"])
def test_config2_config3_every_word_vs_oracle(env, orc, engine):
    """BASELINE configs 2 and 3 at their full size: EVERY output word (a[500], b) and variance of the
    4096-bit NAND equals the CPU oracle (4096 x 502 words), and so does every word of ALL 4096 ciphertexts of the
    4096-bit MUX (two blind rotations per bit); test/test_gates.py:178-228 structure, lwe_cpu.py:62-93
    for the keyswitch.  The oracle needs ~4-5 minutes on the box's host cores (run once, shared by both engines of the
    NTT path: the u64 prime-field kernels and the exact fp64 engine of nufhe_cloudkey_set_engine)."""
    H = env['H']; vm = env['vm']; ctx = env['ctx']; sk = env['secret_key']; ck = env['ck']
    B = 4096
    if 'inputs' not in _every_word_cache:
        rng = numpy.random.RandomState(2024)
        ms = [rng.randint(0, 2, size=(B,)).astype(bool) for _ in range(3)]
        cs = [ctx.encrypt(sk, m) for m in ms]
        host = [H.ct_arrays(c) for c in cs]
        _every_word_cache['inputs'] = (ms, cs, host)
        _every_word_cache['nand'] = orc.gate('gate_nand', ck, host[0][:2], host[1][:2])
        _every_word_cache['mux'] = orc.gate_mux(ck, *[tuple(h[:2]) for h in host])
    ms, cs, host = _every_word_cache['inputs']
    env['cloud_key'].set_engine(engine)
    try:
        assert env['cloud_key'].engine == engine
        ra, rb, rcv = H.ct_arrays(vm.gate_nand(cs[0], cs[1]))
        rm = vm.gate_mux(cs[0], cs[1], cs[2])
    finally:
        env['cloud_key'].set_engine('native')
    exp = _every_word_cache['nand']
    assert ra.size + rb.size == B * 501
    assert int((ra != exp[0]).sum()) == 0 and int((rb != exp[1]).sum()) == 0 and int((rcv != exp[2]).sum()) == 0
    assert (ctx.decrypt(sk, rm) == numpy.where(ms[0], ms[1], ms[2])).all()
    ma, mb, mcv = H.ct_arrays(rm)
    expm = _every_word_cache['mux']
    assert ma.size + mb.size == B * 501
    assert int((ma != expm[0]).sum()) == 0 and int((mb != expm[1]).sum()) == 0 and int((mcv != expm[2]).sum()) == 0


def test_config4_eight_logical_shards_on_one_device(env, orc):
    """BASELINE config 4 (32768-bit NAND over 8 GPUs, examples/multi_gpu.py:86-114) as far as one GPU
    allows: the batch is cut by the same shard_bounds the ranks use, each of the 8 shards of 4096 bits
    goes through gate_nand as its own call (what rank r would run), the slices are concatenated in rank
    order.  Every decrypted bit and the first 32 ciphertexts of EACH shard are checked against the
    truth table / the CPU oracle.  (N-GPU wall-clock is not measured here: see DESIGN.md §6.)"""
    import nufhe_amd
    from nufhe_amd import multi_gpu
    H = env['H']; vm = env['vm']; ctx = env['ctx']; sk = env['secret_key']; ck = env['ck']
    rng = numpy.random.RandomState(4)
    B, G = 32768, 8
    m1 = rng.randint(0, 2, size=(B,)).astype(bool); m2 = rng.randint(0, 2, size=(B,)).astype(bool)
    c1 = ctx.encrypt(sk, m1); c2 = ctx.encrypt(sk, m2)
    parts = []
    for r in range(G):
        lo, hi = multi_gpu.shard_bounds(B, G, r)
        assert hi - lo == 4096
        s1 = multi_gpu.shard_ciphertext(c1, G, r); s2 = multi_gpu.shard_ciphertext(c2, G, r)
        parts.append(vm.gate_nand(s1, s2))
        exp = orc.gate('gate_nand', ck, tuple(x[lo:lo + 32] for x in H.ct_arrays(c1)[:2]),
                       tuple(x[lo:lo + 32] for x in H.ct_arrays(c2)[:2]))
        pa, pb, pcv = H.ct_arrays(parts[-1])
        assert (pa[:32] == exp[0]).all() and (pb[:32] == exp[1]).all() and (pcv[:32] == exp[2]).all(), r
    full = nufhe_amd.concatenate(parts)
    assert full.shape == (B,)
    assert (ctx.decrypt(sk, full) == ~(m1 & m2)).all()
    # one unsharded call over all 32768 bits gives the same ciphertexts (bits are independent)
    whole = vm.gate_nand(c1, c2)
    assert whole == full


def test_bench_rccl_route_world_size_one():
    """bench.py launched by torch.distributed.run with ONE rank and the `nccl` backend: the process
    group is created with device_id=, the per-step result gather (device int32 / float32 tensors) and
    the timing collectives go through RCCL -- the same code path the 8-GPU run takes, minus the xGMI
    links."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != 'NUFHE_BENCH_BACKEND'}
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
           '--master-addr', '127.0.0.1', '--master-port', '29541', os.path.join(root, 'bench.py'),
           '--gpus', '1', '--steps', '2', '--warmup', '1', '--bits', '512', '--no-extra', '--no-cpu-baseline']
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['correct']
    assert d['gather']['backend'] == 'nccl' and d['gather']['verified'] is True and d['gather_ms'] > 0
    assert len(d['per_rank_ms_per_step']) == 1


def _bench_line(proc):
    import json
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-3000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, proc.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` -- the driver's literal command form, NO torch.distributed.run around it -- starts
    two ranks by itself (nufhe_amd.multi_gpu.launch_ranks; the reference's example starts its own per-GPU workers,
    examples/multi_gpu.py:86-114) and prints ONE line with n_gpus == 2, per-rank step times and a parity verdict
    from both ranks over 256 ciphertexts each.  Two ranks share this box's GPU over the gloo test route."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NUFHE_BENCH_BACKEND='gloo')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--bits', '256', '--no-extra']
    d = _bench_line(subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root))
    assert d['n_gpus'] == 2 and d['correct'] and d['scaling'] == 'weak'
    assert len(d['per_rank_ms_per_step']) == 2 and all(t > 0 for t in d['per_rank_ms_per_step'])
    assert d['parity']['ranks_reporting'] == 2 and d['parity']['differing'] == 0
    assert d['parity']['bits_per_rank'] == 256 and d['parity']['words'] == 2 * 256 * 501
    assert d['parity']['variances_differing'] == 0
    assert abs(d['value'] - 2 * 256 * 1000.0 / d['ms_per_step']) < 1e-6 * d['value']
    assert d['config']['workload'].endswith('(BASELINE config 4)')
    _check_multi_rank_line(d, 2)


def _check_multi_rank_line(d, world):
    """what an N > 1 line must carry to be judged like the N = 1 line: cpu_baseline (timed in this run), roofline with the
    per-rank kernel times / clocks, parity over every rank"""
    cb = d['cpu_baseline']
    assert cb['kind'] == 'port' and cb['unit'] == 'gates/s' and cb['value'] > 0 and cb['cores'] >= world
    assert cb['ranks'] == world and 'sample' in cb and cb['reference_python']['measured_in_this_run'] is False
    roof = d['roofline']
    assert roof['kernel_ms'] > 0 and roof['bound'] == 'hbm' and roof['unit'] == 'GB/s' and 0 < roof['frac'] < 1.5
    assert abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-12 and 'valu_issue' in roof
    pr = roof['per_rank']
    assert len(pr['kernel_ms']) == world and len(pr['clock_ghz_in_kernel']) == world
    assert 0 < pr['kernel_ms_min'] <= roof['kernel_ms'] <= pr['kernel_ms_max']
    # (256 bits per rank run the small-batch kernels, which carry no clock probe: None there, > 1 GHz where measured)
    assert all(c is None or c > 1.0 for c in pr['clock_ghz_in_kernel'])
    assert d['gather']['verified'] is True and d['gather_ms'] > 0


def test_bench_gpus_8_rehearsal_on_one_gpu():
    """The driver's 8-GPU command, rehearsed: `python bench.py --gpus 8` (literal form, no launcher) with eight ranks
    sharing this box's one GPU over the gloo test route.  The line must be complete -- n_gpus 8, parity from all 8
    ranks with 0 differing words, cpu_baseline, roofline incl. per-rank kernel times -- and arrive in bounded time, so
    that the first run on 8 physical GPUs (RCCL instead of gloo, nothing else changes) is not lost to a formality."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NUFHE_BENCH_BACKEND='gloo')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    # (64 of the 256 ciphertexts of every rank go through the CPU oracle: the eight ranks share this host's cores)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--steps', '3', '--warmup', '1',
           '--bits', '256', '--no-extra', '--cpu-sample-bits', '64']
    t0 = time.time()
    d = _bench_line(subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root))
    wall = time.time() - t0
    assert d['n_gpus'] == 8 and d['correct'] and d['scaling'] == 'weak' and d['steps'] == 3
    assert d['parity']['ranks_reporting'] == 8 and d['parity']['differing'] == 0
    assert d['parity']['words'] == 8 * 64 * 501 and d['parity']['variances_differing'] == 0
    assert len(d['per_rank_ms_per_step']) == 8
    assert abs(d['value'] - 8 * 256 * 1000.0 / d['ms_per_step']) < 1e-6 * d['value']
    _check_multi_rank_line(d, 8)
    assert wall < 600, wall


def test_bench_refuses_more_rccl_ranks_than_gpus():
    """Over RCCL every rank needs a GPU of its own: `--gpus N` with fewer devices fails loudly BEFORE starting anything
    (it must never fall back to measuring one GPU and printing n_gpus 1), and a launcher whose world size contradicts
    --gpus is refused as well."""
    import subprocess
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in (
        'NUFHE_BENCH_BACKEND', 'RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    n = torch.cuda.device_count() + 1
    proc = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(n), '--steps', '1'],
                          capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert proc.returncode != 0 and 'need %d GPUs' % n in proc.stderr and '{' not in proc.stdout
    env2 = dict(env, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    proc = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1'],
                          capture_output=True, text=True, timeout=300, env=env2, cwd=root)
    assert proc.returncode != 0 and 'WORLD_SIZE=1' in proc.stderr and '{' not in proc.stdout


@pytest.mark.parametrize('via_host', [False, True])
def test_multi_gpu_example_gpus_2_starts_its_own_ranks(via_host):
    """`python examples/multi_gpu.py --gpus 2` without a launcher; the cloud key reaches rank 1 as one device broadcast of
    its image (default) or, with --via-host, pickled through the host like the reference's example."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in (
        'NUFHE_BENCH_BACKEND', 'RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    cmd = [sys.executable, os.path.join(root, 'examples', 'multi_gpu.py'), '--gpus', '2', '--bits', '45',
           '--backend', 'gloo'] + (['--via-host'] if via_host else [])
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-3000:]
    assert 'multi-GPU gate_nand OK: 45 bits over 2 GPU(s), gathered to rank 0 over gloo' in proc.stdout
    assert ('key via host pickle' if via_host else 'key as one device broadcast') in proc.stdout


def test_multi_gpu_example_gpus_8_rehearsal():
    """`python examples/multi_gpu.py --gpus 8 --bits 2048`: the reference example's shape (examples/multi_gpu.py:86-114) at
    the node's rank count, eight ranks sharing this GPU over gloo, key replicated by one broadcast of its image"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in (
        'NUFHE_BENCH_BACKEND', 'RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    cmd = [sys.executable, os.path.join(root, 'examples', 'multi_gpu.py'), '--gpus', '8', '--bits', '2048',
           '--backend', 'gloo']
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-3000:]
    assert 'multi-GPU gate_nand OK: 2048 bits over 8 GPU(s), gathered to rank 0 over gloo' in proc.stdout


def test_cloud_key_device_image_round_trip(env, orc):
    """NuFHECloudKey.device_image / from_device_image (nufhe_cloudkey_export_image / _import_image): the key rebuilt
    from its device image is the same key -- reference-format bootstrapping and keyswitch arrays equal, and gates of
    every kernel family (8-wave team: 5 bits, pair: 300, wave + matrix-core keyswitch: 2100) give identical words."""
    import torch
    from nufhe_amd.api_low_level import NuFHECloudKey
    H = env['H']; thr = env['thr']; ctx = env['ctx']; sk = env['secret_key']; ck0 = env['cloud_key']
    params, image = ck0.device_image()
    assert image.dtype == torch.uint8 and image.numel() == ck0._native.image_bytes() >= 32_768_000 + 49_152_000
    ck1 = NuFHECloudKey.from_device_image(thr, params, image.clone())
    del image
    assert ck1 == ck0                                    # downloads both keys in the reference's formats
    vm0 = ctx.make_virtual_machine(ck0); vm1 = ctx.make_virtual_machine(ck1)
    rng = numpy.random.RandomState(77)
    for B in (5, 300, 2100):
        m1 = rng.randint(0, 2, size=(B,)).astype(bool); m2 = rng.randint(0, 2, size=(B,)).astype(bool)
        c1 = ctx.encrypt(sk, m1); c2 = ctx.encrypt(sk, m2)
        r0 = vm0.gate_nand(c1, c2); r1 = vm1.gate_nand(c1, c2)
        assert r0 == r1 and (ctx.decrypt(sk, r1) == ~(m1 & m2)).all()
    with pytest.raises(ValueError):
        ck1._native.import_image(torch.zeros(1000, dtype=torch.uint8, device=thr.device))


@pytest.mark.parametrize('transform', ['NTT', 'FFT'])
def test_secure_rng_end_to_end(transform):
    """test/test_api_high_level.py:114-132 (`test_rngs[SecureRNG]`): Context(rng=SecureRNG()) -> key pair -> encrypt ->
    gates -> decrypt equals the truth tables; NAND as in the reference's test plus MUX, on both transforms."""
    import random
    import nufhe_amd as nufhe
    size = 32
    bits = [[random.choice([False, True]) for _ in range(size)] for _ in range(3)]
    ctx = nufhe.Context(rng=nufhe.SecureRNG())
    secret_key, cloud_key = ctx.make_key_pair(transform_type=transform)
    cts = [ctx.encrypt(secret_key, b) for b in bits]
    vm = ctx.make_virtual_machine(cloud_key)
    b = [numpy.array(x) for x in bits]
    assert all(ctx.decrypt(secret_key, vm.gate_nand(cts[0], cts[1])) == ~(b[0] & b[1]))
    assert all(ctx.decrypt(secret_key, vm.gate_mux(cts[0], cts[1], cts[2])) == numpy.where(b[0], b[1], b[2]))
    # a second key pair from the same SecureRNG is a different key
    sk2, _ = ctx.make_key_pair(transform_type=transform)
    assert not (sk2 == secret_key)


@pytest.mark.parametrize('single_kernel_bootstrap', [True, False])
def test_gate_over_view(env, orc, single_kernel_bootstrap):
    """test/test_gates.py:514-559 with its own slices: operands and result are strided views of (5, 8) arrays, through
    vm.gate_nand, in both bootstrap modes (fused kernels / the reference's step-by-step sequence).  Beyond the
    reference's decrypt check: the words written through the view equal the oracle's on the same operands, and
    nothing outside the view is touched."""
    import nufhe_amd
    from nufhe_amd.performance import PerformanceParameters
    H = env['H']; thr = env['thr']; ctx = env['ctx']; sk = env['secret_key']; cloud_key = env['cloud_key']; ck = env['ck']
    vm = ctx.make_virtual_machine(cloud_key, perf_params=PerformanceParameters(
        cloud_key.params, single_kernel_bootstrap=single_kernel_bootstrap))
    rng = numpy.random.RandomState(5)
    shape = (5, 8)
    slices1 = (slice(3, 5), slice(1, 7, 2))
    slices2 = (slice(1, 3), slice(2, 8, 2))
    result_slices = (slice(2, 4), slice(0, 6, 2))
    pts = [rng.randint(0, 2, size=shape).astype(bool) for _ in range(2)]
    cts = [ctx.encrypt(sk, p) for p in pts]
    answer = vm.empty_ciphertext(shape)
    answer.a.fill_(7); answer.b.fill_(7); answer.current_variances.fill_(7.0)
    vm.gate_nand(cts[0][slices1], cts[1][slices2], dest=answer[result_slices])
    ha = [H.ct_arrays(c) for c in cts]
    exp = orc.gate('gate_nand', ck, tuple(x[slices1] for x in ha[0][:2]), tuple(x[slices2] for x in ha[1][:2]))
    ra, rb, rcv = H.ct_arrays(answer)
    assert (ra[result_slices] == exp[0]).all() and (rb[result_slices] == exp[1]).all()
    mask = numpy.ones(shape, bool); mask[result_slices] = False
    assert (ra[mask] == 7).all() and (rb[mask] == 7).all() and (rcv[mask] == 7.0).all()
    dec = ctx.decrypt(sk, answer)
    assert (dec[result_slices] == ~(pts[0][slices1] & pts[1][slices2])).all()


@pytest.mark.slow
def test_extended_parity_every_word():
    """tools/extended_parity.py as a test (`pytest -m "gpu and slow"`; skipped by a plain `-m gpu` run: ~2-3 minutes of
    CPU oracle): EVERY output word of 2048-bit NAND / XOR / MUX (wave kernel), 700-bit NAND (pair kernel), 200-bit
    NAND / XNOR / MUX and 100-bit MUX (8-wave half-ring team kernel / pair kernel) and a 256-bit k = 2 NAND equals the
    CPU oracle -- 3.9 M words, zero differing."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run([sys.executable, os.path.join(root, 'tools', 'extended_parity.py')],
                          capture_output=True, text=True, timeout=3000, cwd=root)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
    d = json.loads([l for l in proc.stdout.splitlines() if l.startswith('{')][-1])
    assert d['total_differing'] == 0
    assert sum(v['words'] for part in ('k1', 'k2') for v in d[part].values()) > 3_800_000
