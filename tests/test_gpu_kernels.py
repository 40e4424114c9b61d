"""
GPU parity tests, kernel granularity (the reference's differential-test structure: every kernel
against its CPU reference function, exact equality for integers): HIP library through the C ABI
vs (a) golden outputs of the REFERENCE's own CPU functions and (b) the oracle.
"""

import ctypes

import numpy
import pytest

import golden_inputs as gi

pytestmark = pytest.mark.gpu

P = 2**64 - 2**32 + 1


@pytest.fixture(scope='module')
def thr():
    from nufhe_amd.device import DeviceThread
    t = DeviceThread(0)
    yield t


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


def _ntt(thr, H, fn, data, out_dtype):
    d = H.dev(thr, data)
    out = thr.array(data.shape, out_dtype)
    call(fn, thr.handle, ptr(out), ptr(d), data.size // 1024)
    thr.synchronize()
    return H.host_u64(out) if out_dtype == numpy.uint64 else H.host(out)


def test_ntt_vs_reference_golden(thr, H, golden, orc):
    polys_i32, polys_ff = gi.ntt_inputs()
    assert (_ntt(thr, H, "nufhe_ntt_forward_i32", polys_i32, numpy.uint64) == golden['ntt_forward_i32']).all()
    assert (_ntt(thr, H, "nufhe_ntt_forward_u64", polys_ff, numpy.uint64) == golden['ntt_forward_u64']).all()
    assert (_ntt(thr, H, "nufhe_ntt_inverse_i32", polys_ff, numpy.int32) == golden['ntt_inverse_i32']).all()
    assert (_ntt(thr, H, "nufhe_ntt_inverse_u64", polys_ff, numpy.uint64) == golden['ntt_inverse_u64']).all()


def test_ntt_batch_vs_oracle_and_product(thr, H, orc):
    # test_computation.py:33-124 of the reference: exact transform, and transform-based negacyclic
    # product == schoolbook, on a batch that spans many blocks (odd size: ragged last block)
    rs = numpy.random.RandomState(31)
    a = rs.randint(-2**31, 2**31, size=(1037, 1024), dtype=numpy.int32)
    f = _ntt(thr, H, "nufhe_ntt_forward_i32", a, numpy.uint64)
    assert (f == orc.ntt_forward(a)).all()
    assert (_ntt(thr, H, "nufhe_ntt_inverse_i32", f, numpy.int32) == a).all()
    b = rs.randint(-1000, 1000, size=(5, 1024)).astype(numpy.int32)
    x = H.dev(thr, a[:35]); y = H.dev(thr, b)
    out = thr.array((35, 1024), numpy.int32)
    call("nufhe_poly_mul_i32", thr.handle, ptr(out), ptr(x), ptr(y), 35, 5)
    exp = orc.poly_mul_schoolbook(a[:35], b[numpy.arange(35) % 5])
    assert (H.host(out) == exp).all()


def test_t32_to_phase(thr, H, golden):
    x = gi.modswitch_inputs()
    out = thr.array(x.shape, numpy.int32)
    d_x = H.dev(thr, x)
    call("nufhe_t32_to_phase", thr.handle, ptr(out), ptr(d_x), x.size, 2048)
    assert (H.host(out) == golden['t32_to_phase']).all()


def test_shift(thr, H, golden):
    src, powers, N = gi.shift_inputs()['n1024']
    d_src = H.dev(thr, src); d_pow = H.dev(thr, powers)     # keep the device buffers alive
    for minus_one in (False, True):
        for invert in (False, True):
            out = thr.array(src.shape, numpy.int32)
            call("nufhe_shift_torus_polynomial", thr.handle, ptr(out), ptr(d_src),
                 ptr(d_pow), 1, 0, powers.size, 2, int(minus_one), int(invert))
            assert (H.host(out) == golden['shift_n1024_m%d_i%d' % (minus_one, invert)]).all()
    src, powers_arr, idx, N = gi.shift_view_inputs()
    d_src = H.dev(thr, src); d_pow = H.dev(thr, powers_arr)
    out = thr.array(src.shape, numpy.int32)
    call("nufhe_shift_torus_polynomial", thr.handle, ptr(out), ptr(d_src),
         ptr(d_pow), powers_arr.shape[1], idx, powers_arr.shape[0], 2, 1, 0)
    assert (H.host(out) == golden['shift_view']).all()


def test_extract(thr, H, golden):
    tl = gi.tlwe_extract_inputs()
    ra = thr.array((2, 3, 1024), numpy.int32); rb = thr.array((2, 3), numpy.int32)
    d_tl = H.dev(thr, tl)
    call("nufhe_tlwe_extract", thr.handle, ptr(ra), ptr(rb), ptr(d_tl), 6, 1)
    assert (H.host(ra) == golden['tlwe_extract_a']).all() and (H.host(rb) == golden['tlwe_extract_b']).all()


def test_simd_partners_finish_together(thr):
    """The throughput kernels put two one-bit waves on every SIMD and pace them against each other (BrPace,
    csrc/blind_rotate.h): the arbiter alone would let the older wave run ahead and leave the younger one a third of
    the kernel ALONE at ~60 % issue utilisation (46 ms instead of 39 for the 4096-bit NAND).  On a batch of 8 x CUs
    bits the waves of work-group 0 that share a SIMD must end within 2 % of the kernel time of each other, and every
    wave must live to (almost) the end of the kernel -- for both transforms."""
    import ctypes
    import nufhe_amd
    from nufhe_amd import _lib
    L = _lib.lib()
    cus = thr.device_params.compute_units
    B = 8 * cus
    for transform in ('NTT', 'FFT'):
        ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(5), thread=thr)
        sk, ck = ctx.make_key_pair(transform_type=transform)
        vm = ctx.make_virtual_machine(ck)
        rs = numpy.random.RandomState(3)
        c1 = ctx.encrypt(sk, rs.randint(0, 2, size=B).astype(bool)); c2 = ctx.encrypt(sk, rs.randint(0, 2, size=B).astype(bool))
        out = vm.empty_ciphertext((B,))
        vm.gate_nand(c1, c2, dest=out)                      # warm-up
        L.nufhe_profile_enable(thr.handle, 1)
        try:
            vm.gate_nand(c1, c2, dest=out)
            br = ctypes.c_float(); ks = ctypes.c_float()
            _lib.check(L.nufhe_profile_last(thr.handle, ctypes.byref(br), ctypes.byref(ks)))
            start = (ctypes.c_double * 8)(); end = (ctypes.c_double * 8)(); simd = (ctypes.c_int * 8)(); n = ctypes.c_int()
            _lib.check(L.nufhe_profile_waves(thr.handle, start, end, simd, 8, ctypes.byref(n)))
        finally:
            L.nufhe_profile_enable(thr.handle, 0)
        assert n.value == 8
        kernel_ms = br.value
        by_simd = {}
        for w in range(8):
            by_simd.setdefault(simd[w], []).append(end[w])
        assert sorted(len(v) for v in by_simd.values()) == [2, 2, 2, 2], by_simd     # two waves on each of the 4 SIMDs
        spread = max(abs(v[0] - v[1]) for v in by_simd.values())
        print("%s: kernel %.2f ms, wave ends %s ms, largest partner gap %.3f ms" % (
            transform, kernel_ms, [round(end[w], 2) for w in range(8)], spread))
        assert spread <= 0.02 * kernel_ms, (transform, spread, kernel_ms)
        # (the wave clocks start after the launch ramp and the table load, ~0.4 ms: 6 % of the 7 ms FFT kernel)
        assert min(end[w] for w in range(8)) >= 0.88 * kernel_ms
        del vm, ck


def test_tgsw_decompose_vs_reference_golden(thr, H, golden):
    """the gadget decomposition as a standalone device call (test/test_tgsw.py:44-69): full-range accumulators with
    the boundary values in front, against the output of the reference's tgsw_polynomial_decomp_trf_reference"""
    x = gi.decomp_inputs()                                   # int32 (2, 3, 2, 1024)
    d_x = H.dev(thr, x)
    out = thr.array(x.shape[:-1] + (2, 1024), numpy.int32)
    call("nufhe_tgsw_decompose", thr.handle, ptr(out), ptr(d_x), x.size // 1024)
    got = H.host(out)
    assert got.shape == golden['tgsw_decomp'].shape and (got == golden['tgsw_decomp']).all()
    assert got.min() >= -512 and got.max() <= 511
    with pytest.raises(ValueError):
        call("nufhe_tgsw_decompose", thr.handle, ptr(out), ptr(d_x), -1)


def test_tgsw_mac_vs_reference_golden(thr, H, golden):
    """the transformed-domain multiply-accumulate as a standalone device call (test/test_tgsw.py:72-115) on
    reference-format arrays, against tlwe_transformed_add_mul_to_trf_reference's output"""
    tr_sample, bk, row = gi.mac_inputs()                     # uint64 (2, 3, 2, 2, 1024), (4, 2, 2, 2, 1024)
    d_s = H.dev(thr, tr_sample); d_bk = H.dev(thr, bk)
    out = thr.array(tr_sample.shape[:-3] + (2, 1024), numpy.uint64)
    call("nufhe_tgsw_mac", thr.handle, ptr(out), ptr(d_s), ptr(d_bk), bk.shape[0], row, 6, 1)
    got = H.host_u64(out)
    assert (got == golden['tgsw_mac']).all()
    with pytest.raises(ValueError):
        call("nufhe_tgsw_mac", thr.handle, ptr(out), ptr(d_s), ptr(d_bk), bk.shape[0], bk.shape[0], 6, 1)   # row out of range
    with pytest.raises(ValueError):
        call("nufhe_tgsw_mac", thr.handle, ptr(out), ptr(d_s), ptr(d_bk), bk.shape[0], row, 6, 3)


def test_lwe_linear_and_trivial(thr, H, golden):
    from nufhe_amd import lwe as L
    from nufhe_amd.api_low_level import NuFHEParameters
    params = NuFHEParameters()
    res, src = gi.linear_inputs()
    for p, add in ((1, False), (-1, True), (2, True), (-2, True)):
        r = H.ciphertext_from_arrays(thr, res); s = H.ciphertext_from_arrays(thr, src)
        L._linear(thr, r, s, p, add)
        ra, rb, rcv = H.ct_arrays(r)
        assert (ra == golden['linear_p%d_add%d_a' % (p, add)]).all()
        assert (rb == golden['linear_p%d_add%d_b' % (p, add)]).all()
        assert (rcv == golden['linear_p%d_add%d_cv' % (p, add)]).all()
    # broadcast of the source over a leading axis + strided destination view (test_lwe.py:216-294)
    r = H.ciphertext_from_arrays(thr, res)
    s1 = H.ciphertext_from_arrays(thr, tuple(x[0] for x in src))
    L.lwe_add_to(thr, r[::2], s1)
    ra, rb, rcv = H.ct_arrays(r)
    exp_a = res[0].copy(); exp_a[::2] += src[0][0]
    assert (ra == exp_a).all()
    t = L.LweSampleArray.empty(thr, params.in_out_params, (5, 3))
    L.lwe_noiseless_trivial_constant(thr, t, -123)
    ta, tb, tcv = H.ct_arrays(t)
    assert (ta == 0).all() and (tb == -123).all() and (tcv == 0).all()
    # lwe_noiseless_trivial with full-shape, trailing-axis and scalar sources (test_lwe.py:325-385)
    rs = numpy.random.RandomState(77)
    for src_shape in ((5, 3), (3,), ()):
        mus = rs.randint(-2**31, 2**31, size=src_shape, dtype=numpy.int32)
        t = H.ciphertext_from_arrays(thr, (rs.randint(-9, 9, size=(5, 3, 500), dtype=numpy.int32),
                                           rs.randint(-9, 9, size=(5, 3), dtype=numpy.int32),
                                           numpy.ones((5, 3), numpy.float32)))
        L.lwe_noiseless_trivial(thr, t, mus)
        ta, tb, tcv = H.ct_arrays(t)
        assert (ta == 0).all() and (tb == numpy.broadcast_to(mus, (5, 3))).all() and (tcv == 0).all()


def _key_with_bk(thr, bk):
    from nufhe_amd.bootstrap import NativeCloudKey
    native = NativeCloudKey(thr, bk.shape[0])
    bk = numpy.ascontiguousarray(bk, numpy.uint64)
    call("nufhe_bk_upload_reference", native.handle, bk.ctypes.data_as(ctypes.c_void_p))
    return native


def test_bk_roundtrip_reference_format(thr, H):
    rs = numpy.random.RandomState(33)
    bk = rs.randint(0, P, size=(7, 2, 2, 2, 1024), dtype=numpy.uint64)
    native = _key_with_bk(thr, bk)
    back = numpy.empty_like(bk)
    call("nufhe_bk_download_reference", native.handle, back.ctypes.data_as(ctypes.c_void_p))
    assert (back == bk).all()


def test_external_mul_vs_reference_golden(thr, H, golden):
    # test_tgsw.py:118-154 of the reference
    for full in (False, True):
        accum, bk, row = gi.extmul_inputs(full_range=full)
        native = _key_with_bk(thr, bk)
        acc = H.dev(thr, accum)
        call("nufhe_external_mul", thr.handle, native.handle, ptr(acc), row, 6)
        assert (H.host(acc) == golden['tgsw_extmul_full' if full else 'tgsw_extmul']).all()


def test_blind_rotate_vs_reference_golden(thr, H, golden):
    acc0, bk, bara = gi.blind_rotate_inputs()
    native = _key_with_bk(thr, bk)
    acc = H.dev(thr, acc0); d_bara = H.dev(thr, bara)
    call("nufhe_blind_rotate", thr.handle, native.handle, ptr(acc), ptr(d_bara), bara.shape[1],
         bk.shape[0], acc0.shape[0])
    assert (H.host(acc) == golden['blind_rotate_acc']).all()
    ra = thr.array((2, 1024), numpy.int32); rb = thr.array((2,), numpy.int32)
    call("nufhe_tlwe_extract", thr.handle, ptr(ra), ptr(rb), ptr(acc), 2, 1)
    assert (H.host(ra) == golden['blind_rotate_ext_a']).all() and (H.host(rb) == golden['blind_rotate_ext_b']).all()


def test_blind_rotate_batch_vs_oracle(thr, H, orc):
    # odd batch (ragged last work-group), more rows
    rs = numpy.random.RandomState(34)
    B, n = 21, 9
    bk = rs.randint(0, P, size=(n, 2, 2, 2, 1024), dtype=numpy.uint64)
    acc0 = rs.randint(-2**31, 2**31, size=(B, 2, 1024), dtype=numpy.int32)
    bara = rs.randint(0, 2048, size=(B, n)).astype(numpy.int32)
    native = _key_with_bk(thr, bk)
    acc = H.dev(thr, acc0); d_bara = H.dev(thr, bara)
    call("nufhe_blind_rotate", thr.handle, native.handle, ptr(acc), ptr(d_bara), n, n, B)
    assert (H.host(acc) == orc.blind_rotate(acc0, bk, bara)).all()


def test_keyswitch_vs_reference_golden(thr, H, golden):
    # test_lwe.py:47-101 of the reference
    from nufhe_amd import lwe as L
    from nufhe_amd.api_low_level import NuFHEParameters, NuFHECloudKey
    from nufhe_amd.bootstrap import NativeCloudKey
    params = NuFHEParameters()
    ks_a, ks_b, ks_cv, src_a, src_b = gi.keyswitch_inputs()
    native = NativeCloudKey(thr, 500)
    ks = L.LweKeyswitchKey(L.HostLweSampleArray(params.in_out_params, ks_a, ks_b, ks_cv))
    NuFHECloudKey._attach_keyswitch(native, ks)
    src = L.LweSampleArray(params.tgsw_params.tlwe_params.extracted_lweparams, H.dev(thr, src_a),
                           H.dev(thr, src_b), thr.zeros(src_b.shape, numpy.float32))
    res = L.LweSampleArray.empty(thr, params.in_out_params, src_b.shape)
    L.lwe_keyswitch(thr, res, ks, src)
    ra, rb, rcv = H.ct_arrays(res)
    assert (ra == golden['ks_a']).all()
    assert (rb == golden['ks_b']).all()
    assert (rcv == golden['ks_cv']).all()


def test_keyswitch_rejects_nonzero_base0(thr):
    from nufhe_amd import lwe as L
    from nufhe_amd.api_low_level import NuFHEParameters, NuFHECloudKey
    from nufhe_amd.bootstrap import NativeCloudKey
    params = NuFHEParameters()
    ks_a = numpy.zeros((1024, 8, 4, 500), numpy.int32); ks_a[3, 1, 0, 7] = 1
    ks = L.LweKeyswitchKey(L.HostLweSampleArray(
        params.in_out_params, ks_a, numpy.zeros((1024, 8, 4), numpy.int32), numpy.zeros((1024, 8, 4), numpy.float32)))
    with pytest.raises(ValueError):
        NuFHECloudKey._attach_keyswitch(NativeCloudKey(thr, 500), ks)


def test_lwe_encrypt_decrypt_vs_reference_golden(thr, H, golden, orc):
    """LweEncrypt / LweDecrypt (test_lwe.py:149-213 of the reference) through nufhe_lwe_phase, and
    the API-level encrypt == the oracle's encrypt from the same seed."""
    msgs, key, na, nb = gi.lwe_encrypt_inputs()
    base = (msgs.astype(numpy.uint32) + nb.astype(numpy.uint32)).view(numpy.int32).reshape(-1)
    d_a = H.dev(thr, na.reshape(-1, 500)); d_base = H.dev(thr, base); d_key = H.dev(thr, key)
    out = thr.array((base.size,), numpy.int32)
    call("nufhe_lwe_phase", thr.handle, ptr(out), 1, ptr(d_a), 500, ptr(d_base), 1, ptr(d_key), 1, base.size, 500)
    b = H.host(out).reshape(msgs.shape)
    assert (b == golden['lwe_encrypt_b']).all()
    d_b = H.dev(thr, b.reshape(-1))
    dec = thr.array((base.size,), numpy.int32)
    call("nufhe_lwe_phase", thr.handle, ptr(dec), 1, ptr(d_a), 500, ptr(d_b), 1, ptr(d_key), -1, base.size, 500)
    assert (H.host(dec).reshape(msgs.shape) == golden['lwe_decrypt']).all()

    import nufhe_amd
    lwe_key = orc.DeterministicRNG(3).uniform_bool((500,))
    sk = H.secret_key_from_array(thr, lwe_key)
    ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(77), thread=thr)
    m = numpy.random.RandomState(1).randint(0, 2, size=(5, 7)).astype(bool)
    ct = ctx.encrypt(sk, m)
    exp = orc.encrypt(orc.DeterministicRNG(77), lwe_key, m)
    a, b2, cv = H.ct_arrays(ct)
    assert (a == exp[0]).all() and (b2 == exp[1]).all() and (cv == exp[2]).all()
    assert (ctx.decrypt(sk, ct) == m).all()
    assert (ctx.decrypt(sk, ct[1:4, ::2]) == m[1:4, ::2]).all()      # strided view


def test_ff_primitives_on_device_vs_bigint(thr, H):
    """GF(P) primitives as compiled for the device (borrow chains, v_mad_u64_u32 carry-out, 129-bit
    lazy dot product) against Python integers, on the reference's boundary values
    (test_arithmetic.py:154-157,172,248) plus random canonical elements."""
    P = 2**64 - 2**32 + 1
    edge = [0, 1, 2, 2**32 - 2, 2**32 - 1, 2**32, 2**32 + 1, 2**33, 2**63, P // 2, P // 2 + 1,
            P - 2**32, P - 2**32 - 1, P - 2, P - 1, 0xFFFFFFFE00000001, 0xFFFFFFFF, 0xFFFFFFFF00000000]
    rs = numpy.random.RandomState(99)
    rnd = [int(x) % P for x in rs.randint(0, 2**63, size=3000, dtype=numpy.int64).astype(object) * 2 + 1]
    vals = edge + rnd

    def run(op, cols, shift=0):
        n = len(cols[0])
        devs = [H.dev(thr, numpy.array(c, dtype=numpy.uint64)) for c in cols]
        out = thr.array((n,), numpy.uint64)
        ptrs = [ptr(d) for d in devs] + [None] * (5 - len(devs))
        call("nufhe_ff_op", thr.handle, ptr(out), *ptrs, op, shift, n)
        return [int(x) for x in H.host_u64(out)]

    # all edge x edge pairs + random pairs
    A = [a for a in edge for _ in edge] + rnd
    B = [b for _ in edge for b in edge] + rnd[::-1]
    assert run(0, [A, B]) == [(a + b) % P for a, b in zip(A, B)]
    assert run(1, [A, B]) == [(a - b) % P for a, b in zip(A, B)]
    assert run(2, [A, B]) == [(a * b) % P for a, b in zip(A, B)]
    C = B[3:] + B[:3]; D = A[7:] + A[:7]; E = A[11:] + A[:11]
    assert run(3, [A, B, C, D]) == [(a * b + c * d) % P for a, b, c, d in zip(A, B, C, D)]
    assert run(4, [A, B, C, D, E]) == [(a * b + c * d + e) % P for a, b, c, d, e in zip(A, B, C, D, E)]
    # maximal operands: the 129-bit accumulator's top bit and the addend's carry
    M = [P - 1] * 4
    assert run(4, [M, M, M, M, M]) == [((P - 1) * (P - 1) * 2 + P - 1) % P] * 4
    # per-element shifts 0..31 and every compile-time shift 0..191 (the reference's lsh family)
    sh = [i % 32 for i in range(len(vals))]
    assert run(5, [vals, sh]) == [(v << s) % P for v, s in zip(vals, sh)]
    for shift in range(192):
        assert run(6, [vals[:64]], shift) == [(v << shift) % P for v in vals[:64]], shift
    # reduce96 accepts ANY 64-bit low word (not only canonical ones)
    lo = [2**64 - 1, 2**64 - 2, P, P + 1, 0, 2**64 - 2**32] + rnd[:500]
    h0 = [2**32 - 1, 2**32 - 1, 1, 0, 2**32 - 1, 2**32 - 1] + [x & 0xFFFFFFFF for x in rnd[500:1000]]
    assert run(7, [lo, h0]) == [(l + (h << 64)) % P for l, h in zip(lo, h0)]
    # int32 conversions (ntt.mako:395-408, test_ntt_cpu.py:60-67)
    ints = [0, 1, -1, 2**31 - 1, -2**31, 12345, -12345]
    got = run(8, [[i & 0xFFFFFFFFFFFFFFFF for i in ints] + vals[:32]])
    for i, g in zip(ints, got):
        assert (g & 0xFFFFFFFF) == (i & 0xFFFFFFFF)
    for v, g in zip(vals[:32], got[len(ints):]):
        signed = v - P if v > P // 2 else v
        assert (g >> 32) == (signed & 0xFFFFFFFF)


def test_l4_primitives_on_device_vs_bigint(thr, H):
    """The redundant 24-bit-limb arithmetic of the blind-rotation transforms (csrc/ff24.h) as compiled
    for gfx950 (v_perm_b32 byte selectors, the carry-out of v_mad_u64_u32 in l4_to_u64, per-lane
    twiddles) against Python integers: the same checks tests/test_emu_l4.py applies to the host build."""
    import l4_checks

    def run(op, a, b=None, c=None, shift=0):
        a = numpy.ascontiguousarray(a, numpy.uint32)
        n = a.shape[0]
        d_a = H.dev(thr, a.view(numpy.int32))
        d_b = None if b is None else H.dev(thr, numpy.ascontiguousarray(b, numpy.uint32).view(numpy.int32))
        d_c = None if c is None else H.dev(thr, numpy.ascontiguousarray(c, numpy.uint32).view(numpy.int32))
        out = thr.array((n, 4), numpy.int32); out2 = thr.array((n, 4), numpy.int32)
        out2.zero_()
        call("nufhe_l4_op", thr.handle, ptr(out), ptr(out2), ptr(d_a), None if d_b is None else ptr(d_b),
             None if d_c is None else ptr(d_c), op, shift, n)
        return H.host(out).view(numpy.uint32), H.host(out2).view(numpy.uint32)

    l4_checks.check_all(run, n=1500)
