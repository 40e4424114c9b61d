"""
GPU tests of the ciphertext container and the non-bootstrapped operations, on DEVICE
tensors: copy / roll / concatenate / slice assignment against numpy on the downloaded arrays (the
reference's test/test_lwe.py:397-511 with the same shapes, shifts and slices), gate_not / gate_copy /
gate_constant at ciphertext level (nufhe/gates.py:292-387: every word of a, b and the variances), strided
and broadcast views through the kernels, the view -> contiguous-temporary route and its log, and the LWE
size checks that guard the raw-pointer C ABI.
"""

import logging

import numpy
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def thr():
    from nufhe_amd.device import DeviceThread
    return DeviceThread(0)


@pytest.fixture(scope='module')
def params():
    from nufhe_amd.api_low_level import NuFHEParameters
    return NuFHEParameters()


def mock_ciphertext(thr, lwe_params, shape, seed=0):
    from nufhe_amd.lwe import LweSampleArray
    rs = numpy.random.RandomState(seed)
    n = lwe_params.size
    a = rs.randint(-2**31, 2**31, size=tuple(shape) + (n,), dtype=numpy.int32)
    b = rs.randint(-2**31, 2**31, size=tuple(shape), dtype=numpy.int32)
    cv = rs.uniform(0, 1, size=tuple(shape)).astype(numpy.float32)
    ct = LweSampleArray(lwe_params, thr.to_device(a), thr.to_device(b), thr.to_device(cv))
    assert ct.a.is_cuda and ct.b.is_cuda and ct.current_variances.is_cuda
    return ct


def get(ct):
    return tuple(x.detach().cpu().numpy() for x in (ct.a, ct.b, ct.current_variances))


def test_copy_on_device(thr, params):
    ct = mock_ciphertext(thr, params.in_out_params, (3, 4, 5))
    cp = ct.copy()
    assert ct == cp and cp.a.is_cuda
    assert cp.a.data_ptr() != ct.a.data_ptr() and cp.b.data_ptr() != ct.b.data_ptr()
    assert cp.current_variances.data_ptr() != ct.current_variances.data_ptr()
    cp.b[0, 0, 0] += 1
    assert ct != cp


@pytest.mark.parametrize('shift', [7, -9, 0])
@pytest.mark.parametrize('axis', [0, 1, -1])
def test_roll_on_device(thr, params, shift, axis):
    ct = mock_ciphertext(thr, params.in_out_params, (3, 4, 5))
    rolled = ct.copy()
    rolled.roll(shift, axis=axis)
    for src, res in zip(get(ct), get(rolled)):
        assert (numpy.roll(src, shift, axis % 3) == res).all()


@pytest.mark.parametrize('axis', [0, 1])
@pytest.mark.parametrize('out_none', [False, True])
def test_concatenate_on_device(thr, params, axis, out_none):
    from nufhe_amd.lwe import concatenate
    shapes = [(3, 4), (1, 4), (4, 4)] if axis == 0 else [(4, 3), (4, 1), (4, 4)]
    cts = [mock_ciphertext(thr, params.in_out_params, s, seed=i) for i, s in enumerate(shapes)]
    out = None if out_none else mock_ciphertext(thr, params.in_out_params, (8, 4) if axis == 0 else (4, 8), seed=9)
    res = concatenate(cts, axis=axis, out=out)
    if not out_none:
        assert res is out
    for k, got in enumerate(get(res)):
        assert (numpy.concatenate([get(c)[k] for c in cts], axis=axis) == got).all()


@pytest.mark.parametrize('case', [
    ((3, 4), (slice(1, None),), (3, 4), (slice(None, -1),)),
    ((10,), (slice(1, 10, 2),), (10,), (slice(None, 10, 2),)),
    ((5,), (1,), (5,), (2,)),
], ids=["contig", "discontig", "scalar"])
def test_assign_on_device(thr, params, case):
    src_shape, src_slice, dst_shape, dst_slice = case
    src = mock_ciphertext(thr, params.in_out_params, src_shape, seed=1)
    dst = mock_ciphertext(thr, params.in_out_params, dst_shape, seed=2)
    ref = [x.copy() for x in get(dst)]
    dst[dst_slice] = src[src_slice]
    for r, s in zip(ref, get(src)):
        r[dst_slice] = s[src_slice]
    for r, g in zip(ref, get(dst)):
        assert (r == g).all()
    with pytest.raises(ValueError):
        dst[dst_slice] = 5


def test_not_copy_constant_ciphertext_level(thr, params):
    """gate_not / gate_copy / gate_constant (nufhe/gates.py:292-387): exact words, incl. the variance
    rule cv_res = p^2 cv_src of LweLinear (lwe_cpu.py:115-143)."""
    import nufhe_amd
    from nufhe_amd import gates
    ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(5), thread=thr)
    ct = mock_ciphertext(thr, params.in_out_params, (4, 7), seed=3)
    a, b, cv = get(ct)
    res = mock_ciphertext(thr, params.in_out_params, (4, 7), seed=4)
    gates.gate_not(thr, None, res, ct)
    ra, rb, rcv = get(res)
    assert (ra == (-a.astype(numpy.int64)).astype(numpy.int32)).all() and (rb == (0 - b.astype(numpy.int64)).astype(numpy.int32)).all()
    assert (rcv == cv).all()
    gates.gate_copy(thr, None, res, ct)
    assert all((x == y).all() for x, y in zip(get(res), (a, b, cv)))
    # broadcast: a (7,) ciphertext copied into every row of a (4, 7) destination
    row = mock_ciphertext(thr, params.in_out_params, (7,), seed=5)
    gates.gate_copy(thr, None, res, row)
    for got, src in zip(get(res), get(row)):
        assert (got == numpy.broadcast_to(src, got.shape)).all()
    # constants: (0, +-1/8), zero variance; values broadcast over the leading axis
    vals = numpy.array([1, 0, 0, 1, 1, 0, 1], bool)
    gates.gate_constant(thr, None, res, vals)
    ra, rb, rcv = get(res)
    assert (ra == 0).all() and (rcv == 0).all()
    assert (rb == numpy.broadcast_to(numpy.where(vals, 2**29, -2**29).astype(numpy.int32), (4, 7))).all()
    with pytest.raises(ValueError):
        gates.gate_constant(thr, None, res, numpy.ones((4,), bool))
    with pytest.raises(ValueError):
        gates.gate_not(thr, None, res, mock_ciphertext(thr, params.in_out_params, (3, 7)))


def test_views_through_the_kernels_and_copy_log(thr, params, caplog):
    """Views that collapse to one bit stride reach the kernels as (pointer, stride) without a copy;
    a view that does not (two independent strides) goes through a contiguous temporary, is written
    back correctly, and says so on the 'nufhe_amd' logger."""
    from nufhe_amd import lwe as L
    src = mock_ciphertext(thr, params.in_out_params, (6, 8), seed=6)
    a, b, cv = get(src)
    before = L.flat_copies
    # stepped slice of the leading axis of a 1-d batch, and a whole contiguous block
    flat = mock_ciphertext(thr, params.in_out_params, (12,), seed=7)
    fa, fb, fcv = get(flat)
    dst = mock_ciphertext(thr, params.in_out_params, (6,), seed=8)
    L.lwe_negate(thr, dst, flat[1::2])
    assert (get(dst)[0] == (-fa[1::2].astype(numpy.int64)).astype(numpy.int32)).all()
    dst2 = mock_ciphertext(thr, params.in_out_params, (6, 8), seed=10)
    L.lwe_copy(thr, dst2, src)
    assert (get(dst2)[0] == a).all()
    assert L.flat_copies == before
    # a column block of a 2-d batch: rows and columns have unrelated strides
    out = mock_ciphertext(thr, params.in_out_params, (6, 8), seed=11)
    oa, ob, ocv = get(out)
    with caplog.at_level(logging.DEBUG, logger="nufhe_amd"):
        L.lwe_copy(thr, out[:, 2:5], src[:, 4:7])
    assert L.flat_copies > before
    assert any("contiguous temporary" in r.message for r in caplog.records)
    oa[:, 2:5] = a[:, 4:7]; ob[:, 2:5] = b[:, 4:7]; ocv[:, 2:5] = cv[:, 4:7]
    for exp, got in zip((oa, ob, ocv), get(out)):
        assert (exp == got).all()


def test_c_boundary_refuses_wrong_operand_sizes(thr):
    """BELOW the Python wrapper: raw ctypes calls into libnufhe_hip.so with hand-made nufhe_lwe descriptors over raw
    device pointers (nufhe_alloc), the way INTEGRATION.md's stub would call it.  A descriptor whose `size` is not what
    the key / operation needs, a NULL pointer, or a stride shorter than the sample must come back as NUFHE_EINVAL
    with a message -- the reference's typed Reikna signatures refuse such calls before launching
    (lwe_gpu.py:151-159, blind_rotate.py:226-234) -- and must not touch memory."""
    import ctypes
    import nufhe_amd
    from nufhe_amd import _lib
    L = _lib.lib()
    ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(11), thread=thr)
    sk, ck = ctx.make_key_pair()
    key = ck._native.handle
    h = thr.handle
    nbits, n, ext = 4, 500, 1024

    def alloc(nbytes):
        p = ctypes.c_void_p()
        assert L.nufhe_alloc(h, nbytes, ctypes.byref(p)) == 0
        return p
    bufs = [alloc(nbits * ext * 4) for _ in range(8)]

    def desc(i, size, a_stride=None, null=None):
        d = _lib.NufheLwe(a=bufs[2 * i].value, b=bufs[2 * i + 1].value, cv=bufs[2 * i + 1].value,
                          a_stride=size if a_stride is None else a_stride, b_stride=1, size=size)
        if null:
            setattr(d, null, None)
        return d
    EINVAL = -1

    def refused(rc, text):
        assert rc == EINVAL, rc
        msg = L.nufhe_last_error().decode()
        assert text in msg, msg
    MU = 2**29
    ok = desc(0, n)
    refused(L.nufhe_gate_binary(h, key, desc(0, ext), desc(1, n), desc(2, n), MU, -1, -1, MU, nbits), "gate result: LWE size 1024, expected 500")
    refused(L.nufhe_gate_binary(h, key, ok, desc(1, ext), desc(2, n), MU, -1, -1, MU, nbits), "gate operand a: LWE size 1024")
    refused(L.nufhe_gate_binary(h, key, ok, desc(1, n), desc(2, 0), MU, -1, -1, MU, nbits), "gate operand b: LWE size 0")
    refused(L.nufhe_gate_binary(h, key, ok, desc(1, n, a_stride=499), desc(2, n), MU, -1, -1, MU, nbits), "a_stride 499 shorter")
    refused(L.nufhe_gate_binary(h, key, ok, desc(1, n, null='a'), desc(2, n), MU, -1, -1, MU, nbits), "NULL a / b")
    refused(L.nufhe_gate_binary(h, key, desc(0, n, null='cv'), desc(1, n), desc(2, n), MU, -1, -1, MU, nbits), "NULL variance")
    refused(L.nufhe_gate_mux(h, key, ok, desc(1, n), desc(2, n), desc(3, ext), nbits), "mux operand c: LWE size 1024")
    refused(L.nufhe_bootstrap(h, key, desc(0, n), desc(1, n), MU, nbits, 1), "bootstrap result: LWE size 500, expected 1024")
    refused(L.nufhe_bootstrap(h, key, desc(0, ext), desc(1, n), MU, nbits, 0), "bootstrap result: LWE size 1024, expected 500")
    refused(L.nufhe_bootstrap(h, key, desc(0, n), desc(1, ext), MU, nbits, 0), "bootstrap input: LWE size 1024")
    refused(L.nufhe_keyswitch(h, key, desc(0, n), desc(1, n), nbits), "keyswitch source: LWE size 500, expected 1024")
    refused(L.nufhe_keyswitch(h, key, desc(0, ext), desc(1, ext), nbits), "keyswitch result: LWE size 1024, expected 500")
    refused(L.nufhe_lwe_linear(h, desc(0, n), desc(1, ext), 1, 0, nbits, n), "lwe_linear source: LWE size 1024")
    refused(L.nufhe_lwe_trivial_const(h, desc(0, ext), MU, nbits, n), "lwe_trivial_const result: LWE size 1024")
    # a RESULT of several bits with a broadcast (zero) stride would have every bit's work-group write the same row
    def bcast(i, size, which):
        d = desc(i, size)
        setattr(d, which, 0)
        return d
    refused(L.nufhe_gate_binary(h, key, bcast(0, n, 'a_stride'), desc(1, n), desc(2, n), MU, -1, -1, MU, nbits), "gate result: zero (broadcast) stride")
    refused(L.nufhe_gate_binary(h, key, bcast(0, n, 'b_stride'), desc(1, n), desc(2, n), MU, -1, -1, MU, nbits), "gate result: zero (broadcast) stride")
    refused(L.nufhe_gate_mux(h, key, bcast(0, n, 'a_stride'), desc(1, n), desc(2, n), desc(3, n), nbits), "mux result: zero (broadcast) stride")
    refused(L.nufhe_bootstrap(h, key, bcast(0, ext, 'b_stride'), desc(1, n), MU, nbits, 1), "bootstrap result: zero (broadcast) stride")
    refused(L.nufhe_keyswitch(h, key, bcast(0, n, 'a_stride'), desc(3, ext), nbits), "keyswitch result: zero (broadcast) stride")
    refused(L.nufhe_lwe_linear(h, bcast(0, n, 'a_stride'), desc(1, n), 1, 0, nbits, n), "lwe_linear result: zero (broadcast) stride")
    refused(L.nufhe_lwe_trivial_const(h, bcast(0, n, 'b_stride'), MU, nbits, n), "lwe_trivial_const result: zero (broadcast) stride")
    # ... a broadcast OPERAND is fine (one ciphertext against a batch), and so is a one-bit result with any stride
    assert L.nufhe_lwe_trivial_const(h, desc(1, n), MU, nbits, n) == 0
    assert L.nufhe_gate_binary(h, key, ok, bcast(1, n, 'a_stride'), desc(1, n), MU, -1, -1, MU, nbits) == 0
    assert L.nufhe_gate_binary(h, key, bcast(0, n, 'a_stride'), desc(1, n), desc(1, n), MU, -1, -1, MU, 1) == 0
    assert L.nufhe_abi_version() == _lib.ABI_VERSION
    # a negative batch is refused as before, and the well-formed calls run
    refused(L.nufhe_gate_binary(h, key, ok, desc(1, n), desc(2, n), MU, -1, -1, MU, -1), "negative batch")
    assert L.nufhe_lwe_trivial_const(h, desc(1, n), MU, nbits, n) == 0
    assert L.nufhe_lwe_trivial_const(h, desc(2, n), 0, nbits, n) == 0
    assert L.nufhe_gate_binary(h, key, ok, desc(1, n), desc(2, n), MU, -1, -1, MU, nbits) == 0
    assert L.nufhe_bootstrap(h, key, desc(3, ext), desc(1, n), MU, nbits, 1) == 0
    assert L.nufhe_keyswitch(h, key, desc(0, n), desc(3, ext), nbits) == 0
    thr.synchronize()
    for b in bufs:
        assert L.nufhe_free(h, b) == 0


def test_lwe_size_validation(thr, params):
    """Operands whose LWE size does not match the key are rejected in Python (the kernels would read or
    write out of bounds): gates, bootstrap with / without keyswitch, keyswitch, decrypt."""
    import nufhe_amd
    from nufhe_amd import gates
    from nufhe_amd.bootstrap import bootstrap
    from nufhe_amd.lwe import LweSampleArray, lwe_keyswitch
    ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(11), thread=thr)
    sk, ck = ctx.make_key_pair()
    n_params = params.in_out_params
    ext_params = params.tgsw_params.tlwe_params.extracted_lweparams
    small = LweSampleArray.empty(thr, n_params, (4,)); small.a.zero_(); small.b.zero_(); small.current_variances.zero_()
    big = LweSampleArray.empty(thr, ext_params, (4,)); big.a.zero_(); big.b.zero_(); big.current_variances.zero_()
    with pytest.raises(ValueError):
        gates.gate_nand(thr, ck, big, small, small)            # extracted-size result
    with pytest.raises(ValueError):
        gates.gate_nand(thr, ck, small, big, small)            # extracted-size argument
    with pytest.raises(ValueError):
        gates.gate_mux(thr, ck, small, small, small, big)
    with pytest.raises(ValueError):
        bootstrap(thr, small, ck.bootstrap_key, ck.keyswitch_key, 2**29, small, no_keyswitch=True)
    with pytest.raises(ValueError):
        bootstrap(thr, big, ck.bootstrap_key, ck.keyswitch_key, 2**29, small, no_keyswitch=False)
    with pytest.raises(ValueError):
        lwe_keyswitch(thr, small, ck.keyswitch_key, small)     # source must be LWE(1024)
    with pytest.raises(ValueError):
        lwe_keyswitch(thr, big, ck.keyswitch_key, big)         # result must be LWE(500)
    with pytest.raises(ValueError):
        ctx.decrypt(sk, big)
    # the valid combinations still work
    bootstrap(thr, big, ck.bootstrap_key, ck.keyswitch_key, 2**29, small, no_keyswitch=True)
    lwe_keyswitch(thr, small, ck.keyswitch_key, big)
    gates.gate_nand(thr, ck, small, small, small)
