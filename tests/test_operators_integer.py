"""
Integer operators over gate chains (reference: nufhe/operators_integer.py, test/test_gates.py:170-176,
248-249: uint_min on shape (4, 16) against numpy.minimum).
"""

import numpy
import pytest


def test_bit_conversions_roundtrip_and_order():
    from nufhe_amd.operators_integer import uintarray_to_bitarray, bitarray_to_uintarray
    xs = numpy.array([[0, 1, 0x8000, 0xFFFF], [0x1234, 0xBEEF, 2, 0x7FFF]], numpy.uint16)
    bits = uintarray_to_bitarray(xs)
    assert bits.shape == (2, 4, 16) and bits.dtype == bool
    assert bits[0, 1].tolist() == [False] * 15 + [True]          # big-endian: LSB is the last entry
    assert bits[0, 2].tolist() == [True] + [False] * 15
    assert (bitarray_to_uintarray(bits) == xs).all() and bitarray_to_uintarray(bits).dtype == numpy.uint16
    for dt in (numpy.uint8, numpy.uint32, numpy.uint64):
        v = numpy.random.RandomState(3).randint(0, 2**63, size=(5,), dtype=numpy.uint64).astype(dt)
        assert (bitarray_to_uintarray(uintarray_to_bitarray(v)) == v).all()
    # narrower item size than the dtype (operators_integer.py:41-46, itemsize argument)
    assert uintarray_to_bitarray(numpy.array([5], numpy.uint8), itemsize=3).tolist() == [[True, False, True]]
    # the module is reachable under the reference's name
    import nufhe.operators_integer as alias
    assert alias.uint_min.__module__ == 'nufhe_amd.operators_integer'


@pytest.mark.gpu
def test_uint_min_gate_chain():
    """test/test_gates.py:248-249: shape (4, 16), result == numpy.minimum on the decrypted integers."""
    import nufhe_amd
    from nufhe_amd.operators_integer import uint_min, uintarray_to_bitarray, bitarray_to_uintarray
    ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(77))
    sk, ck = ctx.make_key_pair()
    rs = numpy.random.RandomState(5)
    x = rs.randint(0, 2**16, size=4).astype(numpy.uint16)
    y = rs.randint(0, 2**16, size=4).astype(numpy.uint16)
    x[3] = y[3]                                                   # the equal case keeps carry = 0 -> a
    y[2] = x[2] ^ 1                                               # differ in the last bit only
    ca = ctx.encrypt(sk, uintarray_to_bitarray(x)); cb = ctx.encrypt(sk, uintarray_to_bitarray(y))
    answer = nufhe_amd.empty_ciphertext(ctx.thread, ck.params, (4, 16))
    uint_min(ctx.thread, ck, answer, ca, cb)
    got = bitarray_to_uintarray(ctx.decrypt(sk, answer))
    assert (got == numpy.minimum(x, y)).all()

    # the reference's schedule (operators_integer.py:66-95: XNOR and MUX alternate on one-bit slices) gives the same
    # ciphertext, word for word, as the hoisted XNOR of nufhe_amd.operators_integer.uint_min
    from nufhe_amd.gates import gate_constant, gate_xnor, gate_mux
    thr = ctx.thread
    carry = nufhe_amd.empty_ciphertext(thr, ck.params, (4, 1))
    same = nufhe_amd.empty_ciphertext(thr, ck.params, (4, 1))
    ref = nufhe_amd.empty_ciphertext(thr, ck.params, (4, 16))
    gate_constant(thr, ck, carry, False)
    for i in reversed(range(16)):
        gate_xnor(thr, ck, same, ca[..., i:i + 1], cb[..., i:i + 1])
        gate_mux(thr, ck, carry, same, carry, ca[..., i:i + 1])
    gate_mux(thr, ck, ref, carry, cb, ca)
    import torch
    assert torch.equal(ref.a, answer.a) and torch.equal(ref.b, answer.b)
    assert torch.equal(ref.current_variances, answer.current_variances)


@pytest.mark.gpu
def test_uint_min_many_equals_uint_min_per_pair():
    """uint_min_many (every step of several independent comparators as one heterogeneous gate batch): the same ciphertext
    words as uint_min on each pair, for pairs of different leading shapes; decrypted == numpy.minimum."""
    import nufhe_amd
    from nufhe_amd.operators_integer import uint_min, uint_min_many, uintarray_to_bitarray, bitarray_to_uintarray
    ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(78))
    sk, ck = ctx.make_key_pair()
    rs = numpy.random.RandomState(6)
    shapes = [(3,), (1,), (2, 2)]
    xs = [rs.randint(0, 2**8, size=s).astype(numpy.uint8) for s in shapes]
    ys = [rs.randint(0, 2**8, size=s).astype(numpy.uint8) for s in shapes]
    ca = [ctx.encrypt(sk, uintarray_to_bitarray(x)) for x in xs]
    cb = [ctx.encrypt(sk, uintarray_to_bitarray(y)) for y in ys]
    many = [nufhe_amd.empty_ciphertext(ctx.thread, ck.params, s + (8,)) for s in shapes]
    uint_min_many(ctx.thread, ck, many, ca, cb)
    for k, s in enumerate(shapes):
        one = nufhe_amd.empty_ciphertext(ctx.thread, ck.params, s + (8,))
        uint_min(ctx.thread, ck, one, ca[k], cb[k])
        assert one == many[k]
        assert (bitarray_to_uintarray(ctx.decrypt(sk, many[k])) == numpy.minimum(xs[k], ys[k])).all()
    with pytest.raises(ValueError):
        uint_min_many(ctx.thread, ck, many[:2], ca, cb)
