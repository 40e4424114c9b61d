"""
Integer operators composed of bootstrapped gates (reference: nufhe/operators_integer.py:29-95).

``uint_min`` is the reference's example of a gate *circuit*: a ripple comparator, one XNOR and one
MUX per bit position.  All intermediate ciphertexts stay in HBM and every gate is one fused launch.  The
reference issues the 2 x itemsize gates one after the other on one-bit-wide slices; here the XNORs, which do
not depend on the carry, are hoisted into ONE gate over the whole array (a batch that fills the chip instead
of itemsize narrow ones), and only the carry chain of MUXes stays sequential.  A gate's output depends on
nothing but its inputs, so every ciphertext is bit-identical to the reference's schedule
(tests/test_operators_integer.py); (128, 32): 466 -> ~310 ms.  Bits are big-endian along the last axis, as in
the reference.
"""

import numpy

from .api_low_level import empty_ciphertext
from .gates import gate_constant, gate_xnor, gate_mux, gate_batch


def _uint_to_bits(x, bitsize):
    return numpy.array([((int(x) >> i) & 1) != 0 for i in reversed(range(bitsize))])


def uintarray_to_bitarray(xs, itemsize=None):
    """Unsigned integers -> booleans ``xs.shape + (itemsize,)``, most significant bit first
    (operators_integer.py:41-46)."""
    xs = numpy.asarray(xs)
    assert numpy.issubdtype(xs.dtype, numpy.unsignedinteger)
    if itemsize is None:
        itemsize = xs.itemsize * 8
    shifts = numpy.arange(itemsize - 1, -1, -1, dtype=numpy.uint64)
    bits = (xs.astype(numpy.uint64)[..., None] >> shifts) & numpy.uint64(1)
    return bits.astype(bool)


def bitarray_to_uintarray(xs):
    """Inverse of :func:`uintarray_to_bitarray`; the last axis must be 8, 16, 32 or 64 long
    (operators_integer.py:49-63)."""
    xs = numpy.asarray(xs).astype(bool)
    itemsize = xs.shape[-1]
    dtype = {8: numpy.uint8, 16: numpy.uint16, 32: numpy.uint32, 64: numpy.uint64}[itemsize]
    shifts = numpy.arange(itemsize - 1, -1, -1, dtype=numpy.uint64)
    vals = (xs.astype(numpy.uint64) << shifts).sum(axis=-1, dtype=numpy.uint64)
    return vals.astype(dtype)


def uint_min(thread, cloud_key, answer, a, b, perf_params=None):
    """
    answer = min(a, b) element-wise for encrypted unsigned integers stored as big-endian bit
    arrays ``shape + (itemsize,)``  (operators_integer.py:66-95).

    Walks from the least significant bit: ``carry = (a_i == b_i) ? carry : a_i`` leaves, after the
    most significant position, carry = 1 iff b < a; then ``answer = carry ? b : a``.
    """
    params = cloud_key.params
    itemsize = answer.shape[-1]
    lead = tuple(a.shape[:-1])
    carry = empty_ciphertext(thread, params, lead + (1,))
    same = empty_ciphertext(thread, params, lead + (itemsize,))
    gate_constant(thread, cloud_key, carry, False)
    gate_xnor(thread, cloud_key, same, a, b, perf_params=perf_params)          # every position at once
    for i in reversed(range(itemsize)):
        gate_mux(thread, cloud_key, carry, same[..., i:i + 1], carry, a[..., i:i + 1], perf_params=perf_params)
    gate_mux(thread, cloud_key, answer, carry, b, a, perf_params=perf_params)


def uint_min_many(thread, cloud_key, answers, a_list, b_list, perf_params=None):
    """
    ``uint_min`` for SEVERAL independent operand pairs at once -- different shapes and buffers allowed -- with every step
    of all comparators issued as ONE heterogeneous gate batch (:func:`nufhe_amd.gates.gate_batch`): the XNORs of all
    circuits in one launch, then ``itemsize`` launches for the carry chains (one MUX per circuit each), then one launch for
    the final selections.  A narrow comparator occupies a fraction of the chip and every gate costs a full blind rotation
    however few bits it has, so C circuits in lock step take about the time of one (four (4, 16)-bit circuits: 331 -> 88 ms).
    Every ciphertext equals what :func:`uint_min` writes for the same pair (no reference counterpart; the circuit is
    nufhe/operators_integer.py:66-95).  All operands must have the same ``itemsize`` (last axis).
    """
    params = cloud_key.params
    count = len(answers)
    if not (count == len(a_list) == len(b_list)):
        raise ValueError("uint_min_many: %d answers, %d / %d operands" % (count, len(a_list), len(b_list)))
    if count == 0:
        return
    itemsize = answers[0].shape[-1]
    if any(x.shape[-1] != itemsize for x in list(answers) + list(a_list) + list(b_list)):
        raise ValueError("uint_min_many: all operands must have the same number of bits per integer")
    carries, sames = [], []
    for a in a_list:
        lead = tuple(a.shape[:-1])
        carry = empty_ciphertext(thread, params, lead + (1,))
        gate_constant(thread, cloud_key, carry, False)
        carries.append(carry)
        sames.append(empty_ciphertext(thread, params, lead + (itemsize,)))
    gate_batch(thread, cloud_key, [('gate_xnor', sames[k], a_list[k], b_list[k]) for k in range(count)], perf_params)
    for i in reversed(range(itemsize)):
        gate_batch(thread, cloud_key, [('gate_mux', carries[k], sames[k][..., i:i + 1], carries[k], a_list[k][..., i:i + 1])
                                       for k in range(count)], perf_params)
    gate_batch(thread, cloud_key, [('gate_mux', answers[k], carries[k], b_list[k], a_list[k]) for k in range(count)],
               perf_params)
