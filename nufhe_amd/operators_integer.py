"""
Integer operators composed of bootstrapped gates (reference: nufhe/operators_integer.py:29-95).

``uint_min`` is the reference's example of a gate *circuit*: a ripple comparator, one XNOR and one
MUX per bit position, executed as a chain on the device.  All intermediate ciphertexts stay in HBM;
every gate of the chain is one fused launch (the bootstrap kernel sizes its work-groups to the
batch, so the narrow slices of the chain are spread over all CUs).  Bits are big-endian along the
last axis, as in the reference.
"""

import numpy

from .api_low_level import empty_ciphertext
from .gates import gate_constant, gate_xnor, gate_mux


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
    same = empty_ciphertext(thread, params, lead + (1,))
    gate_constant(thread, cloud_key, carry, False)
    for i in reversed(range(itemsize)):
        a_bit = a[..., i:i + 1]
        b_bit = b[..., i:i + 1]
        gate_xnor(thread, cloud_key, same, a_bit, b_bit, perf_params=perf_params)
        gate_mux(thread, cloud_key, carry, same, carry, a_bit, perf_params=perf_params)
    gate_mux(thread, cloud_key, answer, carry, b, a, perf_params=perf_params)
