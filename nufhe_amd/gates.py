"""
Bootstrapped logic gates on encrypted bit arrays (reference: nufhe/gates.py:42-664).

Same names, argument order, broadcasting rules and error behaviour as the reference.  Each binary
gate is `result = KS(BS((0, c) + pa * a + pb * b))`; the linear pre-combination,
mod-switch, blind rotation, sample extraction and keyswitch of a gate are a single C-ABI call
(`nufhe_gate_binary` / `nufhe_gate_mux`), not the reference's chain of ~10 kernel launches.
"""

import numpy

from . import _lib
from .lwe import (
    LweSampleArray, _Flat, check_lwe_size,
    lwe_negate, lwe_copy, lwe_noiseless_trivial,
    )
from .numeric_functions import phase_to_t32


def get_shape(obj):
    """Shape of a plaintext / ciphertext argument: anything with a ``shape``, or a (nested) list
    (reference behaviour: nufhe/gates.py:42-48)."""
    shape = getattr(obj, 'shape', None)
    if shape is not None:
        return tuple(shape)
    if isinstance(obj, list):
        return numpy.asarray(obj).shape
    raise ValueError("%s objects have no shape: expected an array-like argument" % type(obj).__name__)


def result_shape(*shapes):
    """Broadcast of the operand shapes, aligned at the LAST axis (reference: nufhe/gates.py:51-69).  An axis
    of length <= 1 stretches to the other operand's length; two different lengths above 1 do not combine."""
    out = ()
    for shape in shapes:
        shape = tuple(shape)
        rank = max(len(out), len(shape))
        a = (1,) * (rank - len(out)) + out
        b = (1,) * (rank - len(shape)) + shape
        merged = []
        for axis, (x, y) in enumerate(zip(a, b)):
            if x > 1 and y > 1 and x != y:
                raise ValueError("operand shapes %s do not broadcast: lengths %d and %d meet on axis %d (from the left of "
                                 "the aligned shapes)" % (", ".join(str(tuple(s)) for s in shapes), x, y, axis))
            merged.append(x if x > 1 else y)
        out = tuple(merged)
    return out


def check_shape(result, *args):
    """The broadcast shape of the arguments must be the trailing part of the destination's shape
    (leading destination axes repeat the gate; reference: nufhe/gates.py:72-78)."""
    need = result_shape(*[get_shape(arg) for arg in args])
    have = tuple(result.shape)
    if len(need) > len(have) or have[len(have) - len(need):] != need:
        raise ValueError("gate arguments broadcast to %s, which is not a trailing part of the destination shape %s"
                         % (need, have))


MU = phase_to_t32(1, 8)


def _check_sizes(cloud_key, result, *args):
    n = cloud_key.params.in_out_params.size
    check_lwe_size("gate result", result, n)
    for x in args:
        check_lwe_size("gate argument", x, n)


def _binary_gate(thr, cloud_key, result, a, b, c, pa, pb):
    check_shape(result, a, b)
    _check_sizes(cloud_key, result, a, b)
    thr.check_stream()
    res = _Flat(result, result.shape, output=True)
    fa = _Flat(a, result.shape)
    fb = _Flat(b, result.shape)
    _lib.call("nufhe_gate_binary", thr.handle, cloud_key._native.handle, res.desc, fa.desc, fb.desc,
              int(c), int(pa), int(pb), int(MU), res.nbits)
    res.writeback()


def gate_nand(thr, cloud_key, result, a, b, perf_params=None):
    """Homomorphic bootstrapped NAND: (0, 1/8) - a - b  (nufhe/gates.py:81-121)."""
    _binary_gate(thr, cloud_key, result, a, b, phase_to_t32(1, 8), -1, -1)


def gate_or(thr, cloud_key, result, a, b, perf_params=None):
    """Homomorphic bootstrapped OR: (0, 1/8) + a + b  (nufhe/gates.py:124-163)."""
    _binary_gate(thr, cloud_key, result, a, b, phase_to_t32(1, 8), 1, 1)


def gate_and(thr, cloud_key, result, a, b, perf_params=None):
    """Homomorphic bootstrapped AND: (0, -1/8) + a + b  (nufhe/gates.py:166-205)."""
    _binary_gate(thr, cloud_key, result, a, b, phase_to_t32(-1, 8), 1, 1)


def gate_xor(thr, cloud_key, result, a, b, perf_params=None):
    """Homomorphic bootstrapped XOR: (0, 1/4) + 2 (a + b)  (nufhe/gates.py:208-247)."""
    _binary_gate(thr, cloud_key, result, a, b, phase_to_t32(1, 4), 2, 2)


def gate_xnor(thr, cloud_key, result, a, b, perf_params=None):
    """Homomorphic bootstrapped XNOR: (0, -1/4) - 2 (a + b)  (nufhe/gates.py:250-289)."""
    _binary_gate(thr, cloud_key, result, a, b, phase_to_t32(-1, 4), -2, -2)


def gate_not(thr, cloud_key, result, a, perf_params=None):
    """Homomorphic NOT (no bootstrap: negation)  (nufhe/gates.py:292-309)."""
    check_shape(result, a)
    lwe_negate(thr, result, a)


def gate_copy(thr, cloud_key, result, a, perf_params=None):
    """Homomorphic COPY (no bootstrap)  (nufhe/gates.py:312-329)."""
    check_shape(result, a)
    lwe_copy(thr, result, a)


def gate_constant(thr, cloud_key, result, vals, perf_params=None):
    """Trivial encryptions of plaintext bits  (nufhe/gates.py:332-387)."""
    vals = numpy.asarray(vals)
    if len(vals.shape) > len(result.shape) or vals.shape != result.shape[len(result.shape)-len(vals.shape):]:
        raise ValueError(
            "The shape of the values {vshape} cannot be broadcasted to the shape of the destination {rshape}".format(
                vshape=vals.shape, rshape=result.shape))
    mus = numpy.where(vals.astype(bool), MU, -MU).astype(numpy.int32)
    lwe_noiseless_trivial(thr, result, mus)


def gate_nor(thr, cloud_key, result, a, b, perf_params=None):
    """Homomorphic bootstrapped NOR: (0, -1/8) - a - b  (nufhe/gates.py:390-429)."""
    _binary_gate(thr, cloud_key, result, a, b, phase_to_t32(-1, 8), -1, -1)


def gate_andny(thr, cloud_key, result, a, b, perf_params=None):
    """Homomorphic bootstrapped AND(NOT a, b): (0, -1/8) - a + b  (nufhe/gates.py:432-471)."""
    _binary_gate(thr, cloud_key, result, a, b, phase_to_t32(-1, 8), -1, 1)


def gate_andyn(thr, cloud_key, result, a, b, perf_params=None):
    """Homomorphic bootstrapped AND(a, NOT b): (0, -1/8) + a - b  (nufhe/gates.py:474-513)."""
    _binary_gate(thr, cloud_key, result, a, b, phase_to_t32(-1, 8), 1, -1)


def gate_orny(thr, cloud_key, result, a, b, perf_params=None):
    """Homomorphic bootstrapped OR(NOT a, b): (0, 1/8) - a + b  (nufhe/gates.py:516-555)."""
    _binary_gate(thr, cloud_key, result, a, b, phase_to_t32(1, 8), -1, 1)


def gate_oryn(thr, cloud_key, result, a, b, perf_params=None):
    """Homomorphic bootstrapped OR(a, NOT b): (0, 1/8) + a - b  (nufhe/gates.py:558-597)."""
    _binary_gate(thr, cloud_key, result, a, b, phase_to_t32(1, 8), 1, -1)


def gate_mux(thr, cloud_key, result, a, b, c, perf_params=None):
    """
    Homomorphic bootstrapped MUX (``b if a else c``): two bootstraps without keyswitch,
    (0, 1/8) + u1 + u2, one keyswitch  (nufhe/gates.py:600-664).
    """
    check_shape(result, a, b, c)
    _check_sizes(cloud_key, result, a, b, c)
    thr.check_stream()
    res = _Flat(result, result.shape, output=True)
    fa = _Flat(a, result.shape)
    fb = _Flat(b, result.shape)
    fc = _Flat(c, result.shape)
    _lib.call("nufhe_gate_mux", thr.handle, cloud_key._native.handle, res.desc, fa.desc, fb.desc, fc.desc,
              res.nbits)
    res.writeback()
