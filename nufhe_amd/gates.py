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
    """nufhe/gates.py:42-48"""
    if hasattr(obj, 'shape'):
        return tuple(obj.shape)
    elif isinstance(obj, list):
        return numpy.asarray(obj).shape
    else:
        raise ValueError("An object of type " + str(type(obj)) + " is not array-like")


def _result_shape_pair(shape1, shape2):
    if len(shape1) > len(shape2):
        shape2 = (1,) * (len(shape1) - len(shape2)) + shape2
    else:
        shape1 = (1,) * (len(shape2) - len(shape1)) + shape1
    if any((l1 != l2 and l1 > 1 and l2 > 1) for l1, l2 in zip(shape1, shape2)):
        raise ValueError("Incompatible shapes: {s1}, {s2}".format(s1=shape1, s2=shape2))
    return tuple((l1 if l1 > 1 else l2) for l1, l2 in zip(shape1, shape2))


def result_shape(*shapes):
    """nufhe/gates.py:63-69"""
    shapes = [tuple(s) for s in shapes]
    if len(shapes) == 1:
        return shapes[0]
    elif len(shapes) == 2:
        return _result_shape_pair(*shapes)
    else:
        return _result_shape_pair(shapes[0], result_shape(*shapes[1:]))


def check_shape(result, *args):
    """nufhe/gates.py:72-78"""
    rshape = result_shape(*[arg.shape for arg in args])
    if len(rshape) > len(result.shape) or rshape != result.shape[len(result.shape)-len(rshape):]:
        raise ValueError(
            ("The shape of the result derived from the arguments {derived_shape} "
            "cannot be broadcasted to the shape of the destination {dest_shape}").format(
            derived_shape=rshape, dest_shape=result.shape))


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
