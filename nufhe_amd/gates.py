"""
Bootstrapped logic gates on encrypted bit arrays (reference: nufhe/gates.py:42-664).

Same names, argument order, broadcasting rules and error behaviour as the reference.  Each binary
gate is `result = KS(BS((0, c) + pa * a + pb * b))`; the linear pre-combination,
mod-switch, blind rotation, sample extraction and keyswitch of a gate are a single C-ABI call
(`nufhe_gate_binary` / `nufhe_gate_mux`), not the reference's chain of ~10 kernel launches.
With ``perf_params.single_kernel_bootstrap == False`` a gate instead runs the reference's own sequence -- trivial
constant, linear combinations, `bootstrap` through the step-by-step driver of nufhe_amd/bootstrap.py -- one launch
per step (the reference's multi-kernel mode, gates.py:104-121 / bootstrap.py:96-196).
"""

import numpy

from . import _lib
from .bootstrap import bootstrap, single_kernel
from .lwe import (
    LweSampleArray, _Flat, check_lwe_size,
    lwe_negate, lwe_copy, lwe_noiseless_trivial, lwe_noiseless_trivial_constant,
    lwe_add_to, lwe_sub_to, lwe_add_mul_to, lwe_sub_mul_to, lwe_keyswitch,
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


def _accumulate(thr, result, p, source):
    """result += p * source with the reference's choice of primitive (add / sub for +-1, the multiplying forms else)"""
    if p == 1:
        lwe_add_to(thr, result, source)
    elif p == -1:
        lwe_sub_to(thr, result, source)
    elif p > 0:
        lwe_add_mul_to(thr, result, p, source)
    else:
        lwe_sub_mul_to(thr, result, -p, source)


def _binary_gate_stepwise(thr, cloud_key, result, a, b, c, pa, pb, perf_params):
    """(0, c) + pa a + pb b in a temporary, then `bootstrap` (nufhe/gates.py:104-121 and siblings)"""
    temp = LweSampleArray.empty(thr, cloud_key.params.in_out_params, result.shape)
    lwe_noiseless_trivial_constant(thr, temp, c)
    _accumulate(thr, temp, pa, a)
    _accumulate(thr, temp, pb, b)
    bootstrap(thr, result, cloud_key.bootstrap_key, cloud_key.keyswitch_key, MU, temp, perf_params)


def _binary_gate(thr, cloud_key, result, a, b, c, pa, pb, perf_params=None):
    check_shape(result, a, b)
    _check_sizes(cloud_key, result, a, b)
    thr.check_stream()
    if not single_kernel(perf_params):
        return _binary_gate_stepwise(thr, cloud_key, result, a, b, int(c), int(pa), int(pb), perf_params)
    res = _Flat(result, result.shape, output=True)
    fa = _Flat(a, result.shape)
    fb = _Flat(b, result.shape)
    _lib.call("nufhe_gate_binary", thr.handle, cloud_key._native.handle, res.desc, fa.desc, fb.desc,
              int(c), int(pa), int(pb), int(MU), res.nbits)
    res.writeback()


def gate_nand(thr, cloud_key, result, a, b, perf_params=None):
    """Homomorphic bootstrapped NAND: (0, 1/8) - a - b  (nufhe/gates.py:81-121)."""
    _binary_gate(thr, cloud_key, result, a, b, phase_to_t32(1, 8), -1, -1, perf_params)


def gate_or(thr, cloud_key, result, a, b, perf_params=None):
    """Homomorphic bootstrapped OR: (0, 1/8) + a + b  (nufhe/gates.py:124-163)."""
    _binary_gate(thr, cloud_key, result, a, b, phase_to_t32(1, 8), 1, 1, perf_params)


def gate_and(thr, cloud_key, result, a, b, perf_params=None):
    """Homomorphic bootstrapped AND: (0, -1/8) + a + b  (nufhe/gates.py:166-205)."""
    _binary_gate(thr, cloud_key, result, a, b, phase_to_t32(-1, 8), 1, 1, perf_params)


def gate_xor(thr, cloud_key, result, a, b, perf_params=None):
    """Homomorphic bootstrapped XOR: (0, 1/4) + 2 (a + b)  (nufhe/gates.py:208-247)."""
    _binary_gate(thr, cloud_key, result, a, b, phase_to_t32(1, 4), 2, 2, perf_params)


def gate_xnor(thr, cloud_key, result, a, b, perf_params=None):
    """Homomorphic bootstrapped XNOR: (0, -1/4) - 2 (a + b)  (nufhe/gates.py:250-289)."""
    _binary_gate(thr, cloud_key, result, a, b, phase_to_t32(-1, 4), -2, -2, perf_params)


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
    if vals.ndim == 0:
        # one constant for the whole array: filled on the device, nothing crosses PCIe (and the call can be part of a
        # captured circuit, nufhe_amd/graph.py)
        lwe_noiseless_trivial_constant(thr, result, MU if bool(vals) else -MU)
        return
    mus = numpy.where(vals.astype(bool), MU, -MU).astype(numpy.int32)
    lwe_noiseless_trivial(thr, result, mus)


def gate_nor(thr, cloud_key, result, a, b, perf_params=None):
    """Homomorphic bootstrapped NOR: (0, -1/8) - a - b  (nufhe/gates.py:390-429)."""
    _binary_gate(thr, cloud_key, result, a, b, phase_to_t32(-1, 8), -1, -1, perf_params)


def gate_andny(thr, cloud_key, result, a, b, perf_params=None):
    """Homomorphic bootstrapped AND(NOT a, b): (0, -1/8) - a + b  (nufhe/gates.py:432-471)."""
    _binary_gate(thr, cloud_key, result, a, b, phase_to_t32(-1, 8), -1, 1, perf_params)


def gate_andyn(thr, cloud_key, result, a, b, perf_params=None):
    """Homomorphic bootstrapped AND(a, NOT b): (0, -1/8) + a - b  (nufhe/gates.py:474-513)."""
    _binary_gate(thr, cloud_key, result, a, b, phase_to_t32(-1, 8), 1, -1, perf_params)


def gate_orny(thr, cloud_key, result, a, b, perf_params=None):
    """Homomorphic bootstrapped OR(NOT a, b): (0, 1/8) - a + b  (nufhe/gates.py:516-555)."""
    _binary_gate(thr, cloud_key, result, a, b, phase_to_t32(1, 8), -1, 1, perf_params)


def gate_oryn(thr, cloud_key, result, a, b, perf_params=None):
    """Homomorphic bootstrapped OR(a, NOT b): (0, 1/8) + a - b  (nufhe/gates.py:558-597)."""
    _binary_gate(thr, cloud_key, result, a, b, phase_to_t32(1, 8), 1, -1, perf_params)


def _mux_stepwise(thr, cloud_key, result, a, b, c, perf_params):
    """AND(a, b) and AND(NOT a, c) bootstrapped WITHOUT keyswitch, (0, 1/8) + u1 + u2 under the extracted key, one
    keyswitch (nufhe/gates.py:633-664)."""
    params = cloud_key.params
    bk, ks = cloud_key.bootstrap_key, cloud_key.keyswitch_key
    temp = LweSampleArray.empty(thr, params.in_out_params, result.shape)
    lwe_noiseless_trivial_constant(thr, temp, phase_to_t32(-1, 8))
    lwe_add_to(thr, temp, a)
    lwe_add_to(thr, temp, b)
    u1 = LweSampleArray.empty(thr, bk.extract_params, result.shape)
    bootstrap(thr, u1, bk, ks, MU, temp, perf_params, no_keyswitch=True)
    lwe_noiseless_trivial_constant(thr, temp, phase_to_t32(-1, 8))
    lwe_sub_to(thr, temp, a)
    lwe_add_to(thr, temp, c)
    u2 = LweSampleArray.empty(thr, bk.extract_params, result.shape)
    bootstrap(thr, u2, bk, ks, MU, temp, perf_params, no_keyswitch=True)
    total = LweSampleArray.empty(thr, bk.extract_params, result.shape)
    lwe_noiseless_trivial_constant(thr, total, MU)
    lwe_add_to(thr, total, u1)
    lwe_add_to(thr, total, u2)
    lwe_keyswitch(thr, result, ks, total)


def gate_mux(thr, cloud_key, result, a, b, c, perf_params=None):
    """
    Homomorphic bootstrapped MUX (``b if a else c``): two bootstraps without keyswitch,
    (0, 1/8) + u1 + u2, one keyswitch  (nufhe/gates.py:600-664).
    """
    check_shape(result, a, b, c)
    _check_sizes(cloud_key, result, a, b, c)
    thr.check_stream()
    if not single_kernel(perf_params):
        return _mux_stepwise(thr, cloud_key, result, a, b, c, perf_params)
    res = _Flat(result, result.shape, output=True)
    fa = _Flat(a, result.shape)
    fb = _Flat(b, result.shape)
    fc = _Flat(c, result.shape)
    _lib.call("nufhe_gate_mux", thr.handle, cloud_key._native.handle, res.desc, fa.desc, fb.desc, fc.desc,
              res.nbits)
    res.writeback()


# (c, pa, pb) of every bootstrapped binary gate: result = KS(BS((0, c) + pa a + pb b)) (the functions above)
BINARY_GATES = {
    'gate_nand': (phase_to_t32(1, 8), -1, -1),
    'gate_or': (phase_to_t32(1, 8), 1, 1),
    'gate_and': (phase_to_t32(-1, 8), 1, 1),
    'gate_xor': (phase_to_t32(1, 4), 2, 2),
    'gate_xnor': (phase_to_t32(-1, 4), -2, -2),
    'gate_nor': (phase_to_t32(-1, 8), -1, -1),
    'gate_andny': (phase_to_t32(-1, 8), -1, 1),
    'gate_andyn': (phase_to_t32(-1, 8), 1, -1),
    'gate_orny': (phase_to_t32(1, 8), -1, 1),
    'gate_oryn': (phase_to_t32(1, 8), 1, -1),
}


def gate_batch(thr, cloud_key, jobs, perf_params=None):
    """
    A list of INDEPENDENT bootstrapped gates -- different kinds, shapes and buffers -- as one bootstrap launch
    (``nufhe_gate_batch``; SURVEY 8f row 4, no reference counterpart beyond the gate-by-gate circuit of
    nufhe/operators_integer.py:64-95).  ``jobs`` = [(name, result, a, b) | ('gate_mux', result, a, b, c), ...] with the
    gate functions' own names and argument order.  A gate of up to one bit per CU takes as long as one bit (500
    dependent blind-rotation steps), so N small independent gates batched this way take the time of ONE instead of N.
    Results are word-for-word those of the individual gate calls.  All operands are read before any result is written:
    a job may update one of its operands in place (``carry = MUX(same, carry, a)``); two jobs must not share a result.
    With ``single_kernel_bootstrap == False`` the gates simply run one after the other (the reference's mode).
    """
    import ctypes
    jobs = list(jobs)
    for job in jobs:
        name = job[0]
        if name != 'gate_mux' and name not in BINARY_GATES:
            raise ValueError("gate_batch: %r is not a bootstrapped gate (gate_not / gate_copy / gate_constant need no batch)" % (name,))
        if len(job) != (5 if name == 'gate_mux' else 4):
            raise ValueError("gate_batch: %s takes %d ciphertext arguments" % (name, 4 if name == 'gate_mux' else 3))
    if not single_kernel(perf_params):
        # same contract as the fused path (include/nufhe_hip.h, nufhe_gate_batch): every operand is read before any
        # result is written.  The gates run one after the other here, so each one writes a temporary and the results
        # are stored at the end -- a job that names another job's result as an operand sees the old value in both modes.
        temps = []
        for job in jobs:
            result = job[1]
            tmp = LweSampleArray.empty(thr, result.params, result.shape)
            globals()[job[0]](thr, cloud_key, tmp, *job[2:], perf_params=perf_params)
            temps.append((result, tmp))
        for result, tmp in temps:
            lwe_copy(thr, result, tmp)
        return
    thr.check_stream()
    descs = (_lib.NufheGateJob * max(1, len(jobs)))()
    flats = []
    count = 0
    for job in jobs:
        name, result, args = job[0], job[1], job[2:]
        check_shape(result, *args)
        _check_sizes(cloud_key, result, *args)
        if int(numpy.prod(result.shape)) == 0:
            continue                                  # an empty gate: nothing to run (the C entry point allows it, too)
        i = count
        count += 1
        res = _Flat(result, result.shape, output=True)
        ops = [_Flat(x, result.shape) for x in args]
        flats.append((res, ops))                      # the temporaries of strided views must outlive the call
        d = descs[i]
        d.nbits = res.nbits
        d.result, d.a, d.b = res.desc, ops[0].desc, ops[1].desc
        if name == 'gate_mux':
            d.kind = _lib.JOB_MUX
            d.c = ops[2].desc
        else:
            d.kind = _lib.JOB_BINARY
            d.c0, d.pa, d.pb = [int(v) for v in BINARY_GATES[name]]
    _lib.call("nufhe_gate_batch", thr.handle, cloud_key._native.handle, descs, count, int(MU))
    for res, _ in flats:
        res.writeback()
