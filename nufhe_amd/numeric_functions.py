"""
Numeric types and torus helpers (reference: nufhe/numeric_functions.py:30-40,
nufhe/numeric_functions_gpu.py:30-77).
"""

import numpy

Torus32 = numpy.int32
Int32 = numpy.int32
ErrorFloat = numpy.float32


def _wrap_i32(value):
    value = int(value) & 0xffffffff
    return numpy.int32(value - (1 << 32) if value >= (1 << 31) else value)


def phase_to_t32(phase: int, mspace_size: int):
    """Nearest torus element of ``phase / mspace_size`` (numeric_functions.py:30-31), with the
    two's-complement wrap NumPy 1.x applied silently (e.g. phase_to_t32(-1, 8) == -2**29)."""
    return _wrap_i32((phase % mspace_size) * (2**32 // mspace_size))


def double_to_t32(d):
    """numeric_functions.py:39-40"""
    d = numpy.asarray(d, numpy.float64)
    return ((d - numpy.trunc(d)) * 2**32).astype(Torus32)


def t32_to_phase(thr, result, messages, mspace_size: int):
    """Modulus switch of torus elements to ``mspace_size`` phases, round to nearest: with
    interval = 2^32 // mspace_size, result = (uint32(messages) + interval // 2) // interval in wrapping 32-bit
    unsigned arithmetic (nufhe/numeric_functions.py:34-36, Torus32ToPhase numeric_functions_gpu.py:39-77,
    numeric_functions_cpu.py:23-37).  ``result`` and ``messages``: int32 device arrays of the same shape."""
    from . import _lib
    from .device import ptr, int32_operand
    if tuple(result.shape) != tuple(messages.shape):
        raise ValueError("result of shape %s, messages of shape %s" % (tuple(result.shape), tuple(messages.shape)))
    if not (0 < mspace_size < 2**32):
        raise ValueError("mspace_size must be in [1, 2^32), got %r" % (mspace_size,))
    thr.check_stream()
    src = int32_operand("messages", messages, thr.device)
    out = result if result.is_contiguous() else result.new_empty(result.shape)
    _lib.call("nufhe_t32_to_phase", thr.handle, ptr(out), ptr(src), src.numel(), int(mspace_size))
    if out is not result:
        result.copy_(out)
