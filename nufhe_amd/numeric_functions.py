"""
Numeric types and torus helpers (reference: nufhe/numeric_functions.py:30-40,
nufhe/numeric_functions_gpu.py:30-36).  Host-side only.
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
