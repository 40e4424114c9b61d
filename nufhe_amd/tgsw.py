"""
TGSW parameter / key records (reference: nufhe/tgsw.py:43-78), the gadget-message step of
bootstrap-key generation (tgsw_add_message, tgsw.py:142-161 / tgsw_cpu.py:109-126) and the external product of
the step-by-step bootstrap driver (tgsw.py:165-172).
"""

import numpy

from . import _lib
from .device import ptr, int32_operand
from .numeric_functions import Torus32
from .tlwe import TLweParams, TLweKey, TLweSampleArray, tlwe_encrypt_zero


class TGswParams:
    """nufhe/tgsw.py:43-67"""

    def __init__(self, tlwe_params: TLweParams, decomp_length: int, bs_log2_base: int):
        decomp_range = numpy.arange(1, decomp_length + 1)
        self.base_powers = (2**(32 - decomp_range * bs_log2_base)).astype(Torus32)
        self.offset = (
            self.base_powers.astype(numpy.int64).sum() * (2**bs_log2_base // 2)).astype(Torus32)
        self.decomp_length = decomp_length
        self.bs_log2_base = bs_log2_base
        self.tlwe_params = tlwe_params

    def __eq__(self, other):
        return (
            self.__class__ == other.__class__
            and self.decomp_length == other.decomp_length
            and self.bs_log2_base == other.bs_log2_base
            and self.tlwe_params == other.tlwe_params)

    def __hash__(self):
        return hash((self.__class__, self.decomp_length, self.bs_log2_base, self.tlwe_params))


# pickled under the reference's module path (nufhe_amd/serialization.py)
TGswParams.__module__ = 'nufhe.tgsw'


class TGswKey:
    """nufhe/tgsw.py:70-78"""

    def __init__(self, params: TGswParams, tlwe_key: TLweKey):
        self.params = params
        self.tlwe_key = tlwe_key

    @classmethod
    def from_rng(cls, thr, params: TGswParams, rng):
        return cls(params, TLweKey.from_rng(thr, params.tlwe_params, rng))


def tgsw_encrypt_int(thr, rng, messages, noise: float, key: TGswKey):
    """
    TGSW encryptions of the integer ``messages`` (one per entry): int32 device array
    ``messages.shape + (k+1, l, k+1, N)`` (nufhe/tgsw.py:155-161).
    result = TGSW(0) + message * H, H = gadget: adds message * 2^(32 - Bgbit (d+1)) to
    coefficient 0 of polynomial m of row (m, d)  (tgsw_cpu.py:121-124).
    """
    params = key.params
    k1 = params.tlwe_params.mask_size + 1
    l = params.decomp_length
    shape = tuple(messages.shape)
    if l != 2 or params.bs_log2_base != 10:
        raise ValueError("the gfx950 kernels support bs_decomp_length=2, bs_log2_base=10")
    result = tlwe_encrypt_zero(thr, rng, shape + (k1, l), noise, key.tlwe_key)
    count = int(numpy.prod(shape)) if len(shape) else 1
    messages = int32_operand("TGSW messages", messages, thr.device)        # the kernel reads int32
    _lib.call("nufhe_tgsw_add_message", thr.handle, ptr(result), ptr(messages), count, k1 - 1)
    return result


class TransformedTGswSampleArray:
    """
    The bootstrapping key as the external product sees it: ``n`` TGSW samples in the transformed domain
    (nufhe/tgsw.py:99-124).  The reference keeps a uint64 / complex128 array ``[n, k+1, l, k+1, N or N/2]``;
    here the samples live on the device in the library's wave layout behind a native key handle
     and this object is the view of them that `tgsw_transformed_external_mul` takes.
    """

    def __init__(self, params: TGswParams, native, length: int):
        self.params = params
        self.shape = (int(length),)
        self._native = native


def tgsw_transformed_external_mul(thr, result: TLweSampleArray, bootstrap_key: TransformedTGswSampleArray,
                                  bk_row_idx: int, perf_params=None):
    """
    result = bootstrap_key[bk_row_idx] (x) result, in place: gadget decomposition of the k + 1 polynomials, forward
    transforms, multiply-accumulate against the row, inverse transforms (nufhe/tgsw.py:165-172,
    TGswTransformedExternalMul tgsw_gpu.py:110-169, reference tgsw_cpu.py:82-106).  One launch of the same
    external-product body the fused blind rotation iterates.
    """
    if len(bootstrap_key.shape) != 1:
        raise ValueError("the bootstrapping key is a one-dimensional array of TGSW samples")
    if not (0 <= bk_row_idx < bootstrap_key.shape[0]):
        raise ValueError("row %d out of range [0, %d)" % (bk_row_idx, bootstrap_key.shape[0]))
    tlwe_params = bootstrap_key.params.tlwe_params
    if result.params != tlwe_params:
        raise ValueError("the accumulator and the key have different TLWE parameters")
    thr.check_stream()
    acc = result.a.coeffs
    work = acc if acc.is_contiguous() else acc.contiguous()
    batch = int(numpy.prod(result.shape)) if result.shape else 1
    _lib.call("nufhe_external_mul", thr.handle, bootstrap_key._native.handle, ptr(work), int(bk_row_idx), batch)
    if work is not acc:
        acc.copy_(work)
