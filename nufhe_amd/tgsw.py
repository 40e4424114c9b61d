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


class TGswSampleArray:
    """
    TGSW samples before the transform (nufhe/tgsw.py:81-96): ``samples`` is a TLweSampleArray of shape
    ``shape + (k + 1, l)`` -- row (m, d) of a sample is the TLWE encryption that carries the gadget entry
    2^(32 - Bgbit (d + 1)) on polynomial m.
    """

    def __init__(self, params: TGswParams, samples: TLweSampleArray):
        self.mask_size = params.tlwe_params.mask_size
        self.decomp_length = params.decomp_length
        self.samples = samples
        self.params = params
        self.shape = tuple(samples.shape[:-2])

    @classmethod
    def empty(cls, thr, params: TGswParams, shape):
        rows = (params.tlwe_params.mask_size + 1, params.decomp_length)
        return cls(params, TLweSampleArray.empty(thr, params.tlwe_params, tuple(shape) + rows))


def _check_gadget(params: TGswParams):
    if params.decomp_length != 2 or params.bs_log2_base != 10:
        raise ValueError("the gfx950 kernels support bs_decomp_length=2, bs_log2_base=10")


def tgsw_encrypt_zero(thr, rng, result: TGswSampleArray, noise: float, key: TGswKey, perf_params=None):
    """result = TGSW(0): every row a fresh TLWE encryption of zero (nufhe/tgsw.py:148-151)."""
    tlwe_encrypt_zero(thr, rng, result.samples, noise, key.tlwe_key, perf_params)


def tgsw_add_message(thr, result: TGswSampleArray, messages):
    """result += messages * H, H the gadget: adds message * 2^(32 - Bgbit (d + 1)) to coefficient 0 of polynomial m
    of row (m, d) (nufhe/tgsw.py:142-145, TGswAddMessage tgsw_gpu.mako:18-39, reference tgsw_cpu.py:109-126);
    ``messages``: one integer per sample."""
    _check_gadget(result.params)
    shape = result.shape
    count = int(numpy.prod(shape)) if shape else 1
    messages = int32_operand("TGSW messages", messages, thr.device, shape=shape)        # the kernel reads int32
    thr.check_stream()
    coeffs = result.samples.a.coeffs
    work = coeffs if coeffs.is_contiguous() else coeffs.contiguous()
    _lib.call("nufhe_tgsw_add_message", thr.handle, ptr(work), ptr(messages), count, result.mask_size)
    if work is not coeffs:
        coeffs.copy_(work)


def tgsw_encrypt_int(thr, rng, result: TGswSampleArray, messages, noise: float, key: TGswKey, perf_params=None):
    """result = TGSW encryptions of the integer ``messages`` (nufhe/tgsw.py:155-161)."""
    _check_gadget(key.params)
    tgsw_encrypt_zero(thr, rng, result, noise, key, perf_params)
    tgsw_add_message(thr, result, messages)


class TransformedTGswSampleArray:
    """
    The bootstrapping key as the external product sees it: ``n`` TGSW samples in the transformed domain
    (nufhe/tgsw.py:99-124).  The reference keeps a uint64 / complex128 array ``[n, k+1, l, k+1, N or N/2]``;
    here the samples live on the device in the library's wave layout behind a native key handle
     and this object is the view of them that `tgsw_transformed_external_mul` takes.
    """

    def __init__(self, params: TGswParams, native, length: int):
        self.mask_size = params.tlwe_params.mask_size
        self.decomp_length = params.decomp_length
        self.params = params
        self.shape = (int(length),)
        self._native = native

    @classmethod
    def empty(cls, thr, params: TGswParams, shape):
        """Device storage for ``shape = (n,)`` transformed samples (a key handle of its own, without a keyswitch
        key; nufhe/tgsw.py:108-114)."""
        from .bootstrap import NativeCloudKey
        shape = tuple(shape)
        if len(shape) != 1:
            raise ValueError("the transformed samples of a bootstrapping key form a one-dimensional array")
        native = NativeCloudKey(thr, shape[0], params.tlwe_params.transform_type, params.tlwe_params.mask_size)
        return cls(params, native, shape[0])


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


def tgsw_transform_samples(thr, result: TransformedTGswSampleArray, source: TGswSampleArray, perf_params=None):
    """result = forward transform of every polynomial of ``source`` in the form the external product multiplies with
    (nufhe/tgsw.py:134-138 -> tlwe_transform_samples tlwe.py:199-207, TLweTransformSamples tlwe_gpu.py:199-236: transform
    + Montgomery preparation in the reference; here the library's wave layout)."""
    if source.params != result.params or source.shape != result.shape:
        raise ValueError("source of shape %s does not fill transformed samples of shape %s" % (source.shape, result.shape))
    thr.check_stream()
    _lib.call("nufhe_bk_from_coeffs", result._native.handle, ptr(source.samples.a.coeffs.contiguous()))
