"""
Bootstrap key and the bootstrap driver (reference: nufhe/bootstrap.py:44-229,
nufhe/blind_rotate.py:262-281).

The reference's driver issues ~10 kernel launches per gate (mod-switch x2, fill, shift, trivial,
fused blind-rotate, keyswitch, ...); here `bootstrap()` is ONE C-ABI call that enqueues the fused
gfx950 bootstrap kernel (+ the keyswitch kernels).

The reference's second mode, ``PerformanceParameters(single_kernel_bootstrap=False)`` (bootstrap.py:96-196: one
shift, one external product and one addition per key row, 1,500 launches per blind rotation), is kept as the
step-by-step driver below -- `mux_rotate`, `blind_rotate`, `blind_rotate_and_extract` with the reference's names and
argument order, every step one launch of a gfx950 kernel.  It is the structure the CPU oracle mirrors
(SURVEY 3.2), serves callers of the low-level API, and gives the fused kernels a second, independently composed
device path to be compared with (tests/test_gpu_stepwise.py): same bits, ~100x the time.
"""

import ctypes
import os
import pickle

import numpy

from . import _lib
from . import serialization
from .device import ptr
from .lwe import LweParams, LweKey, LweSampleArray, LweKeyswitchKey, _Flat, check_lwe_size, lwe_keyswitch
from .numeric_functions import Torus32, t32_to_phase
from .polynomials import TorusPolynomialArray, shift_tp_inverted_power
from .tgsw import (TGswKey, TGswParams, TGswSampleArray, TransformedTGswSampleArray, tgsw_encrypt_int,
                   tgsw_transform_samples, tgsw_transformed_external_mul)
from .tlwe import (TLweSampleArray, tlwe_noiseless_trivial, tlwe_extract_lwe_samples, tlwe_shift_polynomials,
                   tlwe_add_to, tlwe_copy)
from .utils import arrays_equal


class NativeCloudKey:
    """Owner of the device copies of the bootstrapping key (wave layout) and the keyswitch key."""

    TRANSFORMS = {'NTT': 0, 'FFT': 1}

    def __init__(self, thr, lwe_size, transform_type='NTT', mask_size=1):
        self.thr = thr
        self.transform_type = transform_type
        self.mask_size = mask_size
        self.lwe_size = int(lwe_size)
        handle = ctypes.c_void_p()
        _lib.call("nufhe_cloudkey_create", thr.handle, int(lwe_size), self.TRANSFORMS[transform_type],
                  int(mask_size), ctypes.byref(handle))
        self.handle = handle
        import weakref
        ref = weakref.ref(thr)
        _lib.register_stream_guard(handle, lambda: (ref() is not None) and ref().check_stream())
        thr._cloud_keys.add(self)
        default_engine = os.environ.get("NUFHE_NTT_ENGINE")
        if default_engine and transform_type == 'NTT':
            self.set_engine(default_engine)

    def set_engine(self, engine):
        """Arithmetic behind the gates of an NTT key: 'native' (u64 prime-field NTT kernels) or 'exact-fft' (fp64 folded FFT
        on a 16-bit split key: the same words for every input, ~0.4 x the time; include/nufhe_hip.h,
        nufhe_cloudkey_set_engine).  The NUFHE_NTT_ENGINE environment variable sets the default of new keys."""
        if engine not in _lib.ENGINES:
            raise ValueError("unknown engine %r (one of %s)" % (engine, sorted(set(_lib.ENGINES))))
        _lib.call("nufhe_cloudkey_set_engine", self.handle, _lib.ENGINES[engine])

    def get_engine(self):
        e = ctypes.c_int(0)
        _lib.call("nufhe_cloudkey_get_engine", self.handle, ctypes.byref(e))
        return {_lib.ENGINE_NATIVE: 'native', _lib.ENGINE_EXACT_FFT: 'exact-fft'}[e.value]

    def image_bytes(self):
        size = ctypes.c_size_t(0)
        _lib.call("nufhe_cloudkey_image_bytes", self.handle, ctypes.byref(size))
        return int(size.value)

    def export_image(self, out=None):
        """The whole key (bootstrapping key in the kernels' layout + keyswitch key) as ONE uint8 device tensor, the
        unit a device collective moves (multi_gpu.broadcast_cloud_key)."""
        import torch
        if out is None:
            out = torch.empty(self.image_bytes(), dtype=torch.uint8, device=self.thr.device)
        _lib.call("nufhe_cloudkey_export_image", self.handle, ptr(out))
        return out

    def import_image(self, image):
        if image.numel() * image.element_size() != self.image_bytes() or not image.is_contiguous():
            raise ValueError("key image of %d bytes, this key needs %d contiguous bytes" % (
                image.numel() * image.element_size(), self.image_bytes()))
        _lib.call("nufhe_cloudkey_import_image", self.handle, ptr(image))

    def destroy(self):
        if self.handle:
            _lib.unregister_stream_guard(self.handle)
        if self.handle and not self.thr._released:
            _lib.lib().nufhe_cloudkey_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class BootstrapKey:
    """
    nufhe/bootstrap.py:44-92.  The transformed TGSW samples live on the device in the library's
    own layout; ``dump``/``load`` convert from/to the reference's storage format (natural-order
    NTT, Montgomery-prepared uint64 [n, k+1, l, k+1, N]).
    """

    def __init__(self, in_out_params: LweParams, bk_params: TGswParams, native: NativeCloudKey):
        accum_params = bk_params.tlwe_params
        self.in_out_params = in_out_params
        self.bk_params = bk_params
        self.accum_params = accum_params
        self.extract_params = accum_params.extracted_lweparams
        self._native = native
        # the transformed TGSW samples as the step-by-step driver addresses them (bootstrap.py:57, `bk.tgsw`)
        self.tgsw = TransformedTGswSampleArray(bk_params, native, in_out_params.size)

    @classmethod
    def from_rng(cls, thr, rng, lwe_key: LweKey, tgsw_key: TGswKey, native: NativeCloudKey,
                 perf_params=None):
        in_out_params = lwe_key.params
        bk_params = tgsw_key.params
        accum_params = bk_params.tlwe_params
        # non-transformed key: TGSW encryptions of the LWE key bits (bootstrap.py:67-69)
        bk = TGswSampleArray.empty(thr, bk_params, (in_out_params.size,))
        tgsw_encrypt_int(thr, rng, bk, lwe_key.key, accum_params.min_noise, tgsw_key, perf_params)
        # to the transformed domain, where it is used (bootstrap.py:72-74)
        key = cls(in_out_params, bk_params, native)
        tgsw_transform_samples(thr, key.tgsw, bk, perf_params)
        thr.synchronize()
        return key

    def transformed_reference_format(self):
        """Host array in the reference's storage format: uint64 [n, k+1, l, k+1, N] (NTT) or
        complex128 [n, k+1, l, k+1, N/2] (FFT)."""
        k1 = self.bk_params.tlwe_params.mask_size + 1
        N = self.bk_params.tlwe_params.polynomial_degree
        fft = self.bk_params.tlwe_params.transform_type == 'FFT'
        shape = (self.in_out_params.size, k1, self.bk_params.decomp_length, k1, N // 2 if fft else N)
        out = numpy.empty(shape, numpy.complex128 if fft else numpy.uint64)
        _lib.call("nufhe_bk_download_reference", self._native.handle, out.ctypes.data_as(ctypes.c_void_p))
        return out

    def dump(self, file_obj):
        """The reference's record sequence (bootstrap.py:78-80 and the nested dumps)."""
        serialization.write_bootstrap_key(
            file_obj, self.in_out_params, self.bk_params, self.transformed_reference_format(),
            serialization.bootstrap_key_variances(self.in_out_params, self.bk_params))

    @classmethod
    def load(cls, file_obj, thr, native: NativeCloudKey):
        in_out_params, bk_params, coeffs, _ = serialization.read_bootstrap_key(file_obj)
        fft = bk_params.tlwe_params.transform_type == 'FFT'
        arr = numpy.ascontiguousarray(coeffs, numpy.complex128 if fft else numpy.uint64)
        k1 = bk_params.tlwe_params.mask_size + 1
        expected = (in_out_params.size, k1, bk_params.decomp_length, k1,
                    bk_params.tlwe_params.polynomial_degree // (2 if fft else 1))
        if tuple(arr.shape) != expected:
            raise ValueError("bootstrap key array has shape %s, expected %s" % (arr.shape, expected))
        if in_out_params.size != native.lwe_size or bk_params.tlwe_params.mask_size != native.mask_size:
            raise ValueError("bootstrap key stream (n = %d, k = %d) does not match the cloud key parameters "
                             "(n = %d, k = %d)" % (in_out_params.size, bk_params.tlwe_params.mask_size,
                                                   native.lwe_size, native.mask_size))
        _lib.call("nufhe_bk_upload_reference", native.handle, arr.ctypes.data_as(ctypes.c_void_p))
        return cls(in_out_params, bk_params, native)

    def __eq__(self, other):
        return (
            self.__class__ == other.__class__
            and self.in_out_params == other.in_out_params
            and self.bk_params == other.bk_params
            and arrays_equal(self.transformed_reference_format(), other.transformed_reference_format()))


def single_kernel(perf_params):
    """False only for a performance-parameter object that asks for the step-by-step driver."""
    return perf_params is None or getattr(perf_params, 'single_kernel_bootstrap', True) is not False


def mux_rotate(thr, result: TLweSampleArray, accum: TLweSampleArray, bki: TransformedTGswSampleArray, bk_idx: int,
               barai, bk_params: TGswParams = None, perf_params=None):
    """result = accum + BK[bk_idx] (x) ((X^barai[..., bk_idx] - 1) accum): one step of the blind rotation as three
    launches (nufhe/bootstrap.py:95-108)."""
    tlwe_shift_polynomials(thr, result, accum, barai, bk_idx)
    tgsw_transformed_external_mul(thr, result, bki, bk_idx, perf_params)
    tlwe_add_to(thr, result, accum)


def blind_rotate(thr, accum: TLweSampleArray, bk: BootstrapKey, bara, n: int, bk_params: TGswParams = None,
                 perf_params=None):
    """accum *= X^(sum_i bara[..., i] s_i) over the first ``n`` key rows, ping-ponging between the accumulator and
    one temporary (nufhe/bootstrap.py:119-142)."""
    if not (0 <= n <= bk.tgsw.shape[0]):
        raise ValueError("%d rows requested from a key of %d" % (n, bk.tgsw.shape[0]))
    spare = TLweSampleArray.empty(thr, accum.params, accum.shape)
    src, dst = accum, spare
    for i in range(n):
        mux_rotate(thr, dst, src, bk.tgsw, i, bara, bk.bk_params, perf_params)
        src, dst = dst, src
    if src is not accum:
        tlwe_copy(thr, accum, src)


def blind_rotate_and_extract(thr, result: LweSampleArray, v: TorusPolynomialArray, bk: BootstrapKey,
                             ks: LweKeyswitchKey, barb, bara, perf_params=None, no_keyswitch=False):
    """result = LWE(v_p), p = barb - sum_i bara_i s_i mod 2N: test vector rotated by -barb, trivial accumulator,
    blind rotation, extraction of the constant coefficient and (unless ``no_keyswitch``) the keyswitch back to the
    input key (nufhe/bootstrap.py:154-196), one launch per step."""
    accum_params = bk.accum_params
    shape = tuple(result.shape)
    extracted = result if no_keyswitch else LweSampleArray.empty(thr, bk.extract_params, shape)
    check_lwe_size("extracted sample", extracted, bk.extract_params.size)
    rotated = TorusPolynomialArray.empty(thr, accum_params.polynomial_degree, shape)
    shift_tp_inverted_power(thr, rotated, barb, v)
    acc = TLweSampleArray.empty(thr, accum_params, shape)
    tlwe_noiseless_trivial(thr, acc, rotated)
    blind_rotate(thr, acc, bk, bara, bk.in_out_params.size, bk.bk_params, perf_params)
    tlwe_extract_lwe_samples(thr, extracted, acc)
    if not no_keyswitch:
        lwe_keyswitch(thr, result, ks, extracted)


def _bootstrap_stepwise(thr, result, bk, ks, mu, x, perf_params, no_keyswitch):
    """nufhe/bootstrap.py:206-229 with ``single_kernel_bootstrap=False``"""
    N = bk.accum_params.polynomial_degree
    shape = tuple(result.shape)
    xa, xb = x.a, x.b
    if tuple(x.shape) != shape:
        # the fused call broadcasts its argument to the result's shape (_Flat); so does this driver
        try:
            xb = xb.expand(shape).contiguous()
            xa = xa.expand(shape + (xa.shape[-1],)).contiguous()
        except RuntimeError:
            raise ValueError("argument of shape %s cannot be broadcast to the result's shape %s" % (
                tuple(x.shape), shape))
    barb = thr.array(shape, Torus32)
    bara = thr.array(shape + (bk.in_out_params.size,), Torus32)
    t32_to_phase(thr, barb, xb, 2 * N)
    t32_to_phase(thr, bara, xa, 2 * N)
    testvect = TorusPolynomialArray.empty(thr, N, shape)
    testvect.coeffs.fill_(int(numpy.int32(mu)))
    blind_rotate_and_extract(thr, result, testvect, bk, ks, barb, bara, perf_params, no_keyswitch=no_keyswitch)


def bootstrap(thr, result: LweSampleArray, bk: BootstrapKey, ks: LweKeyswitchKey, mu, x: LweSampleArray,
              perf_params=None, no_keyswitch=False):
    """
    result = LWE(mu) iff phase(x) > 0, LWE(-mu) iff phase(x) < 0  (nufhe/bootstrap.py:206-229).
    With ``no_keyswitch`` the result is an LWE sample under the extracted key (size N*k).
    ``perf_params.single_kernel_bootstrap == False`` takes the step-by-step driver above instead of the fused call.
    """
    thr.check_stream()
    check_lwe_size("bootstrap result", result, bk.extract_params.size if no_keyswitch else bk.in_out_params.size)
    check_lwe_size("bootstrap argument", x, bk.in_out_params.size)
    if not single_kernel(perf_params):
        return _bootstrap_stepwise(thr, result, bk, ks, mu, x, perf_params, no_keyswitch)
    res = _Flat(result, result.shape, output=True)
    src = _Flat(x, result.shape)
    _lib.call("nufhe_bootstrap", thr.handle, bk._native.handle, res.desc, src.desc,
              int(numpy.int32(mu)), res.nbits, int(bool(no_keyswitch)))
    res.writeback()
