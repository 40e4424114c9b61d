"""
LWE ciphertext containers and operations (reference: nufhe/lwe.py:53-447).

Same classes, function names and argument meaning as the reference; ``thr`` is a
:class:`nufhe_amd.device.DeviceThread`.  Arrays are torch tensors in GPU memory; every kernel is a
C-ABI call into libnufhe_hip.so (LweLinear, LweNoiselessTrivial*, LweKeyswitch, and the
encrypt/decrypt dot products via nufhe_lwe_phase); only the random numbers are drawn on the host,
in the reference's order, so that one seed gives the same ciphertexts as the CPU oracle.
"""

import io
import logging
import pickle

import numpy
import torch

from . import _lib
from .device import DeviceThread, lwe_desc, ptr, int32_operand
from .numeric_functions import Torus32, ErrorFloat
from .random_numbers import rand_uniform_bool, rand_gaussian_torus32_host
from .utils import arrays_equal, to_numpy


class LweParams:
    """nufhe/lwe.py:53-68"""

    def __init__(self, size: int, min_noise: float, max_noise: float):
        self.size = size
        self.min_noise = min_noise
        self.max_noise = max_noise

    def __eq__(self, other):
        return (
            self.__class__ == other.__class__
            and self.size == other.size
            and self.min_noise == other.min_noise
            and self.max_noise == other.max_noise)

    def __hash__(self):
        return hash((self.__class__, self.size, self.min_noise, self.max_noise))


# pickled under the reference's module path (nufhe_amd/serialization.py)
LweParams.__module__ = 'nufhe.lwe'


class LweKey:
    """nufhe/lwe.py:71-106"""

    def __init__(self, params: LweParams, key):
        self.params = params
        self.key = key

    @classmethod
    def from_rng(cls, thr: DeviceThread, params: LweParams, rng):
        return cls(params, rand_uniform_bool(thr, rng, (params.size,)))

    @classmethod
    def from_tlwe_key(cls, params: LweParams, tlwe_key):
        poly_degree = tlwe_key.params.polynomial_degree
        mask_size = tlwe_key.params.mask_size
        assert params.size == poly_degree * mask_size
        return cls(params, tlwe_key.key.coeffs.reshape(-1))

    def dump(self, file_obj):
        pickle.dump(self.params, file_obj)
        pickle.dump(to_numpy(self.key), file_obj)

    @classmethod
    def load(cls, file_obj, thr):
        params = pickle.load(file_obj)
        key = pickle.load(file_obj)
        return cls(params, thr.to_device(key))

    def __eq__(self, other):
        return (
            self.__class__ == other.__class__
            and self.params == other.params
            and arrays_equal(self.key, other.key))


class LweSampleArrayShapeInfo:
    """nufhe/lwe.py:109-132 (shape consistency check; no Reikna types here)."""

    def __init__(self, a, b, current_variances):
        if (not (len(a.shape) - 1 == len(b.shape) == len(current_variances.shape))
                or not (tuple(a.shape[:-1]) == tuple(b.shape) == tuple(current_variances.shape))):
            raise ValueError("Inconsistent shapes: {a}, {b}, {cv}".format(
                a=tuple(a.shape), b=tuple(b.shape), cv=tuple(current_variances.shape)))
        self.shape = tuple(b.shape)
        self.size = a.shape[-1]

    def __eq__(self, other):
        return self.__class__ == other.__class__ and self.shape == other.shape and self.size == other.size

    def __hash__(self):
        return hash((self.__class__, self.shape, self.size))


class LweSampleArray:
    """
    A ciphertext object (nufhe/lwe.py:135-251): ``a[shape + (n,)]`` int32, ``b[shape]`` int32,
    ``current_variances[shape]`` float32, all in GPU memory.

    .. py:attribute:: shape

        The shape of the encrypted plaintext message.
    """

    def __init__(self, params: LweParams, a, b, current_variances):
        self.params = params
        self.a = a
        self.b = b
        self.current_variances = current_variances
        self.shape_info = LweSampleArrayShapeInfo(a, b, current_variances)

    @classmethod
    def empty(cls, thr: DeviceThread, params: LweParams, shape):
        shape = tuple(shape)
        a = thr.array(shape + (params.size,), Torus32)
        b = thr.array(shape, Torus32)
        current_variances = thr.array(shape, ErrorFloat)
        return cls(params, a, b, current_variances)

    @property
    def shape(self):
        return self.shape_info.shape

    @staticmethod
    def _mask_index(index):
        # the mask array has one more (trailing) axis than the message: an Ellipsis in the index
        # must not swallow it
        if isinstance(index, tuple) and any(i is Ellipsis for i in index):
            return index + (slice(None),)
        if index is Ellipsis:
            return (Ellipsis, slice(None))
        return index

    def __getitem__(self, index):
        """A view over the ciphertext, indexed like a numpy array of shape ``shape``."""
        return LweSampleArray(
            self.params, self.a[self._mask_index(index)], self.b[index], self.current_variances[index])

    def __setitem__(self, index, value):
        if not isinstance(value, LweSampleArray):
            raise ValueError("Only assignment of ciphertexts is supported")
        self.a[self._mask_index(index)] = value.a
        self.b[index] = value.b
        self.current_variances[index] = value.current_variances

    def copy(self):
        return LweSampleArray(
            self.params, self.a.clone(), self.b.clone(), self.current_variances.clone())

    def roll(self, shift, axis=-1):
        """Cyclic in-place shift along ``axis`` (numpy.roll semantics), nufhe/lwe.py:185-205."""
        if shift == 0:
            return
        axis = axis % len(self.shape)
        self.a.copy_(torch.roll(self.a, shift, dims=axis))
        self.b.copy_(torch.roll(self.b, shift, dims=axis))
        self.current_variances.copy_(torch.roll(self.current_variances, shift, dims=axis))

    def dump(self, file_obj):
        pickle.dump(self.params, file_obj)
        pickle.dump(to_numpy(self.a), file_obj)
        pickle.dump(to_numpy(self.b), file_obj)
        pickle.dump(to_numpy(self.current_variances), file_obj)

    def dumps(self):
        file_obj = io.BytesIO()
        self.dump(file_obj)
        return file_obj.getvalue()

    @classmethod
    def load(cls, file_obj, thr):
        params = pickle.load(file_obj)
        a = thr.to_device(pickle.load(file_obj))
        b = thr.to_device(pickle.load(file_obj))
        current_variances = thr.to_device(pickle.load(file_obj))
        return cls(params, a, b, current_variances)

    @classmethod
    def loads(cls, s, thr):
        return cls.load(io.BytesIO(s), thr)

    def __eq__(self, other):
        return (
            self.__class__ == other.__class__
            and self.params == other.params
            and arrays_equal(self.a, other.a)
            and arrays_equal(self.b, other.b)
            and arrays_equal(self.current_variances, other.current_variances))


# ---- flattening of (possibly strided / broadcast) N-d views into the C descriptor -------------

_log = logging.getLogger("nufhe_amd")

#: number of times a view could not be described by (pointer, one bit stride) and went through a
#: contiguous temporary (torch copy in, and copy back for outputs); each occurrence is also logged at
#: DEBUG level on the "nufhe_amd" logger.  Views whose leading axes collapse to one stride -- slices of
#: the first axis, steps, broadcasts of one sample -- never copy.
flat_copies = 0


def check_lwe_size(what: str, sample, expected: int):
    """The kernels address ``a`` rows of exactly the key's LWE size; a different size would read or
    write out of bounds (the reference gets a type error from Reikna here)."""
    size = sample.a.shape[-1]
    if size != expected:
        raise ValueError("{what}: LWE size {size} does not match the key's {expected}".format(
            what=what, size=size, expected=expected))


class _Flat:
    """2D/1D views of a sample broadcast to ``shape``; ``writeback()`` copies results back when a
    contiguous temporary had to be used for an output view."""

    def __init__(self, sample: LweSampleArray, shape, output=False):
        n = sample.a.shape[-1]
        shape = tuple(shape)
        nbits = int(numpy.prod(shape)) if len(shape) else 1
        a = sample.a.expand(shape + (n,))
        b = sample.b.expand(shape)
        cv = sample.current_variances.expand(shape)
        self._targets = None
        try:
            a2 = a.view(nbits, n)
            b2 = b.view(nbits)
            cv2 = cv.view(nbits)
            ok = (a2.stride(1) == 1 or n == 1) and (nbits <= 1 or b2.stride(0) == cv2.stride(0))
            if output and nbits > 1 and (a2.stride(0) == 0 or b2.stride(0) == 0):
                ok = False
        except RuntimeError:
            ok = False
        if not ok:
            global flat_copies
            flat_copies += 1
            _log.debug("ciphertext view of shape %s (strides a=%s b=%s) is not expressible as one bit stride: "
                       "%s through a contiguous temporary", shape, tuple(a.stride()), tuple(b.stride()),
                       "result written back" if output else "operand copied")
            if output:
                self._targets = (a, b, cv)
            a2 = a.contiguous().view(nbits, n)
            b2 = b.contiguous().view(nbits)
            cv2 = cv.contiguous().view(nbits)
        self.a, self.b, self.cv = a2, b2, cv2
        self.nbits = nbits
        self.size = n
        self.desc = lwe_desc(a2, b2, cv2, n)

    def writeback(self):
        if self._targets is not None:
            a, b, cv = self._targets
            a.copy_(self.a.view(a.shape))
            b.copy_(self.b.view(b.shape))
            cv.copy_(self.cv.view(cv.shape))


class LweKeyswitchKey:
    """
    nufhe/lwe.py:254-308.  The key lives on the device inside the native cloud-key handle, in the
    library's own layout (digits 1..3 only -- the base-0 slice is all zeros).  ``lwe`` is the key in the
    reference layout (a [N*k, t, base, n], b / current_variances [N*k, t, base]) as host arrays: given
    at construction for keys that come from a stream, downloaded on first use (dump, comparison) for
    keys generated on the device.
    """

    def __init__(self, lwe=None, native=None, params=None, shape=None):
        if lwe is not None:
            params, shape = lwe.params, lwe.shape
        input_size, decomp_length, base = shape
        self._lwe = lwe
        self._params = params
        self._native = native
        self.input_size = input_size
        self.output_size = params.size
        self.decomp_length = decomp_length
        self.log2_base = int(numpy.log2(base))

    @property
    def lwe(self):
        if self._lwe is None:
            shape = (self.input_size, self.decomp_length, 2**self.log2_base)
            a = numpy.empty(shape + (self.output_size,), Torus32)
            b = numpy.empty(shape, Torus32)
            cv = numpy.empty(shape, ErrorFloat)
            _lib.call("nufhe_ks_download_reference", self._native.handle, a.ctypes.data, b.ctypes.data, cv.ctypes.data)
            self._lwe = HostLweSampleArray(self._params, a, b, cv)
        return self._lwe

    @classmethod
    def from_tgsw_key(cls, thr, rng, ks_decomp_length: int, ks_log2_base: int, lwe_key: LweKey, tgsw_key,
                      native=None):
        """MakeLweKeyswitchKey (lwe.py:265-295, lwe_cpu.py:27-59) on the device: the random numbers are
        drawn on the host in the reference's order (lwe.py:285-288: centred Gaussian noises first, then
        the uniform masks) and uploaded ONCE; messages, noise and the <mask, key> products are combined
        by nufhe_ks_make straight into the cloud key's storage."""
        if native is None:
            raise ValueError("a native cloud key (NativeCloudKey) is needed to hold the keyswitch key")
        if ks_decomp_length != 8 or ks_log2_base != 2:
            raise ValueError("the gfx950 keyswitch kernels support ks_decomp_length=8, ks_log2_base=2")
        accum_params = tgsw_key.params.tlwe_params
        extract_params = accum_params.extracted_lweparams
        # nufhe_ks_make reads int32 [mask_size * 1024] and int32 [lwe_size] and indexes the masks by the native key's
        # own sizes: anything else (another dtype, a key of another dimension) is refused here
        in_key = int32_operand("extracted TLWE key", LweKey.from_tlwe_key(extract_params, tgsw_key.tlwe_key).key,
                               thr.device, (native.mask_size * 1024,))
        out_key = int32_operand("LWE key", lwe_key.key, thr.device, (native.lwe_size,))
        input_size = in_key.shape[0]
        output_size = out_key.shape[0]
        noise = lwe_key.params.min_noise
        base = 2**ks_log2_base

        noises_b = rand_gaussian_torus32_host(
            rng, 0, noise, (input_size, ks_decomp_length, base - 1), centered=True)
        noises_a = rng.uniform_torus32((input_size, ks_decomp_length, base - 1, output_size))
        d_noises_a = thr.to_device(noises_a)
        d_noises_b = thr.to_device(noises_b)
        _lib.call("nufhe_ks_make", native.handle, ptr(d_noises_a), ptr(d_noises_b), ptr(in_key), ptr(out_key),
                  float(numpy.float32(noise**2)))
        thr.synchronize()            # the uploads may be released
        return cls(native=native, params=lwe_key.params, shape=(input_size, ks_decomp_length, base))

    def dump(self, file_obj):
        self.lwe.dump(file_obj)

    @classmethod
    def load(cls, file_obj, thr):
        params = pickle.load(file_obj)
        a = pickle.load(file_obj)
        b = pickle.load(file_obj)
        cv = pickle.load(file_obj)
        return cls(HostLweSampleArray(params, a, b, cv))

    def __eq__(self, other):
        return self.__class__ == other.__class__ and self.lwe == other.lwe


class HostLweSampleArray:
    """Host-memory twin of LweSampleArray, used for the 65 MB keyswitch key."""

    def __init__(self, params, a, b, current_variances):
        self.params = params
        self.a = numpy.ascontiguousarray(a, Torus32)
        self.b = numpy.ascontiguousarray(b, Torus32)
        self.current_variances = numpy.ascontiguousarray(current_variances, ErrorFloat)
        self.shape_info = LweSampleArrayShapeInfo(self.a, self.b, self.current_variances)

    @property
    def shape(self):
        return self.shape_info.shape

    def dump(self, file_obj):
        pickle.dump(self.params, file_obj)
        pickle.dump(self.a, file_obj)
        pickle.dump(self.b, file_obj)
        pickle.dump(self.current_variances, file_obj)

    def __eq__(self, other):
        return (
            self.__class__ == other.__class__ and self.params == other.params
            and arrays_equal(self.a, other.a) and arrays_equal(self.b, other.b)
            and arrays_equal(self.current_variances, other.current_variances))


def lwe_keyswitch(thr: DeviceThread, result: LweSampleArray, ks: LweKeyswitchKey, sample: LweSampleArray):
    """nufhe/lwe.py:311-322: translate the sample to the output key (LWE(N*k) -> LWE(n))."""
    if ks._native is None:
        raise ValueError("this keyswitch key is not attached to a cloud key on the device")
    check_lwe_size("lwe_keyswitch result", result, ks.output_size)
    check_lwe_size("lwe_keyswitch sample", sample, ks.input_size)
    res = _Flat(result, result.shape, output=True)
    src = _Flat(sample, result.shape)
    _lib.call("nufhe_keyswitch", thr.handle, ks._native.handle, res.desc, src.desc, res.nbits)
    res.writeback()


def lwe_encrypt(thr: DeviceThread, rng, result: LweSampleArray, messages, noise: float, key: LweKey):
    """nufhe/lwe.py:325-333: randomness on the host in the reference's order (Gaussian b-noise, then
    the uniform mask), b = mu + e + a.s on the GPU (LweEncrypt, lwe_cpu.py:96-104)."""
    messages = to_numpy(messages).astype(Torus32)
    lwe_size = key.params.size
    noises_b = rand_gaussian_torus32_host(rng, 0, noise, messages.shape)
    noises_a = rng.uniform_torus32(messages.shape + (lwe_size,))
    base = (noises_b.astype(numpy.uint32) + messages.astype(numpy.uint32)).view(Torus32)
    a_dev = thr.to_device(noises_a).reshape(-1, lwe_size)
    base_dev = thr.to_device(base).reshape(-1)
    b_dev = thr.array((base_dev.shape[0],), Torus32)
    key_dev = key.key.contiguous()
    _lib.call("nufhe_lwe_phase", thr.handle, ptr(b_dev), 1, ptr(a_dev), lwe_size, ptr(base_dev), 1,
              ptr(key_dev), 1, base_dev.shape[0], lwe_size)
    result.a.copy_(a_dev.reshape(result.a.shape))
    result.b.copy_(b_dev.reshape(result.b.shape))
    result.current_variances.fill_(float(numpy.float32(noise**2)))


def lwe_decrypt(thr: DeviceThread, sample: LweSampleArray, key: LweKey):
    """nufhe/lwe.py:336-343: phase b - a.s on the GPU (LweDecrypt, lwe_cpu.py:107-112), returned
    as a host array."""
    check_lwe_size("lwe_decrypt sample", sample, key.params.size)
    flat = _Flat(sample, sample.shape)
    out = thr.array((flat.nbits,), Torus32)
    key_dev = key.key.contiguous()
    _lib.call("nufhe_lwe_phase", thr.handle, ptr(out), 1, ptr(flat.a), flat.desc.a_stride, ptr(flat.b),
              flat.desc.b_stride, ptr(key_dev), -1, flat.nbits, flat.size)
    return to_numpy(out).reshape(sample.shape)


def lwe_noiseless_trivial(thr: DeviceThread, result: LweSampleArray, mus):
    """nufhe/lwe.py:346-351: (0, mu) for each mu (broadcast to the result shape)."""
    mus_dev = mus if hasattr(mus, 'device') and not isinstance(mus, numpy.ndarray) else thr.to_device(
        numpy.asarray(mus, Torus32))
    result.a.zero_()
    result.b.copy_(mus_dev.expand(result.b.shape))
    result.current_variances.zero_()


def lwe_noiseless_trivial_constant(thr: DeviceThread, result: LweSampleArray, mu):
    """nufhe/lwe.py:354-359"""
    res = _Flat(result, result.shape, output=True)
    _lib.call("nufhe_lwe_trivial_const", thr.handle, res.desc, int(numpy.int32(mu)), res.nbits, res.size)
    res.writeback()


def _linear(thr, result, source, p, add_result):
    thr.check_stream()
    res = _Flat(result, result.shape, output=True)
    src = _Flat(source, result.shape)
    if src.size != res.size:
        raise ValueError("LWE sizes differ: %d vs %d" % (res.size, src.size))
    _lib.call("nufhe_lwe_linear", thr.handle, res.desc, src.desc, int(p), int(add_result), res.nbits, res.size)
    res.writeback()


def lwe_negate(thr, result, source):
    """result = -sample (nufhe/lwe.py:365-373)"""
    _linear(thr, result, source, -1, False)


def lwe_copy(thr, result, source):
    """result = sample (nufhe/lwe.py:376-384)"""
    _linear(thr, result, source, 1, False)


def lwe_add_to(thr, result, source):
    """result += sample (nufhe/lwe.py:387-395)"""
    _linear(thr, result, source, 1, True)


def lwe_add_mul_to(thr, result, p: int, source):
    """result += p * sample (nufhe/lwe.py:398-402)"""
    _linear(thr, result, source, p, True)


def lwe_sub_to(thr, result, source):
    """result -= sample (nufhe/lwe.py:405-412)"""
    _linear(thr, result, source, -1, True)


def lwe_sub_mul_to(thr, result, p: int, source):
    """result -= p * sample (nufhe/lwe.py:415-422)"""
    _linear(thr, result, source, -p, True)


def concatenate(lwe_sample_arrays, axis=0, out=None):
    """Concatenates several ciphertext arrays along ``axis`` (nufhe/lwe.py:425-447)."""
    if len(lwe_sample_arrays) == 0:
        raise ValueError("Need at least one ciphertext to concatenate")
    params = lwe_sample_arrays[0].params
    axis = axis % len(lwe_sample_arrays[0].shape)
    a = torch.cat([lwe.a for lwe in lwe_sample_arrays], dim=axis)
    b = torch.cat([lwe.b for lwe in lwe_sample_arrays], dim=axis)
    cv = torch.cat([lwe.current_variances for lwe in lwe_sample_arrays], dim=axis)
    if out is None:
        return LweSampleArray(params, a, b, cv)
    out.a.copy_(a)
    out.b.copy_(b)
    out.current_variances.copy_(cv)
    return out
