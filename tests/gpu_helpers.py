"""Helpers for the GPU parity tests: device buffers from numpy arrays and raw C-ABI calls."""

import ctypes

import numpy
import torch

import nufhe_amd
from nufhe_amd import _lib
from nufhe_amd.api_low_level import NuFHECloudKey, NuFHESecretKey, NuFHEParameters
from nufhe_amd.bootstrap import BootstrapKey, NativeCloudKey
from nufhe_amd.device import DeviceThread, ptr
from nufhe_amd.lwe import LweKey, LweKeyswitchKey, LweSampleArray, HostLweSampleArray


def dev(thr, arr):
    return thr.to_device(numpy.ascontiguousarray(arr))


def host(t):
    a = t.detach().cpu().numpy()
    return a


def host_u64(t):
    a = t.detach().cpu().numpy()
    return a.view(numpy.uint64) if a.dtype == numpy.int64 else a


def cloud_key_from_arrays(thr, ck, params=None):
    """Builds a device cloud key from the oracle's host arrays (reference formats)."""
    params = params or NuFHEParameters()
    native = NativeCloudKey(thr, params.in_out_params.size, params._transform_type, params._tlwe_mask_size)
    bk = numpy.ascontiguousarray(ck.bk, numpy.complex128 if params._transform_type == 'FFT' else numpy.uint64)
    _lib.call("nufhe_bk_upload_reference", native.handle, bk.ctypes.data_as(ctypes.c_void_p))
    bkey = BootstrapKey(params.in_out_params, params.tgsw_params, native)
    ks = LweKeyswitchKey(HostLweSampleArray(params.in_out_params, ck.ks_a, ck.ks_b, ck.ks_cv))
    NuFHECloudKey._attach_keyswitch(native, ks)
    return NuFHECloudKey(params, bkey, ks, native)


def secret_key_from_array(thr, lwe_key, params=None):
    params = params or NuFHEParameters()
    return NuFHESecretKey(params, LweKey(params.in_out_params, dev(thr, lwe_key)))


def ciphertext_from_arrays(thr, ct, params=None):
    params = params or NuFHEParameters()
    a, b, cv = ct
    lwe_params = params.in_out_params if a.shape[-1] == params.in_out_params.size else (
        params.tgsw_params.tlwe_params.extracted_lweparams)
    return LweSampleArray(lwe_params, dev(thr, a), dev(thr, b), dev(thr, cv))


def ct_arrays(ct):
    return host(ct.a), host(ct.b), host(ct.current_variances)
