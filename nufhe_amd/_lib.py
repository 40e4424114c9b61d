"""
ctypes binding of libnufhe_hip.so (the C ABI of include/nufhe_hip.h).

There is NO fallback: if the shared library is missing or fails to load, importing this module's
``lib()`` raises -- the product path never routes through a CPU implementation.
"""

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NUFHE_HIP_LIBRARY: load another build of the same library (kernel-tuning experiments, tools/)
LIB_PATH = os.environ.get("NUFHE_HIP_LIBRARY") or os.path.join(_HERE, "libnufhe_hip.so")


class NufheError(RuntimeError):
    pass


class NufheLwe(ctypes.Structure):
    """nufhe_lwe of include/nufhe_hip.h"""
    _fields_ = [
        ("a", ctypes.c_void_p),
        ("b", ctypes.c_void_p),
        ("cv", ctypes.c_void_p),
        ("a_stride", ctypes.c_long),
        ("b_stride", ctypes.c_long),
        ("size", ctypes.c_int32),
    ]


class NufheGateJob(ctypes.Structure):
    """nufhe_gate_job of include/nufhe_hip.h (one gate of a nufhe_gate_batch call)"""
    _fields_ = [
        ("kind", ctypes.c_int32),
        ("c0", ctypes.c_int32),
        ("pa", ctypes.c_int32),
        ("pb", ctypes.c_int32),
        ("nbits", ctypes.c_long),
        ("result", NufheLwe),
        ("a", NufheLwe),
        ("b", NufheLwe),
        ("c", NufheLwe),
    ]


class NufheTuning(ctypes.Structure):
    """nufhe_tuning of include/nufhe_hip.h (batch-size switch points of one context)"""
    _fields_ = [
        ("team_max_bits", ctypes.c_long),
        ("team_max_bits_fft", ctypes.c_long),
        ("pair_max_bits_ntt", ctypes.c_long),
        ("pair_max_bits_fft", ctypes.c_long),
        ("ks_mfma_min_bits", ctypes.c_long),
        ("ring_k2", ctypes.c_int32),
        ("k2_roomy_ratio_pct", ctypes.c_int32),
        ("measured", ctypes.c_int32),
        ("num_cus", ctypes.c_int32),
        ("arch_name", ctypes.c_char * 64),
    ]


ENGINE_NATIVE = 0
ENGINE_EXACT_FFT = 1
ENGINES = {'native': ENGINE_NATIVE, 'ntt': ENGINE_NATIVE, 'exact-fft': ENGINE_EXACT_FFT, 'exact_fft': ENGINE_EXACT_FFT}

JOB_BINARY = 0
JOB_MUX = 1

_vp = ctypes.c_void_p
_i32 = ctypes.c_int32
_int = ctypes.c_int
_long = ctypes.c_long
_pp = ctypes.POINTER(ctypes.c_void_p)

# name -> argtypes (every entry point returns int, except the three string/pointer getters)
PROTOTYPES = {
    "nufhe_device_count": [ctypes.POINTER(_int)],
    "nufhe_device_name": [_int, ctypes.c_char_p, ctypes.c_size_t],
    "nufhe_ctx_create": [_int, _vp, _int, _pp],
    "nufhe_ctx_destroy": [_vp],
    "nufhe_ctx_synchronize": [_vp],
    "nufhe_ctx_device": [_vp, ctypes.POINTER(_int)],
    "nufhe_alloc": [_vp, ctypes.c_size_t, _pp],
    "nufhe_free": [_vp, _vp],
    "nufhe_h2d": [_vp, _vp, _vp, ctypes.c_size_t],
    "nufhe_d2h": [_vp, _vp, _vp, ctypes.c_size_t],
    "nufhe_cloudkey_create": [_vp, _int, _int, _int, _pp],
    "nufhe_gather": [_vp, _vp, ctypes.POINTER(ctypes.c_size_t), _pp, _pp, ctypes.POINTER(ctypes.c_size_t), _int],
    "nufhe_cloudkey_image_bytes": [_vp, ctypes.POINTER(ctypes.c_size_t)],
    "nufhe_cloudkey_export_image": [_vp, _vp],
    "nufhe_cloudkey_import_image": [_vp, _vp],
    "nufhe_ctx_set_team_max_bits": [_vp, _long],
    "nufhe_ctx_set_pair_max_bits": [_vp, _long],
    "nufhe_ctx_set_team8": [_vp, _int],
    "nufhe_ctx_set_keyswitch_mfma": [_vp, _int],
    "nufhe_ff_op": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _long],
    "nufhe_ks_make": [_vp, _vp, _vp, _vp, _vp, ctypes.c_float],
    "nufhe_ks_download_reference": [_vp, _vp, _vp, _vp],
    "nufhe_tgsw_add_message": [_vp, _vp, _vp, _long, _int],
    "nufhe_l4_op": [_vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _long],
    "nufhe_cloudkey_destroy": [_vp],
    "nufhe_cloudkey_set_engine": [_vp, _int],
    "nufhe_cloudkey_get_engine": [_vp, ctypes.POINTER(_int)],
    "nufhe_bk_upload_reference": [_vp, _vp],
    "nufhe_bk_download_reference": [_vp, _vp],
    "nufhe_bk_from_coeffs": [_vp, _vp],
    "nufhe_ks_upload": [_vp, _vp, _vp, _vp],
    "nufhe_lwe_linear": [_vp, NufheLwe, NufheLwe, _i32, _int, _long, _int],
    "nufhe_lwe_trivial_const": [_vp, NufheLwe, _i32, _long, _int],
    "nufhe_bootstrap": [_vp, _vp, NufheLwe, NufheLwe, _i32, _long, _int],
    "nufhe_keyswitch": [_vp, _vp, NufheLwe, NufheLwe, _long],
    "nufhe_gate_binary": [_vp, _vp, NufheLwe, NufheLwe, NufheLwe, _i32, _i32, _i32, _i32, _long],
    "nufhe_gate_mux": [_vp, _vp, NufheLwe, NufheLwe, NufheLwe, NufheLwe, _long],
    "nufhe_gate_batch": [_vp, _vp, ctypes.POINTER(NufheGateJob), _int, _i32],
    "nufhe_ctx_pin_scratch": [_vp, _int],
    "nufhe_ctx_get_tuning": [_vp, ctypes.POINTER(NufheTuning)],
    "nufhe_ctx_set_tuning": [_vp, ctypes.POINTER(NufheTuning)],
    "nufhe_lwe_phase": [_vp, _vp, _long, _vp, _long, _vp, _long, _vp, _i32, _long, _int],
    "nufhe_t32_to_phase": [_vp, _vp, _vp, _long, ctypes.c_uint32],
    "nufhe_shift_torus_polynomial": [_vp, _vp, _vp, _vp, _long, _long, _long, _int, _int, _int],
    "nufhe_tlwe_extract": [_vp, _vp, _vp, _vp, _long, _int],
    "nufhe_tgsw_decompose": [_vp, _vp, _vp, _long],
    "nufhe_tgsw_mac": [_vp, _vp, _vp, _vp, _int, _int, _long, _int],
    "nufhe_ntt_forward_i32": [_vp, _vp, _vp, _long],
    "nufhe_ntt_forward_u64": [_vp, _vp, _vp, _long],
    "nufhe_ntt_inverse_i32": [_vp, _vp, _vp, _long],
    "nufhe_ntt_inverse_u64": [_vp, _vp, _vp, _long],
    "nufhe_poly_mul_i32": [_vp, _vp, _vp, _vp, _long, _long],
    "nufhe_fft_forward_i32": [_vp, _vp, _vp, _long],
    "nufhe_fft_inverse_i32": [_vp, _vp, _vp, _long],
    "nufhe_external_mul": [_vp, _vp, _vp, _int, _long],
    "nufhe_blind_rotate": [_vp, _vp, _vp, _vp, _long, _int, _long],
    "nufhe_tlwe_encrypt_zero": [_vp, _vp, _vp, _vp, _vp, _long, _int],
    "nufhe_profile_enable": [_vp, _int],
    "nufhe_profile_last": [_vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)],
    "nufhe_profile_history": [_vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), _int, ctypes.POINTER(_int)],
    "nufhe_profile_clock": [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)],
    "nufhe_profile_waves": [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                            ctypes.POINTER(_int), _int, ctypes.POINTER(_int)],
}

# NUFHE_ABI_VERSION of the include/nufhe_hip.h these prototypes were written against
ABI_VERSION = 6

_lib = None


def lib():
    """Loads libnufhe_hip.so (built by __graft_entry__.build() / nufhe_amd/csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NufheError(
                "libnufhe_hip.so not found at %s: build it with `make -C nufhe_amd/csrc` "
                "(there is no CPU fallback)" % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        try:
            handle.nufhe_abi_version.restype = ctypes.c_int
            handle.nufhe_abi_version.argtypes = []
            abi = handle.nufhe_abi_version()
        except AttributeError:
            abi = None
        if abi != ABI_VERSION:
            raise NufheError("%s has ABI version %s, this binding needs %d (structs are passed by value: rebuild with "
                             "`make -C nufhe_amd/csrc`)" % (LIB_PATH, abi, ABI_VERSION))
        for name, argtypes in PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        handle.nufhe_last_error.restype = ctypes.c_char_p
        handle.nufhe_last_error.argtypes = []
        handle.nufhe_version.restype = ctypes.c_char_p
        handle.nufhe_version.argtypes = []
        handle.nufhe_ctx_stream.restype = ctypes.c_void_p
        handle.nufhe_ctx_stream.argtypes = [_vp]
        _lib = handle
    return _lib


def check(rc):
    """Maps a C status code to the Python exception types the reference raises."""
    if rc == 0:
        return
    msg = lib().nufhe_last_error().decode("utf-8", "replace")
    if rc == -1:
        raise ValueError(msg)
    if rc == -5:
        raise MemoryError(msg)
    raise NufheError("nufhe_hip error %d: %s" % (rc, msg))


# context / cloud-key handle value -> callable that raises when the caller's torch stream is not the stream the
# library enqueues on (DeviceThread.check_stream).  Centralised here so that EVERY entry point that hands device
# memory to the library is guarded, not only the ones that remember to ask.
_stream_guards = {}


def register_stream_guard(handle, guard):
    _stream_guards[int(handle.value)] = guard


def unregister_stream_guard(handle):
    if handle is not None and handle.value is not None:
        _stream_guards.pop(int(handle.value), None)


def call(name, *args):
    if args and isinstance(args[0], ctypes.c_void_p) and args[0].value is not None:
        guard = _stream_guards.get(int(args[0].value))
        if guard is not None:
            guard()
    check(getattr(lib(), name)(*args))
