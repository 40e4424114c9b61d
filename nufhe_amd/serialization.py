"""
Reference-compatible serialization.

The reference serializes keys and ciphertexts as a *sequence of pickle records* (parameter objects
and host NumPy arrays), in this order:

  ciphertext  (lwe.py:207-214):            LweParams, a, b, current_variances
  secret key  (api_low_level.py:116-121):  NuFHEParameters, [LweKey (lwe.py:92-94):] LweParams, key
  cloud key   (api_low_level.py:198-204):  NuFHEParameters,
      BootstrapKey (bootstrap.py:78-80):   LweParams (in/out),
        TransformedTGswSampleArray (tgsw.py:116-118):   TGswParams,
          TransformedTLweSampleArray (tlwe.py:135-138): TLweParams,
            TransformedPolynomialArray (polynomials.py:72-74): transform_type (str), coeffs
                 -- uint64 [n,k+1,l,k+1,N] natural-order NTT, Montgomery form, or complex128 [..., N/2]
            current_variances float32 [n, k+1, l]
      LweKeyswitchKey (lwe.py:297-298) = LweSampleArray: LweParams, a [N k,t,base,n], b, current_variances

The parameter classes pickle under the REFERENCE's module paths (``nufhe.lwe.LweParams`` ...):
their ``__module__`` is set accordingly and the top-level alias package ``nufhe`` re-exports them,
so streams written here are readable by the reference and vice versa.  The functions below work on
host arrays only (no GPU needed); the device classes call them.
"""

import io
import pickle

import numpy


def write_ciphertext(file_obj, params, a, b, current_variances):
    pickle.dump(params, file_obj)
    pickle.dump(numpy.ascontiguousarray(a), file_obj)
    pickle.dump(numpy.ascontiguousarray(b), file_obj)
    pickle.dump(numpy.ascontiguousarray(current_variances), file_obj)


def read_ciphertext(file_obj):
    params = pickle.load(file_obj)
    a = pickle.load(file_obj)
    b = pickle.load(file_obj)
    cv = pickle.load(file_obj)
    return params, a, b, cv


def write_bootstrap_key(file_obj, in_out_params, bk_params, coeffs, current_variances):
    tlwe_params = bk_params.tlwe_params
    pickle.dump(in_out_params, file_obj)                # BootstrapKey.dump
    pickle.dump(bk_params, file_obj)                    # TransformedTGswSampleArray.dump
    pickle.dump(tlwe_params, file_obj)                  # TransformedTLweSampleArray.dump
    pickle.dump(tlwe_params.transform_type, file_obj)   # TransformedPolynomialArray.dump
    pickle.dump(numpy.ascontiguousarray(coeffs), file_obj)
    pickle.dump(numpy.ascontiguousarray(current_variances, numpy.float32), file_obj)


def read_bootstrap_key(file_obj):
    in_out_params = pickle.load(file_obj)
    bk_params = pickle.load(file_obj)
    tlwe_params = pickle.load(file_obj)
    transform_type = pickle.load(file_obj)
    coeffs = pickle.load(file_obj)
    current_variances = pickle.load(file_obj)
    if transform_type != tlwe_params.transform_type or tlwe_params != bk_params.tlwe_params:
        raise ValueError("inconsistent bootstrap key stream")
    return in_out_params, bk_params, coeffs, current_variances


def bootstrap_key_variances(in_out_params, bk_params):
    """current_variances of the transformed TGSW samples: TLweEncryptZero fills noise^2
    (tlwe_cpu.py:86) and TLweTransformSamples copies them (tlwe.py:207)."""
    k1 = bk_params.tlwe_params.mask_size + 1
    cv = numpy.empty((in_out_params.size, k1, bk_params.decomp_length), numpy.float32)
    cv.fill(bk_params.tlwe_params.min_noise**2)
    return cv


def record_structure(data: bytes):
    """[(kind, detail)] of every pickle record of a stream, with the global names each record
    references -- used by the tests to compare a stream with the reference's documented layout
    without importing the reference."""
    out = []
    file_obj = io.BytesIO(data)
    while file_obj.tell() < len(data):
        start = file_obj.tell()
        obj = pickle.load(file_obj)
        end = file_obj.tell()
        import pickletools
        names = []
        for op, arg, _ in pickletools.genops(data[start:end]):
            if op.name == 'GLOBAL':
                names.append(arg.replace(' ', '.'))
            elif op.name == 'STACK_GLOBAL':
                names.append('<stack_global>')
        if isinstance(obj, numpy.ndarray):
            out.append(('ndarray', (str(obj.dtype), obj.shape)))
        elif isinstance(obj, str):
            out.append(('str', obj))
        else:
            out.append((type(obj).__module__ + '.' + type(obj).__qualname__, tuple(sorted(vars(obj)))))
    return out


def canonical_records(data: bytes):
    """The pickle records of a stream in a form that compares by CONTENT: arrays as (dtype, shape, bytes),
    parameter objects as (class path, sorted attributes), recursively.  Two streams with equal canonical
    records unpickle to the same objects; their bytes may still differ in pickle memo references (e.g. a
    parameter record that was loaded and written again no longer shares one dtype object between its
    NumPy fields)."""
    def canon(obj):
        if isinstance(obj, numpy.ndarray):
            return ('ndarray', obj.dtype.str, obj.shape, numpy.ascontiguousarray(obj).tobytes())
        if isinstance(obj, numpy.generic):
            return ('scalar', obj.dtype.str, obj.item())
        if isinstance(obj, (tuple, list)):
            return (type(obj).__name__,) + tuple(canon(x) for x in obj)
        if isinstance(obj, dict):
            return ('dict',) + tuple(sorted((k, canon(v)) for k, v in obj.items()))
        if hasattr(obj, '__dict__'):
            return (type(obj).__module__ + '.' + type(obj).__qualname__,) + tuple(
                sorted((k, canon(v)) for k, v in vars(obj).items()))
        return obj
    out = []
    file_obj = io.BytesIO(data)
    while file_obj.tell() < len(data):
        out.append(canon(pickle.load(file_obj)))
    return out
