"""
make_golden_serialization.py -- streams written by the REFERENCE's own serialization code
(nufhe/api_low_level.py:116-148,198-232, bootstrap.py:78-86, tgsw.py:116-124, tlwe.py:135-145,
polynomials.py:72-80, lwe.py:92-103,207-243), for tests/test_serialization_layout.py and the -m gpu test
tests/test_gpu_serialization.py.

    python tests/golden/make_golden_serialization.py        # needs /root/reference (build container only)

How: the reference package is imported AS `nufhe` from where it lies (/root/reference first on sys.path,
this repository's alias package of the same name kept out), with `reikna` replaced by a permissive stub
(every attribute is a dummy class; the reference's class DEFINITIONS only need the names to exist at
import time -- no kernel is ever built).  The reference's own classes are then instantiated around host
arrays from the CPU oracle (a NumPy subclass with the `.get()` of a Reikna array) and the reference's
own `dump` methods write the bytes.  The same script also lets the reference's `load` read the bytes
back (its own, and a stream assembled by this repository's host-side writer) and records the verdicts.

To keep the fixture small (the real cloud key is 98 MB) the LWE dimension is n = 8; everything else is
the default parameter set.  Output: tests/golden/reference_serialized/*.bin + manifest.json.
"""
import io
import json
import os
import pickle
import sys
import types

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE_ROOT = os.environ.get("NUFHE_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(HERE, "reference_serialized")
N_SMALL = 8


class _StubMeta(type):
    def __getattr__(cls, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _Stub


class _Stub(metaclass=_StubMeta):
    """Stands for any reikna name: usable as a base class, callable, subscriptable, iterable-safe."""

    def __init__(self, *args, **kwds):
        pass

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _Stub()

    def __call__(self, *args, **kwds):
        return _Stub()

    def __getitem__(self, item):
        return _Stub()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _Stub


class HostArray(numpy.ndarray):
    """host array with the .get() of a Reikna device array"""

    def get(self):
        return numpy.array(self)


def host(x):
    return numpy.ascontiguousarray(x).view(HostArray)


class HostThread:
    """the two Thread methods the reference's load() functions call"""

    def to_device(self, arr):
        return host(arr)

    def array(self, shape, dtype):
        return host(numpy.empty(shape, dtype))


def import_reference():
    for name in list(sys.modules):
        if name == 'nufhe' or name.startswith('nufhe.'):
            del sys.modules[name]
    for name in ('reikna', 'reikna.core', 'reikna.cluda', 'reikna.cluda.api', 'reikna.cluda.dtypes',
                 'reikna.cluda.functions', 'reikna.helpers', 'reikna.algorithms', 'reikna.transformations',
                 'reikna.core.computation', 'reikna.core.signature', 'reikna.cluda.cuda', 'reikna.cluda.ocl'):
        sys.modules[name] = _StubModule(name)
    sys.modules['reikna'].__path__ = []
    # Type.from_value must compare equal for equal (shape, dtype): the ciphertext shape-info uses it
    class Type:
        def __init__(self, dtype, shape=None):
            self.dtype = numpy.dtype(dtype); self.shape = tuple(shape or ())
        @classmethod
        def from_value(cls, v):
            return cls(v.dtype, v.shape)
        def __eq__(self, o):
            return isinstance(o, Type) and self.dtype == o.dtype and self.shape == o.shape
        def __hash__(self):
            return hash((self.dtype, self.shape))
    sys.modules['reikna.core'].Type = Type
    sys.path = [p for p in sys.path if os.path.abspath(p or '.') != ROOT]
    sys.path.insert(0, REFERENCE_ROOT)
    import nufhe
    assert os.path.abspath(nufhe.__file__).startswith(os.path.abspath(REFERENCE_ROOT)), nufhe.__file__
    return nufhe


def oracle_material():
    """keys and ciphertexts from the CPU oracle at n = N_SMALL (reference formats, host arrays)"""
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    params = orc.Params(lwe_size=N_SMALL)
    lwe_key, tlwe_key, ck = orc.make_key_pair(orc.DeterministicRNG(2024), params)
    rng = orc.DeterministicRNG(77)
    m1 = rng.uniform_bool((3, 5)).astype(bool); m2 = rng.uniform_bool((3, 5)).astype(bool)
    c1 = orc.encrypt(rng, lwe_key, m1, params); c2 = orc.encrypt(rng, lwe_key, m2, params)
    nand = orc.gate('gate_nand', ck, c1, c2)
    assert (orc.decrypt(lwe_key, nand) == ~(m1 & m2)).all()
    sys.path.remove(ROOT)
    for name in list(sys.modules):
        if name == 'oracle' or name.startswith('oracle.'):
            pass
    return dict(lwe_key=lwe_key, ck=ck, m1=m1, m2=m2, c1=c1, c2=c2, nand=nand)


def main():
    mat = oracle_material()
    nufhe = import_reference()
    from nufhe.api_low_level import NuFHEParameters, NuFHESecretKey, NuFHECloudKey
    from nufhe.lwe import LweParams, LweKey, LweSampleArray, LweKeyswitchKey
    from nufhe.bootstrap import BootstrapKey
    from nufhe.tgsw import TransformedTGswSampleArray
    from nufhe.tlwe import TransformedTLweSampleArray
    from nufhe.polynomials import TransformedPolynomialArray
    # the synthetic polynomial_transform of the stubbed package would be a stub: give the two methods
    # TransformedPolynomialArray needs their documented values (polynomial_transform_ntt.py:29-42)
    import nufhe.polynomials as polys
    polys.get_transform = lambda t: types.SimpleNamespace(
        transformed_dtype=lambda: numpy.dtype('uint64' if t == 'NTT' else 'complex128'),
        transformed_length=lambda N: N if t == 'NTT' else N // 2)

    params = NuFHEParameters()                                      # the reference's own defaults ...
    full = params.in_out_params
    params.in_out_params = LweParams(N_SMALL, full.min_noise, full.max_noise)   # ... at a small LWE dimension
    lwe_params = params.in_out_params
    ck = mat['ck']

    secret_key = NuFHESecretKey(params, LweKey(lwe_params, host(mat['lwe_key'].astype(numpy.int32))))
    k1 = 2
    bk_cv = numpy.full((N_SMALL, k1, 2), numpy.float32(params.tgsw_params.tlwe_params.min_noise**2), numpy.float32)
    tgsw = TransformedTGswSampleArray(
        params.tgsw_params,
        TransformedTLweSampleArray(params.tgsw_params.tlwe_params,
                                   TransformedPolynomialArray('NTT', host(ck.bk)), host(bk_cv)))
    cloud_key = NuFHECloudKey(
        params, BootstrapKey(lwe_params, tgsw),
        LweKeyswitchKey(LweSampleArray(lwe_params, host(ck.ks_a), host(ck.ks_b), host(ck.ks_cv))))
    cts = {name: LweSampleArray(lwe_params, host(c[0]), host(c[1]), host(c[2]))
           for name, c in (('ct1', mat['c1']), ('ct2', mat['c2']), ('nand', mat['nand']))}

    os.makedirs(OUT, exist_ok=True)
    blobs = {'secret_key': secret_key.dumps(), 'cloud_key': cloud_key.dumps()}
    blobs.update({name: ct.dumps() for name, ct in cts.items()})
    for name, data in blobs.items():
        with open(os.path.join(OUT, name + '.bin'), 'wb') as f:
            f.write(data)

    # the reference's own load() reads its streams back ...
    thr = HostThread()
    assert NuFHESecretKey.loads(blobs['secret_key'], thr) == secret_key
    assert NuFHECloudKey.loads(blobs['cloud_key'], thr) == cloud_key
    assert LweSampleArray.loads(blobs['ct1'], thr) == cts['ct1']

    # ... and a stream assembled by THIS repository's host-side writer from the same arrays
    # (nufhe_amd/serialization.py; the device classes call exactly these functions)
    sys.path.insert(1, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location('_our_serialization', os.path.join(ROOT, 'nufhe_amd', 'serialization.py'))
    ours = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ours)
    buf = io.BytesIO()
    pickle.dump(params, buf)
    ours.write_bootstrap_key(buf, lwe_params, params.tgsw_params, ck.bk, bk_cv)
    ours.write_ciphertext(buf, lwe_params, ck.ks_a, ck.ks_b, ck.ks_cv)
    our_cloud = buf.getvalue()
    reference_reads_ours = NuFHECloudKey.loads(our_cloud, thr) == cloud_key
    buf = io.BytesIO()
    ours.write_ciphertext(buf, lwe_params, *mat['c1'])
    reference_reads_our_ct = LweSampleArray.loads(buf.getvalue(), thr) == cts['ct1']

    manifest = {
        'lwe_size': N_SMALL,
        'm1': mat['m1'].astype(int).tolist(), 'm2': mat['m2'].astype(int).tolist(),
        'files': {name: len(data) for name, data in blobs.items()},
        'reference_load_reads_reference_stream': True,
        'reference_load_reads_our_cloud_key_stream': bool(reference_reads_ours),
        'reference_load_reads_our_ciphertext_stream': bool(reference_reads_our_ct),
        'our_host_writer_bytes_equal_reference_bytes': bool(our_cloud == blobs['cloud_key']),
        'generator': 'tests/golden/make_golden_serialization.py (reference classes imported from /root/reference '
                     'under a reikna stub; arrays from oracle.make_key_pair(DeterministicRNG(2024), n=8))',
    }
    with open(os.path.join(OUT, 'manifest.json'), 'w') as f:
        json.dump(manifest, f, indent=1)
    print(json.dumps({k: v for k, v in manifest.items() if k not in ('m1', 'm2')}, indent=1))


if __name__ == '__main__':
    main()
