"""
Reference-compatible serialization, CPU only: the record sequence, the class
paths inside the pickles and the array formats are the reference's (nufhe/api_low_level.py:116-232,
lwe.py:92-100,207-243,297-303, bootstrap.py:78-86, tgsw.py:116-124, tlwe.py:135-145,
polynomials.py:72-80): checked structurally, and against streams written by the reference's own dump
code (tests/golden/reference_serialized, made by tests/golden/make_golden_serialization.py).
"""

import io
import os
import pickle
import pickletools

import numpy
import pytest


def _module_strings(data):
    """every module-like string (starting with a package name) inside a pickle stream"""
    out = set()
    f = io.BytesIO(data)
    while f.tell() < len(data):
        start = f.tell()
        pickle.load(f)
        for op, arg, _ in pickletools.genops(data[start:f.tell()]):
            if op.name in ('SHORT_BINUNICODE', 'BINUNICODE', 'UNICODE') and arg.startswith('nufhe'):
                out.add(arg)
            elif op.name == 'GLOBAL' and arg.startswith('nufhe'):
                out.add(arg.split(' ')[0])
    return out


def test_cloud_key_stream_layout(orc, oracle_keys):
    import nufhe                       # the alias package
    from nufhe_amd import serialization as S
    from nufhe_amd.lwe import HostLweSampleArray
    lwe_key, tlwe_key, ck = oracle_keys
    params = nufhe.NuFHEParameters()
    f = io.BytesIO()
    pickle.dump(params, f)                                                  # NuFHECloudKey.dump
    S.write_bootstrap_key(f, params.in_out_params, params.tgsw_params, ck.bk,
                          S.bootstrap_key_variances(params.in_out_params, params.tgsw_params))
    HostLweSampleArray(params.in_out_params, ck.ks_a, ck.ks_b, ck.ks_cv).dump(f)   # LweKeyswitchKey.dump
    data = f.getvalue()
    recs = S.record_structure(data)
    kinds = [r[0] for r in recs]
    assert kinds == [
        'nufhe.api_low_level.NuFHEParameters',     # api_low_level.py:202
        'nufhe.lwe.LweParams',                     # bootstrap.py:79
        'nufhe.tgsw.TGswParams',                   # tgsw.py:117
        'nufhe.tlwe.TLweParams',                   # tlwe.py:136
        'str',                                     # polynomials.py:73 transform_type
        'ndarray',                                 # polynomials.py:74 coeffs
        'ndarray',                                 # tlwe.py:138 current_variances
        'nufhe.lwe.LweParams', 'ndarray', 'ndarray', 'ndarray',   # lwe.py:211-214
    ]
    assert recs[4][1] == 'NTT'
    assert recs[5][1] == ('uint64', (500, 2, 2, 2, 1024))
    assert recs[6][1] == ('float32', (500, 2, 2))
    assert recs[8][1] == ('int32', (1024, 8, 4, 500)) and recs[9][1] == ('int32', (1024, 8, 4))
    assert recs[10][1] == ('float32', (1024, 8, 4))
    # attribute names of the parameter records are the reference's
    assert recs[0][1] == ('_tlwe_mask_size', '_transform_type', 'in_out_params', 'ks_decomp_length',
                          'ks_log2_base', 'tgsw_params')
    assert recs[1][1] == ('max_noise', 'min_noise', 'size')
    assert recs[2][1] == ('base_powers', 'bs_log2_base', 'decomp_length', 'offset', 'tlwe_params')
    assert recs[3][1] == ('extracted_lweparams', 'mask_size', 'max_noise', 'min_noise',
                          'polynomial_degree', 'transform_type')
    # every class named inside the stream lives under the reference's package name
    assert b'nufhe_amd' not in data
    mods = _module_strings(data)
    assert mods == {'nufhe.api_low_level', 'nufhe.lwe', 'nufhe.tgsw', 'nufhe.tlwe'}, mods
    # and the stream reads back
    g = io.BytesIO(data)
    p2 = pickle.load(g)
    in_out, bkp, coeffs, cv = S.read_bootstrap_key(g)
    assert p2 == params and in_out == params.in_out_params and bkp == params.tgsw_params
    assert (coeffs == ck.bk).all() and (cv == numpy.float32(params.tgsw_params.tlwe_params.min_noise**2)).all()


def test_ciphertext_and_secret_key_stream_layout(orc, oracle_keys):
    import nufhe
    from nufhe_amd import serialization as S
    lwe_key, tlwe_key, ck = oracle_keys
    params = nufhe.NuFHEParameters()
    ct = orc.encrypt(orc.DeterministicRNG(1), lwe_key, [True, False, True])
    f = io.BytesIO()
    S.write_ciphertext(f, params.in_out_params, *ct)
    recs = S.record_structure(f.getvalue())
    assert [r[0] for r in recs] == ['nufhe.lwe.LweParams', 'ndarray', 'ndarray', 'ndarray']
    assert recs[1][1] == ('int32', (3, 500)) and recs[2][1] == ('int32', (3,)) and recs[3][1] == ('float32', (3,))
    # secret key: NuFHEParameters, LweParams, key (api_low_level.py:120-121, lwe.py:93-94)
    f = io.BytesIO()
    pickle.dump(params, f); pickle.dump(params.in_out_params, f); pickle.dump(lwe_key, f)
    recs = S.record_structure(f.getvalue())
    assert [r[0] for r in recs] == ['nufhe.api_low_level.NuFHEParameters', 'nufhe.lwe.LweParams', 'ndarray']


def test_import_nufhe_alias_surface():
    """`import nufhe` exposes the reference's public names (nufhe/__init__.py:18-59)."""
    import nufhe
    for name in ('make_key_pair', 'encrypt', 'decrypt', 'empty_ciphertext', 'NuFHEParameters',
                 'NuFHESecretKey', 'NuFHECloudKey', 'LweSampleArray', 'concatenate',
                 'gate_nand', 'gate_or', 'gate_and', 'gate_xor', 'gate_xnor', 'gate_not', 'gate_copy',
                 'gate_constant', 'gate_nor', 'gate_andny', 'gate_andyn', 'gate_orny', 'gate_oryn',
                 'gate_mux', 'PerformanceParameters', 'DeterministicRNG', 'SecureRNG',
                 'clear_computation_cache', 'find_devices', 'Context'):
        assert hasattr(nufhe, name), name
    import nufhe.lwe, nufhe.gates, nufhe.bootstrap, nufhe.api_low_level   # noqa: E401
    assert nufhe.lwe.LweParams is nufhe.api_low_level.LweParams


# ---- streams written by the REFERENCE's own dump code (tests/golden/make_golden_serialization.py) ----

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_serialized')


def _golden(name):
    with open(os.path.join(GOLDEN_DIR, name + '.bin'), 'rb') as f:
        return f.read()


def test_reference_written_streams_parse_with_our_readers(orc):
    """The bytes the reference's own NuFHESecretKey / NuFHECloudKey / LweSampleArray.dump wrote (n = 8
    fixture) unpickle into THIS package's parameter classes through the `nufhe` alias package and carry
    exactly the arrays the oracle generates from the same seed; our host-side writer reproduces the
    reference's bytes."""
    import io
    import json
    import pickle
    from nufhe_amd import serialization, NuFHEParameters
    from nufhe_amd.lwe import LweParams
    manifest = json.load(open(os.path.join(GOLDEN_DIR, 'manifest.json')))
    assert manifest['reference_load_reads_our_cloud_key_stream'] and manifest['our_host_writer_bytes_equal_reference_bytes']
    n = manifest['lwe_size']
    params = orc.Params(lwe_size=n)
    lwe_key, tlwe_key, ck = orc.make_key_pair(orc.DeterministicRNG(2024), params)

    f = io.BytesIO(_golden('secret_key'))
    p = pickle.load(f)
    assert type(p) is NuFHEParameters and p.in_out_params.size == n and p == NuFHEParameters()
    kp = pickle.load(f); key = pickle.load(f)
    assert type(kp) is LweParams and kp.size == n and (key == lwe_key).all() and f.read() == b''

    f = io.BytesIO(_golden('cloud_key'))
    p = pickle.load(f)
    in_out, bk_params, coeffs, cv = serialization.read_bootstrap_key(f)
    assert in_out.size == n and bk_params == p.tgsw_params
    assert coeffs.dtype == numpy.uint64 and coeffs.shape == (n, 2, 2, 2, 1024) and (coeffs == ck.bk).all()
    assert cv.shape == (n, 2, 2) and (cv == numpy.float32(bk_params.tlwe_params.min_noise**2)).all()
    ksp, a, b, kcv = serialization.read_ciphertext(f)
    assert ksp.size == n and (a == ck.ks_a).all() and (b == ck.ks_b).all() and (kcv == ck.ks_cv).all()
    assert f.read() == b''
    # our writer, same arrays -> the reference's records (byte-identical when the parameter object is
    # freshly built, as the generator checks; a loaded-and-rewritten parameter record only differs in
    # pickle memo references, hence the content comparison here)
    out = io.BytesIO()
    pickle.dump(p, out)
    serialization.write_bootstrap_key(out, in_out, bk_params, coeffs, cv)
    serialization.write_ciphertext(out, ksp, a, b, kcv)
    assert serialization.canonical_records(out.getvalue()) == serialization.canonical_records(_golden('cloud_key'))
    assert out.getvalue()[-2_000_000 + 300_000:] == _golden('cloud_key')[-2_000_000 + 300_000:]   # the array records

    rng = orc.DeterministicRNG(77)
    m1 = rng.uniform_bool((3, 5)).astype(bool); m2 = rng.uniform_bool((3, 5)).astype(bool)
    assert m1.astype(int).tolist() == manifest['m1'] and m2.astype(int).tolist() == manifest['m2']
    c1 = orc.encrypt(rng, lwe_key, m1, params); c2 = orc.encrypt(rng, lwe_key, m2, params)
    for name, exp in (('ct1', c1), ('ct2', c2), ('nand', orc.gate('gate_nand', ck, c1, c2))):
        cp, a, b, cv = serialization.read_ciphertext(io.BytesIO(_golden(name)))
        assert cp.size == n and (a == exp[0]).all() and (b == exp[1]).all() and (cv == exp[2]).all()
