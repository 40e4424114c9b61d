"""
The boundary from plain C: examples/c_abi_gate.c is compiled with gcc against include/nufhe_hip.h and linked to
libnufhe_hip.so -- no Python, no torch, no HIP headers on the host side -- and runs NAND / MUX gates on a key handed over
in the reference's array formats.  Its output words must equal the CPU oracle's.  ("extern C, POD args only";
INTEGRATION.md shows the ctypes form of the same calls.)
"""
import os
import subprocess

import numpy
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def c_host(tmp_path_factory):
    out = tmp_path_factory.mktemp('c_host') / 'c_abi_gate'
    lib_dir = os.path.join(ROOT, 'nufhe_amd')
    cmd = ['gcc', '-std=c99', '-O2', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'),
           os.path.join(ROOT, 'examples', 'c_abi_gate.c'), '-L', lib_dir, '-lnufhe_hip', '-Wl,-rpath,' + lib_dir, '-o', str(out)]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr
    return str(out)


def test_c_host_program_builds_without_a_gpu(c_host):
    """gcc -Wall -Werror over the public header from C99, linked against the library (CPU-only check: the program is not run)"""
    assert os.access(c_host, os.X_OK)
    header = open(os.path.join(ROOT, 'include', 'nufhe_hip.h')).read()
    import re
    assert sorted(re.findall(r'#include\s+<([^>]+)>', header)) == ['stddef.h', 'stdint.h']      # plain C, nothing else


@pytest.mark.gpu
@pytest.mark.parametrize('gate,transform,nbits', [(0, 0, 37), (1, 0, 5), (0, 1, 300), (0, 0, 0), (2, 0, 41), (2, 1, 7)])
def test_c_host_gates_equal_the_oracle(c_host, orc, oracle_keys, tmp_path, gate, transform, nbits):
    lwe_key, tlwe_key, ck = oracle_keys
    n = 500
    rng = orc.DeterministicRNG(900 + nbits)
    ms = [rng.uniform_bool((nbits,)).astype(bool) for _ in range(3)]
    cs = [orc.encrypt(rng, lwe_key, m) for m in ms]
    if transform == 1:
        from oracle import oracle_fft as of          # the same TGSW samples in the FFT domain (reference format)
        bk = numpy.ascontiguousarray(of.bk_from_coeffs(of.tgsw_coeffs_from_reference_bk(ck.bk)), numpy.complex128)
    else:
        bk = numpy.ascontiguousarray(ck.bk, numpy.uint64)
    with open(tmp_path / 'in.bin', 'wb') as f:
        numpy.array([n, nbits, transform, gate], numpy.int32).tofile(f)
        bk.tofile(f)
        numpy.ascontiguousarray(ck.ks_a, numpy.int32).tofile(f)
        numpy.ascontiguousarray(ck.ks_b, numpy.int32).tofile(f)
        numpy.ascontiguousarray(ck.ks_cv, numpy.float32).tofile(f)
        for c in cs:
            numpy.ascontiguousarray(c[0], numpy.int32).tofile(f)
            numpy.ascontiguousarray(c[1], numpy.int32).tofile(f)
    proc = subprocess.run([c_host, str(tmp_path / 'in.bin'), str(tmp_path / 'out.bin')], capture_output=True, text=True,
                          timeout=600)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    assert 'c_abi_gate OK' in proc.stdout and 'LWE size 501, expected 500' in proc.stdout
    raw = numpy.fromfile(tmp_path / 'out.bin', numpy.int32)
    a = raw[:nbits * n].reshape(nbits, n); b = raw[nbits * n:nbits * n + nbits]
    cv = raw[nbits * n + nbits:].view(numpy.float32)
    if gate == 2:
        # one nufhe_gate_batch call: NAND on the first half of the bits, MUX on the rest
        h = nbits // 2
        lo = orc.gate('gate_nand', ck, tuple(x[:h] for x in cs[0]), tuple(x[:h] for x in cs[1]))
        hi = orc.gate_mux(ck, *[tuple(x[h:] for x in c) for c in cs])
        exp = tuple(numpy.concatenate([l, u]) for l, u in zip(lo, hi))
        assert 'switch points of gfx950' in proc.stdout
    else:
        exp = orc.gate_mux(ck, cs[0], cs[1], cs[2]) if gate == 1 else orc.gate('gate_nand', ck, cs[0], cs[1])
    da = (a.astype(numpy.int64) - exp[0].astype(numpy.int64) + 2**31) % 2**32 - 2**31
    db = (b.astype(numpy.int64) - exp[1].astype(numpy.int64) + 2**31) % 2**32 - 2**31
    tol = 0 if transform == 0 else 16
    assert nbits == 0 or (abs(da).max() <= tol and abs(db).max() <= tol)
    assert (cv == exp[2]).all()
