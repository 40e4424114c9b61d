"""
Whole-gate checks of the CPU oracle at the default parameters (n=500, N=1024):
 * against golden outputs of the REFERENCE's own CPU functions composed in the reference's driver
   order (tests/golden/make_golden_gate.py; B=2, NAND and MUX),
 * decrypt == truth table for every gate (test/test_gates.py:178-245 of the reference).
CPU only.
"""

import os

import numpy
import pytest

import make_golden_gate


@pytest.fixture(scope='module')
def gate_golden():
    return numpy.load(os.path.join(os.path.dirname(make_golden_gate.__file__), 'reference_gate_outputs.npz'))


@pytest.fixture(scope='module')
def gate_inputs():
    return make_golden_gate.gate_inputs()


def test_oracle_nand_vs_reference_full_size(orc, gate_golden, gate_inputs):
    lwe_key, tlwe_key, ck, cts, ms = gate_inputs
    MU = 2**29
    ta = (-cts[0][0] - cts[1][0]).astype(numpy.int32)
    tb = (numpy.int32(MU) - cts[0][1] - cts[1][1]).astype(numpy.int32)
    ea, eb = orc.bootstrap_extract(ck.bk, ta, tb, MU)
    assert (ea == gate_golden['nand_ext_a']).all() and (eb == gate_golden['nand_ext_b']).all()
    ra, rb, rcv = orc.gate('gate_nand', ck, cts[0], cts[1])
    assert (ra == gate_golden['nand_a']).all()
    assert (rb == gate_golden['nand_b']).all()
    assert (rcv == gate_golden['nand_cv']).all()


def test_oracle_mux_vs_reference_full_size(orc, gate_golden, gate_inputs):
    lwe_key, tlwe_key, ck, cts, ms = gate_inputs
    ra, rb, rcv = orc.gate_mux(ck, cts[0], cts[1], cts[2])
    assert (ra == gate_golden['mux_a']).all()
    assert (rb == gate_golden['mux_b']).all()
    assert (rcv == gate_golden['mux_cv']).all()


TRUTH = {
    'gate_nand': lambda a, b: ~(a & b),
    'gate_or': lambda a, b: a | b,
    'gate_and': lambda a, b: a & b,
    'gate_nor': lambda a, b: ~(a | b),
    'gate_xor': lambda a, b: a ^ b,
    'gate_xnor': lambda a, b: ~(a ^ b),
    'gate_andny': lambda a, b: ~a & b,
    'gate_andyn': lambda a, b: a & ~b,
    'gate_orny': lambda a, b: ~a | b,
    'gate_oryn': lambda a, b: a | ~b,
}


def test_oracle_truth_tables(orc, oracle_keys):
    """BASELINE config 1 on the CPU path: 32-bit batch NAND (and every sibling gate on 8 bits)."""
    lwe_key, tlwe_key, ck = oracle_keys
    rng = orc.DeterministicRNG(456)
    m1 = rng.uniform_bool((32,)).astype(bool); m2 = rng.uniform_bool((32,)).astype(bool)
    c1 = orc.encrypt(rng, lwe_key, m1); c2 = orc.encrypt(rng, lwe_key, m2)
    assert (orc.decrypt(lwe_key, c1) == m1).all()
    r = orc.gate('gate_nand', ck, c1, c2)
    assert (orc.decrypt(lwe_key, r) == ~(m1 & m2)).all()
    s1 = tuple(x[:8] for x in c1); s2 = tuple(x[:8] for x in c2)
    for name, fn in TRUTH.items():
        if name == 'gate_nand':
            continue
        r = orc.gate(name, ck, s1, s2)
        assert (orc.decrypt(lwe_key, r) == fn(m1[:8], m2[:8])).all(), name
    m3 = rng.uniform_bool((8,)).astype(bool)
    c3 = orc.encrypt(rng, lwe_key, m3)
    r = orc.gate_mux(ck, s1, s2, c3)
    assert (orc.decrypt(lwe_key, r) == numpy.where(m1[:8], m2[:8], m3)).all()
