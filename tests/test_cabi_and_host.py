"""
CPU-only checks of the product's host side: the C-ABI shared library loads and exports every
symbol include/nufhe_hip.h declares (no compute calls without a GPU), the product fails loudly
without a GPU, and the host logic (shapes, parameters, RNG, torus helpers) behaves like the
reference's.
"""

import ctypes
import os
import re
import sys

import numpy
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def built_lib():
    import __graft_entry__
    if not os.path.exists(os.path.join(ROOT, 'nufhe_amd', 'libnufhe_hip.so')):
        __graft_entry__.build()
    from nufhe_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol(built_lib):
    header = open(os.path.join(ROOT, 'include', 'nufhe_hip.h')).read()
    declared = set(re.findall(r'\b(nufhe_[a-z0-9_]+)\s*\(', header))
    declared -= {'nufhe_ctx', 'nufhe_cloudkey', 'nufhe_lwe'}
    assert len(declared) >= 35
    lib = ctypes.CDLL(built_lib.LIB_PATH)
    missing = [name for name in sorted(declared) if not hasattr(lib, name)]
    assert not missing, missing
    # the ctypes prototypes cover the header
    proto = set(built_lib.PROTOTYPES) | {'nufhe_last_error', 'nufhe_version', 'nufhe_ctx_stream', 'nufhe_abi_version'}
    assert declared <= proto, sorted(declared - proto)
    # the binding, the header and the binary agree on the ABI version (structs are passed by value)
    macro = int(re.search(r'#define\s+NUFHE_ABI_VERSION\s+(\d+)', header).group(1))
    lib.nufhe_abi_version.restype = ctypes.c_int
    assert macro == built_lib.ABI_VERSION == lib.nufhe_abi_version()
    lib.nufhe_version.restype = ctypes.c_char_p
    assert b'0.6' in lib.nufhe_version()


def test_no_cpu_fallback(built_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import nufhe_amd
    lib = built_lib.lib()
    assert lib.nufhe_version().startswith(b'nufhe_hip')
    count = ctypes.c_int(-1)
    rc = lib.nufhe_device_count(ctypes.byref(count))
    assert rc != 0 and count.value == 0
    with pytest.raises((built_lib.NufheError, ValueError)):
        nufhe_amd.Context()


def test_product_does_not_import_oracle():
    """The product package never imports, includes or loads anything under oracle/ or tests/."""
    pkg = os.path.join(ROOT, 'nufhe_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith(('.py', '.hip', '.h', '.cpp')):
                continue
            for line in open(os.path.join(dirpath, f)).read().splitlines():
                low = line.lower()
                if 'oracle' in low or 'emu' in low.split('//')[0]:
                    assert not re.search(r'^\s*(from|import|#\s*include)\b', line), (f, line)
                    assert 'cdll' not in low and 'dlopen' not in low, (f, line)


def test_phase_to_t32_wraps():
    from nufhe_amd.numeric_functions import phase_to_t32, double_to_t32
    # intended values (NumPy 1.x wrapped silently)
    assert phase_to_t32(1, 8) == 2**29
    assert phase_to_t32(-1, 8) == -2**29
    assert phase_to_t32(-1, 4) == -2**30
    assert phase_to_t32(1, 4) == 2**30
    assert double_to_t32(numpy.array([0.0, 2.0**-15]))[1] == 2**17


def test_result_shape_rules():
    # nufhe/gates.py:51-78 / test_api_high_level.py:135-172
    from nufhe_amd.gates import result_shape, check_shape

    class S:
        def __init__(self, shape): self.shape = shape
    assert result_shape((3, 4), (4,)) == (3, 4)
    assert result_shape((1, 4), (3, 1)) == (3, 4)
    assert result_shape((2, 3, 4), (4,), (3, 1)) == (2, 3, 4)
    with pytest.raises(ValueError):
        result_shape((3, 4), (5,))
    check_shape(S((2, 3, 4)), S((3, 4)), S((4,)))
    with pytest.raises(ValueError):
        check_shape(S((3, 4)), S((2, 3, 4)))


def test_parameters_match_reference_defaults():
    # nufhe/api_low_level.py:49-61
    from nufhe_amd import NuFHEParameters
    p = NuFHEParameters()
    assert p.in_out_params.size == 500
    assert p.tgsw_params.tlwe_params.polynomial_degree == 1024
    assert p.tgsw_params.tlwe_params.mask_size == 1
    assert p.tgsw_params.decomp_length == 2 and p.tgsw_params.bs_log2_base == 10
    assert p.ks_decomp_length == 8 and p.ks_log2_base == 2
    assert int(p.tgsw_params.offset) == -2145386496   # (int32)(2^31 + 2^21), tgsw.py:49-52
    assert p.tgsw_params.tlwe_params.extracted_lweparams.size == 1024
    assert p == NuFHEParameters() and hash(p) == hash(NuFHEParameters())
    assert NuFHEParameters(transform_type='FFT') != p
    p2 = NuFHEParameters(tlwe_mask_size=2)
    assert p2.tgsw_params.tlwe_params.extracted_lweparams.size == 2048 and p2 != p
    p3 = NuFHEParameters(transform_type='FFT', tlwe_mask_size=2)      # every (transform, k <= 2) pair exists
    assert p3.tgsw_params.tlwe_params.extracted_lweparams.size == 2048 and p3 != p2
    with pytest.raises(NotImplementedError):
        NuFHEParameters(tlwe_mask_size=3)


def test_rng_order_matches_oracle(orc):
    """DeterministicRNG draws (random_numbers.py:46-62) are the oracle's / the reference's."""
    from nufhe_amd.random_numbers import DeterministicRNG, rand_gaussian_torus32_host
    r1 = DeterministicRNG(9); r2 = orc.DeterministicRNG(9)
    assert (r1.uniform_bool((7,)) == r2.uniform_bool((7,))).all()
    assert (r1.uniform_torus32((3, 5)) == r2.uniform_torus32((3, 5))).all()
    a = rand_gaussian_torus32_host(r1, 0, 1e-4, (4, 6), centered=True)
    b = orc.rand_gaussian_torus32(r2, 0, 1e-4, (4, 6), centered=True)
    assert (a == b).all()


def test_secure_rng_shapes():
    from nufhe_amd import SecureRNG
    r = SecureRNG()
    assert r.uniform_bool((3, 5)).shape == (3, 5) and set(numpy.unique(r.uniform_bool((64,)))) <= {0, 1}
    assert r.uniform_torus32((4,)).dtype == numpy.int32
    g = r.gauss((1001,), 2.0)
    assert g.shape == (1001,) and 1.0 < g.std() < 3.0


def test_ciphertext_view_indexing_with_ellipsis():
    """Views are indexed like the plaintext array (lwe.py:163-172): an Ellipsis addresses the
    message axes only, never the trailing mask axis."""
    import torch
    from nufhe_amd.lwe import LweSampleArray, LweParams
    p = LweParams(500, 0., 1.)
    a = torch.arange(4 * 16 * 500, dtype=torch.int32).reshape(4, 16, 500)
    ct = LweSampleArray(p, a, torch.zeros(4, 16, dtype=torch.int32), torch.zeros(4, 16))
    v = ct[..., 3:4]
    assert v.shape == (4, 1) and tuple(v.a.shape) == (4, 1, 500) and (v.a == a[:, 3:4, :]).all()
    assert ct[:, 3:4].shape == (4, 1) and ct[...].shape == (4, 16) and ct[1].shape == (16,)
    ct[..., 0:1] = ct[..., 5:6]
    assert (ct.a[:, 0] == a[:, 5]).all()


def test_tracked_profile_inputs_of_the_bench_roofline_are_current():
    """bench.py's issue roofline multiplies the ISA counts of profiles/isa_mix.json by the measured
    class costs of profiles/valu_issue_costs.json: the counts must belong to the CURRENT device sources
    (tools/isa_mix.py --write records a hash of nufhe_amd/csrc)."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location('isa_mix', os.path.join(ROOT, 'tools', 'isa_mix.py'))
    isa_mix = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(isa_mix)
    mix = json.load(open(os.path.join(ROOT, 'profiles', 'isa_mix.json')))
    assert mix['source_hash'] == isa_mix.source_hash(), "run `python tools/isa_mix.py --write` (device sources changed)"
    for kernel in ('k_bootstrap<1>', 'k_bootstrap_fft'):
        k = mix[kernel]
        assert k['valu'] > 1000 and k['vgprs'] <= 256
    # the loop of the NTT kernel is allowed a handful of spilled dwords (5 stores + 5 loads per iteration measured
    # faster than the spill-free code generation), not the hundreds of bytes of a lost allocation
    assert mix['k_bootstrap<1>']['scratch_bytes'] <= 64 and mix['k_bootstrap<1>'].get('scratch', 0) <= 16
    for kernel in ('k_bootstrap<1>', 'k_bootstrap_fft'):
        assert mix[kernel]['valu_plain'] + mix[kernel]['valu_other'] == mix[kernel]['valu']
    costs = json.load(open(os.path.join(ROOT, 'profiles', 'valu_issue_costs.json')))
    assert 2.0 <= costs['machine_plain_cycles'] < costs['machine_other_cycles'] <= 5.0
    assert costs['two_wave_plain_cycles'] < 2.5 and costs['plain_cycles']['1'] > 4.0      # pairing needs two waves
    assert os.path.isfile(os.path.join(ROOT, 'profiles', costs['source']))
    for name in ('pmc_NTT.json', 'pmc_FFT.json', 'pmc_traffic.json'):
        assert os.path.isfile(os.path.join(ROOT, 'profiles', name))


def test_keyswitch_byte_plane_formulation_is_exact():
    """The arithmetic behind k_keyswitch_mfma / k_ks_planes (kernels.hip), restated in numpy: the keyswitch sum as a
    one-hot matrix product against the key split into four balanced signed byte planes, recombined with shifts, equals
    the direct row gather mod 2^32 -- including key words at the int32 extremes, where the balanced split carries."""
    rs = numpy.random.RandomState(12)
    J, n, bits = 16, 24, 20                        # coefficients, output columns, ciphertext bits (reduced sizes)
    ks = rs.randint(-2**31, 2**31, size=(J, 8, 4, n), dtype=numpy.int64)
    ks[:, :, 0, :] = 0                             # digit 0 selects the all-zero row (lwe_cpu.py:30-33)
    ks[0, 0, 1, :6] = [2**31 - 1, -2**31, -1, 127, 128, -129]
    a = rs.randint(-2**31, 2**31, size=(bits, J), dtype=numpy.int64)
    ap = (a + 2**15) % 2**32
    digits = numpy.stack([(ap >> (30 - 2 * k)) & 3 for k in range(8)], axis=-1)          # [bits, J, 8]
    direct = numpy.zeros((bits, n), numpy.int64)
    for b in range(bits):
        for j in range(J):
            for k in range(8):
                direct[b] += ks[j, k, digits[b, j, k]]
    direct = (-direct) % 2**32
    # balanced byte planes, as k_ks_planes builds them (everything mod 2^32)
    v = ks % 2**32
    planes = []
    for p in range(4):
        s8 = ((v & 0xFF) ^ 0x80) - 0x80                                                    # sign-extended low byte
        planes.append(s8)
        v = (((v - s8) % 2**32) ^ 2**31) - 2**31 >> 8                                      # arithmetic shift of the int32
        v = v % 2**32
    assert all(abs(pl).max() <= 128 for pl in planes)
    onehot = (digits[..., None] == numpy.arange(4)).astype(numpy.int64)                    # [bits, J, 8, 4]
    total = numpy.zeros((bits, n), numpy.int64)
    for p, pl in enumerate(planes):
        cp = numpy.einsum('bjkd,jkdn->bn', onehot, pl)                                     # what the MFMAs accumulate
        assert abs(cp).max() < 2**21 * (J * 8) // 8192 + 2**12                             # far inside int32
        total += cp << (8 * p)
    assert ((-total) % 2**32 == direct).all()


def test_bench_roofline_helpers_on_tracked_profiles():
    """bench.py's roofline section is computed from tracked files after the timed region: a KeyError there would lose a
    whole GPU run.  Exercise it with the kernel times of round 3 and check that every fraction names its denominator."""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module('bench')
    for transform, ms, clock in (('NTT', 39.4, 2.39), ('FFT', 12.4, 2.26)):
        r = bench.issue_roofline(transform, 4096, 1, 500, ms, clock)
        assert r['frac_is'] == 'vs_measured_class_rates' and 0.3 < r['frac'] < 1.0
        assert abs(r['frac'] - r['fractions']['vs_measured_class_rates']['frac']) < 1e-12
        for name, f in r['fractions'].items():
            assert 0.0 < f['frac'] < 1.5 and f['denominator'] and f['numerator'], name
        assert r['fractions']['vs_nominal_2_cycle_issue']['frac'] < r['frac']
        assert r['isa_mix_per_iteration']['valu_plain'] + r['isa_mix_per_iteration']['valu_other'] == r['isa_mix_per_iteration']['valu']
    assert 'algorithmic' in bench.issue_roofline('NTT', 4096, 1, 500, 39.4, 2.39)['fractions']
    k2 = bench.keyswitch_roofline(4096, 0.30, True)
    assert k2['bound'] == 'mfma' and 0.2 < k2['frac'] < 0.6
    assert bench.keyswitch_roofline(256, 0.2, False)['bound'] == 'lds'
    assert bench.pmc_traffic('NTT', 'nand', 4096) > 1e8


def test_low_level_names_and_signatures_match_the_reference():
    """The functions of the bootstrap path exist under the reference's module and function names and take the
    reference's positional parameters in the reference's order (trailing optional parameters may be added), and the
    containers have its constructors / (de)serializers: tests/golden/reference_signatures.json, extracted from the
    reference's source by tests/golden/make_reference_signatures.py.  The `nufhe` alias package re-exports them."""
    import importlib
    import inspect
    import json
    ref = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'reference_signatures.json')))
    for package in ('nufhe_amd', 'nufhe'):
        for module, functions in ref['functions'].items():
            mod = importlib.import_module(package + '.' + module)
            for name, args in functions.items():
                have = list(inspect.signature(getattr(mod, name)).parameters)
                assert have[:len(args)] == args, (package, module, name, args, have)
                extra = list(inspect.signature(getattr(mod, name)).parameters.values())[len(args):]
                assert all(p.default is not inspect.Parameter.empty for p in extra), (module, name)
        for module, classes in ref['classes'].items():
            mod = importlib.import_module(package + '.' + module)
            for cname, methods in classes.items():
                cls = getattr(mod, cname)
                assert all(callable(getattr(cls, m)) for m in methods), (module, cname)


def test_asm_sched_is_a_parser_only():
    """tools/asm_sched.py lost its (wrong-code) reordering modes in round 6; what tools/isa_mix.py needs from it is the
    operand / issue-class parser."""
    import importlib.util
    tool = os.path.join(ROOT, 'tools', 'asm_sched.py')
    spec = importlib.util.spec_from_file_location('asm_sched', tool)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.Inst('v_add_u32_e32 v0, v1, v2', 0).plain and not mod.Inst('v_mad_u64_u32 v[0:1], vcc, v2, v3, v[4:5]', 0).plain
    assert not hasattr(mod, 'schedule_region') and not hasattr(mod, 'main')


def test_ctypes_structs_match_the_c_header(tmp_path):
    """The structs that cross the boundary by value / by pointer (nufhe_lwe, nufhe_gate_job, nufhe_tuning) have the same
    size and field offsets in the ctypes binding as in include/nufhe_hip.h compiled by a plain C compiler."""
    import ctypes
    import subprocess
    from nufhe_amd import _lib
    structs = {'nufhe_lwe': (_lib.NufheLwe, ['a', 'b', 'cv', 'a_stride', 'b_stride', 'size']),
               'nufhe_gate_job': (_lib.NufheGateJob, ['kind', 'c0', 'pa', 'pb', 'nbits', 'result', 'a', 'b', 'c']),
               'nufhe_tuning': (_lib.NufheTuning, ['team_max_bits', 'team_max_bits_fft', 'pair_max_bits_ntt', 'pair_max_bits_fft',
                                                   'ks_mfma_min_bits', 'ring_k2', 'k2_roomy_ratio_pct', 'measured', 'num_cus',
                                                   'arch_name'])}
    lines = ['#include <stddef.h>', '#include <stdio.h>', '#include "nufhe_hip.h"', 'int main(void) {']
    for name, (_, fields) in structs.items():
        lines.append('printf("%s %%zu", sizeof(%s));' % (name, name))
        for f in fields:
            lines.append('printf(" %%zu", offsetof(%s, %s));' % (name, f))
        lines.append('printf("\\n");')
    lines += ['printf("abi %d\\n", NUFHE_ABI_VERSION);', 'return 0; }']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-std=c99', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    out = subprocess.check_output([str(exe)], text=True).split('\n')
    for line in out:
        parts = line.split()
        if not parts:
            continue
        if parts[0] == 'abi':
            assert int(parts[1]) == _lib.ABI_VERSION
            continue
        cls, fields = structs[parts[0]]
        assert int(parts[1]) == ctypes.sizeof(cls), parts[0]
        for f, off in zip(fields, parts[2:]):
            assert getattr(cls, f).offset == int(off), (parts[0], f)


def test_gate_batch_refuses_malformed_job_lists_before_touching_the_device():
    """nufhe_amd.gates.gate_batch checks names and arity of every job first (no device object is needed to be told that
    gate_not is not a bootstrapped gate)."""
    from nufhe_amd import gates
    assert set(gates.BINARY_GATES) == {'gate_nand', 'gate_or', 'gate_and', 'gate_xor', 'gate_xnor', 'gate_nor',
                                       'gate_andny', 'gate_andyn', 'gate_orny', 'gate_oryn'}
    for name, (c, pa, pb) in gates.BINARY_GATES.items():
        assert abs(pa) == abs(pb) and abs(pa) in (1, 2) and c in (2**29, -2**29, 2**30, -2**30), name
    with pytest.raises(ValueError, match='not a bootstrapped gate'):
        gates.gate_batch(None, None, [('gate_not', 1, 2, 3)])
    with pytest.raises(ValueError, match='takes 4 ciphertext arguments'):
        gates.gate_batch(None, None, [('gate_mux', 1, 2, 3)])
    with pytest.raises(ValueError, match='takes 3 ciphertext arguments'):
        gates.gate_batch(None, None, [('gate_nand', 1, 2)])


def test_bench_traffic_lookup_covers_every_baseline_configuration():
    """roofline.traffic must not be null on a BASELINE configuration: NAND 4096 (configs 2, 4, 5), MUX 4096 (config 3) and
    the 2048-bit lines all have counter passes in profiles/pmc_traffic.json"""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    argv = sys.argv
    sys.argv = ['bench.py']
    try:
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    for tr in ('NTT', 'FFT'):
        for gate, bits in (('nand', 4096), ('mux', 4096), ('nand', 2048)):
            t = bench.pmc_traffic(tr, gate, bits)
            assert t is not None and 1e8 < t < 3e9, (tr, gate, bits)
        # a MUX launch moves twice the key traffic of a NAND launch
        assert 1.8 < bench.pmc_traffic(tr, 'mux', 4096) / bench.pmc_traffic(tr, 'nand', 4096) < 2.2
    assert bench.pmc_traffic('NTT', 'nand', 77) is None


def test_every_entry_point_catches_cpp_exceptions():
    """include/nufhe_hip.h: "no exceptions cross the boundary" -- every `int nufhe_*` body of api.hip sits between
    NUFHE_API_BEGIN / NUFHE_API_END (std::bad_alloc -> NUFHE_ENOMEM), and the binding maps -5 to MemoryError."""
    src = open(os.path.join(ROOT, 'nufhe_amd', 'csrc', 'api.hip')).read()
    body = src[src.index('extern "C" {'):]
    defs = re.findall(r'^int (nufhe_\w+)\([^;{]*\)\n\{\n(.*?)^\}', body, re.S | re.M)
    assert len(defs) >= 60
    for name, text in defs:
        assert text.lstrip().startswith('NUFHE_API_BEGIN') and text.rstrip().endswith('NUFHE_API_END'), name
    header = open(os.path.join(ROOT, 'include', 'nufhe_hip.h')).read()
    assert re.search(r'#define\s+NUFHE_ENOMEM\s+\(-5\)', header)
    from nufhe_amd import _lib
    import inspect
    assert 'MemoryError' in inspect.getsource(_lib.check)


def test_no_scalar_offset_rewritten_behind_a_buffer_instruction_in_the_exact_fft_unit():
    """round 6 machine fact (profiles/r06_xfft_experiments.txt 4b): an SGPR read as `soffset` by a buffer instruction and
    rewritten right behind it gave wrong words with two waves per SIMD.  The exact-FFT unit is compiled to ISA here and
    scanned (4 s); `python tools/check_soffset_war.py` scans every unit (1 min)."""
    import importlib.util
    import subprocess
    spec = importlib.util.spec_from_file_location('check_soffset_war', os.path.join(ROOT, 'tools', 'check_soffset_war.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    asm = '/tmp/test_war_xfft.s'
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-S', '--cuda-device-only',
                           'kernels_xfft.hip', '-o', asm], cwd=os.path.join(ROOT, 'nufhe_amd', 'csrc'), stderr=subprocess.DEVNULL)
    lines = open(asm).read().split('\n')
    assert sum(1 for l in lines if 'buffer_load_dwordx4' in l) >= 128
    assert mod.scan(lines, 4) == []
    # the scanner does find the pattern
    assert mod.scan(['_Zf:', '\tbuffer_load_dwordx4 v[2:5], v1, s[20:23], s2 offen', '\ts_movk_i32 s2, 0x800'], 4)
