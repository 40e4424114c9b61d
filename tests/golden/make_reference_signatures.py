"""
Writes tests/golden/reference_signatures.json: the positional parameter names of the reference's functions on the
bootstrap path and the public attributes of its containers, read from the reference's source with
`ast` (nothing is imported or executed).  Run in the build container, where /root/reference exists:
    python tests/golden/make_reference_signatures.py
tests/test_cabi_and_host.py::test_low_level_names_and_signatures_match_the_reference compares nufhe_amd with the file.
"""
import ast
import json
import os

REF = '/root/reference/nufhe'
FUNCTIONS = {
    'numeric_functions': ['phase_to_t32', 't32_to_phase', 'double_to_t32'],
    'polynomials': ['shift_tp_inverted_power', 'shift_tp_minus_one_power_from_array'],
    'tlwe': ['tlwe_noiseless_trivial', 'tlwe_extract_lwe_samples', 'tlwe_shift_polynomials', 'tlwe_add_to', 'tlwe_copy',
             'tlwe_encrypt_zero', 'tlwe_transform_samples'],
    'tgsw': ['tgsw_transformed_external_mul', 'tgsw_encrypt_zero', 'tgsw_add_message', 'tgsw_encrypt_int',
             'tgsw_transform_samples'],
    'bootstrap': ['mux_rotate', 'blind_rotate', 'blind_rotate_and_extract', 'bootstrap'],
    'lwe': ['lwe_keyswitch', 'lwe_noiseless_trivial', 'lwe_noiseless_trivial_constant', 'lwe_negate', 'lwe_copy',
            'lwe_add_to', 'lwe_add_mul_to', 'lwe_sub_to', 'lwe_sub_mul_to', 'lwe_encrypt', 'lwe_decrypt'],
    'gates': ['gate_nand', 'gate_or', 'gate_and', 'gate_xor', 'gate_xnor', 'gate_not', 'gate_copy', 'gate_constant',
              'gate_nor', 'gate_andny', 'gate_andyn', 'gate_orny', 'gate_oryn', 'gate_mux'],
}
CLASSES = {
    'polynomials': {'TorusPolynomialArray': ['empty'], 'TransformedPolynomialArray': ['empty', 'dump', 'load']},
    'tlwe': {'TLweSampleArray': ['empty'], 'TransformedTLweSampleArray': ['empty', 'dump', 'load'], 'TLweParams': [],
             'TLweKey': ['from_rng']},
    'tgsw': {'TGswParams': [], 'TGswKey': ['from_rng'], 'TGswSampleArray': ['empty'], 'TransformedTGswSampleArray': ['empty']},
    'bootstrap': {'BootstrapKey': ['from_rng', 'dump', 'load']},
    'lwe': {'LweSampleArray': ['empty', 'copy', 'roll', 'dump', 'dumps', 'load', 'loads'],
            'LweKeyswitchKey': ['from_tgsw_key', 'dump', 'load'], 'LweParams': [], 'LweKey': ['from_rng', 'from_tlwe_key']},
}


def main():
    out = {'functions': {}, 'classes': {}}
    for module, names in FUNCTIONS.items():
        tree = ast.parse(open(os.path.join(REF, module + '.py')).read())
        defs = {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef)}
        out['functions'][module] = {name: [a.arg for a in defs[name].args.args] for name in names}
    for module, classes in CLASSES.items():
        tree = ast.parse(open(os.path.join(REF, module + '.py')).read())
        defs = {n.name: n for n in tree.body if isinstance(n, ast.ClassDef)}
        out['classes'][module] = {}
        for cname, methods in classes.items():
            have = {m.name for m in defs[cname].body if isinstance(m, ast.FunctionDef)}
            missing = [m for m in methods if m not in have]
            assert not missing, (module, cname, missing)
            out['classes'][module][cname] = methods
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_signatures.json')
    with open(path, 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print('wrote', path)


if __name__ == '__main__':
    main()
