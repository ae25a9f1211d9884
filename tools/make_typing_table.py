#!/usr/bin/env python3
"""Regenerate arpeggio_amd/core/data/prot_atom_types.json from the reference's config.py (build container only).

The reference types the atoms of the twenty standard residues from a dictionary, `PROT_ATOM_TYPES` (config.py:150-590:
atom type -> list of '<RESNAME><ATOMNAME>' keys), not from SMARTS (interactions.py:1966-1983).  That dictionary is DATA of
the drop-in contract — including its accidents: two places where a missing comma fuses two keys ('GLNOE1GLNNE2',
'TRPCD1TRPCE3', which therefore match no atom) and the 'TRYCB' spelling — so it is kept as data, inverted into
key -> list of type names, together with STD_RES (config.py:147), the key sets address_ambiguities() removes
(interactions.py:129-133), VALENCE (config.py:41-45, read at interactions.py:1804) and the solvent / nucleotide names the
LIGANDS selector skips (config.py:664-710).

    python tools/make_typing_table.py
"""
import importlib.util
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location('ref_config', '/root/reference/arpeggio/core/config.py')
cfg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(cfg)

inv = {}
for type_name, keys in cfg.PROT_ATOM_TYPES.items():
    for k in keys:
        inv.setdefault(k, [])
        if type_name not in inv[k]:
            inv[k].append(type_name)
out = {
    'source': 'arpeggio/core/config.py of pdbe-arpeggio 1.4.4: PROT_ATOM_TYPES (150-590), STD_RES (147), VALENCE (41-45); '
              'interactions.py:129-133 for the ambiguity sets',
    'std_res': sorted(cfg.STD_RES),
    'table_types': list(cfg.PROT_ATOM_TYPES.keys()),
    'keys': {k: inv[k] for k in sorted(inv)},
    'ambiguous': {'hbond acceptor': ['ASNND2', 'GLNNE2', 'HISCE1', 'HISCD2'], 'hbond donor': ['ASNOD1', 'GLNOE1', 'HISCE1', 'HISCD2'],
                  'xbond acceptor': ['ASNND2', 'GLNNE2', 'HISCE1', 'HISCD2'], 'weak hbond acceptor': ['ASNND2', 'GLNNE2', 'HISCE1', 'HISCD2']},
    'valence': list(cfg.VALENCE),
}
path = os.path.join(HERE, '..', 'arpeggio_amd', 'core', 'data', 'prot_atom_types.json')
json.dump(out, open(path, 'w'), indent=0, sort_keys=True)
print(len(out['keys']), 'keys,', len(out['std_res']), 'residues ->', os.path.normpath(path))

# _chem_comp.type -> one-letter component type (config.py:1143-1200, used by protein_reader.get_component_types): the same
# kind of contract data.  The method builds its dictionary in its body, so it is asked for every key found in its source.
import ast
import inspect
src = inspect.getsource(cfg.ComponentType.from_chem_comp_type)
keys = [n.value for n in ast.walk(ast.parse('class _:\n' + src if src.startswith('    ') else src)) if isinstance(n, ast.Constant) and isinstance(n.value, str)
        and n.value.upper() == n.value and len(n.value) > 4 and 'Maps' not in n.value]
table = {}
for k in keys:
    try:
        table[k] = cfg.ComponentType.from_chem_comp_type(k)
    except KeyError:
        pass
out2 = {'source': 'arpeggio/core/config.py:1143-1200 (ComponentType.from_chem_comp_type), pdbe-arpeggio 1.4.4',
        'letters': {m.name: int(m.value) for m in cfg.ComponentType}, 'chem_comp_type': {k: table[k] for k in sorted(table)}}
path2 = os.path.join(HERE, '..', 'arpeggio_amd', 'core', 'data', 'chem_comp_types.json')
json.dump(out2, open(path2, 'w'), indent=0, sort_keys=True)
print(len(table), 'chem_comp types ->', os.path.normpath(path2))
