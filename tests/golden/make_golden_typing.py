#!/usr/bin/env python3
"""Golden vectors for the table-driven atom typing, produced by EXECUTING the reference's own
``InteractionComplex._ob_atom_typing`` (interactions.py:1923-1983), ``address_ambiguities`` (120-133) and
``_extend_atom_properties`` (1985-1991) on data holders (build container only; needs /root/reference).

The SMARTS engine of OpenBabel is replaced by a holder whose ``GetMapList`` returns a stored, deterministic set of atoms
per pattern (no chemistry): what is pinned is everything the reference does AROUND the matcher — union over patterns,
the water rule, and the dictionary override for the twenty standard residues, typos of the dictionary included.

    python tests/golden/make_golden_typing.py
"""
import json
import os
import sys
import types
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get('ARP_GOLDEN_OUT', HERE)      # (tests/test_fixture_freshness.py regenerates into a scratch directory)
sys.path.insert(0, HERE)
import make_golden_core as core   # noqa: E402  (holders + extraction helpers)

REF = core.REF


def main():
    import logging
    logging.disable(logging.CRITICAL)
    out = {}
    for variant in ('default', 'ambiguities'):
        config = core.load_by_path('ref_config_' + variant, os.path.join(REF, 'config.py'))   # fresh module: address_ambiguities mutates it
        # atoms: every '<RES><ATOM>' key of the dictionary that starts with a standard residue name, some atoms the
        # dictionary does not know, non-standard residues, waters
        names = {}
        for keys in config.PROT_ATOM_TYPES.values():
            for k in keys:
                if k[:3] in config.STD_RES:
                    names.setdefault(k[:3], set()).add(k[3:])
        rows = []
        for res in sorted(config.STD_RES):
            for an in sorted(names.get(res, set()) | {'OXT', 'H', 'XX1'}):
                rows.append((res, an, False))
        for res, ats in (('HEM', ['FE', 'NA', 'C1A', 'O1A']), ('MSE', ['SE', 'CB', 'N', 'O']), ('TRY', ['CB']), ('GLN ', ['OE1']),
                         ('gln', ['NE2'])):
            rows += [(res, an, False) for an in ats]
        rows += [('HOH', 'O', True), ('HOH', 'H1', True), ('DOD', 'O', True)]

        chains = {'A': core.Chain('A')}
        residues, atoms = [], []
        for k, (res, an, water) in enumerate(rows):
            if not residues or residues[-1].resname != res or water:
                residues.append(core.Residue(len(residues), chains['A'], res, 'W' if water else ' ', len(residues) + 1, ' '))
            a = core.Atom(k, residues[-1], an, 'C', np.zeros(3, np.float32), k + 1, water)
            atoms.append(a)
        ob_atoms = {k + 1: types.SimpleNamespace(GetId=(lambda k=k: 5000 + k)) for k in range(len(atoms))}   # 1-based GetAtom

        class SmartsPattern:
            def Init(self, smarts):
                self.smarts = smarts

            def Match(self, mol):
                return True

            def GetMapList(self):
                h = zlib.crc32(self.smarts.encode())
                hits = [k + 1 for k in range(len(atoms)) if (k * 2654435761 + h) % 7 == 0]
                return [(x,) for x in hits]

        ob = types.SimpleNamespace(OBSmartsPattern=SmartsPattern)
        import operator
        from functools import reduce
        ns = {'np': np, 'logging': logging, 'config': config, 'ob': ob, 'reduce': reduce, 'operator': operator}
        IC = core.compile_class(os.path.join(REF, 'interactions.py'), 'InteractionComplex',
                                ['_ob_atom_typing', 'address_ambiguities', '_extend_atom_properties'], ns)
        self_ = IC.__new__(IC)
        self_.s_atoms = atoms
        self_.biopython_str = core.Structure(residues)
        self_.ob_mol = types.SimpleNamespace(GetAtom=lambda idx: ob_atoms[idx])
        self_.ob_to_bio = {5000 + k: a for k, a in enumerate(atoms)}
        if variant == 'ambiguities':
            self_.address_ambiguities()
        self_._extend_atom_properties()
        # what the matcher alone said (the product receives this as the incoming mask of every atom)
        smarts_only = [set() for _ in atoms]
        for atom_type, smartsdict in config.ATOM_TYPES.items():
            for smarts in smartsdict.values():
                p = SmartsPattern()
                p.Init(str(smarts))
                for (idx,) in p.GetMapList():
                    smarts_only[idx - 1].add(atom_type)
        self_._ob_atom_typing()
        out[variant] = {
            'atoms': [{'resname': r, 'name': n, 'water': w} for r, n, w in rows],
            'smarts_types': [sorted(s) for s in smarts_only],
            'final_types': [sorted(a.atom_types) for a in atoms],
        }
        assert any(x != y for x, y in zip(out[variant]['smarts_types'], out[variant]['final_types']))
    # element flags of _extend_atom_properties
    config = core.load_by_path('ref_config_flags', os.path.join(REF, 'config.py'))
    elems = ['C', 'N', 'O', 'S', 'H', 'FE', 'Fe', 'ZN', 'CL', 'Cl', 'BR', 'I', 'F', 'MG', 'CA', 'NA', 'K', 'SE', 'P', 'D', 'W', 'U', 'AT', 'XX']
    res = core.Residue(0, core.Chain('A'), 'XXX', ' ', 1, ' ')
    flagged = [core.Atom(k, res, 'X', e, np.zeros(3, np.float32), k + 1, False) for k, e in enumerate(elems)]
    ns = {'np': np, 'logging': logging, 'config': config}
    IC = core.compile_class(os.path.join(REF, 'interactions.py'), 'InteractionComplex', ['_extend_atom_properties'], ns)
    holder = IC.__new__(IC)
    holder.s_atoms = flagged
    holder._extend_atom_properties()          # interactions.py:1985-1991, executed
    out['element_flags'] = [{'element': a.element, 'is_metal': bool(a.is_metal), 'is_halogen': bool(a.is_halogen)} for a in flagged]
    json.dump(out, open(os.path.join(OUT, 'typing.json'), 'w'), indent=0)
    print({k: len(v['atoms']) for k, v in out.items() if isinstance(v, dict)})


if __name__ == '__main__':
    main()
