#!/usr/bin/env python3
"""Fixtures for the mmCIF reader: tests/golden/reader.json.  Build container only.

Executes, from the sources where they lie under /root/reference (AST-extracted, nothing copied):

  protein_reader.py  _parse_atom_site_biopython, _init_biopython_atom, _get_hetero_flag, _get_res_id, _get_b_factor,
                     _get_ins_code on a RECORDING StructureBuilder (every init_* call with its arguments);
                     get_component_types on a holder for gemmi's block (category names / category as a dict of columns)
  interactions.py    _handle_chains_residues_and_breaks on holder residues and a PPBuilder stand-in that returns given
                     polypeptides

The inputs are `_atom_site` / `_chem_comp` categories made up here as gemmi would deliver them (dict item -> list of str,
None for '?', False for '.'), together with the mmCIF TEXT of the same content — written with the quoting a real file
uses — which is what the native reader parses.  Expected: the recorded builder calls, the component types, the
polypeptide flags and links.

    python tests/golden/make_golden_reader.py
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get('ARP_GOLDEN_OUT', HERE)      # (tests/test_fixture_freshness.py regenerates into a scratch directory)
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from make_golden_core import REF, compile_class, compile_functions, load_by_path  # noqa: E402


# ---- made-up categories ------------------------------------------------------------------------------------------
def cif_value(v):
    if v is None:
        return '?'
    if v is False:
        return '.'
    s = str(v)
    if s == '' or any(c.isspace() for c in s) or s[0] in '_#$\'";[]' or s in ('?', '.') or s.lower().startswith(('data_', 'loop_', 'save_')):
        if "'" not in s:
            return "'" + s + "'"
        return '"' + s + '"'
    if "'" in s or '"' in s:
        return '"' + s + '"' if '"' not in s else "'" + s + "'"
    return s


def cif_text(block, cats, singles=()):
    out = ['# made-up content', 'data_' + block, '#']
    for name, value in singles:
        out.append(f'{name}   {cif_value(value)}')
    for cat, cols in cats:
        out.append('#')
        out.append('loop_')
        items = list(cols)
        out += [cat + it for it in items]
        for r in range(len(cols[items[0]])):
            out.append(' '.join(cif_value(cols[it][r]) for it in items))
    out.append('#')
    return '\n'.join(out) + '\n'


def peptide_atom_site(rng, variant):
    """A small two-chain structure: a peptide with a chain break, an insertion code, alternative locations, a modified
    residue, a ligand with awkward atom names, ions and waters."""
    rows = []
    serial = [0]

    def add(group, el, name, alt, comp, chain, seq, ins, xyz, occ=1.0, model=1, label_seq=None, charge=None):
        serial[0] += 1
        rows.append(dict(group_PDB=group, id=str(serial[0]), type_symbol=el, label_atom_id=name, label_alt_id=alt, label_comp_id=comp,
                         label_asym_id=chain, label_entity_id='1', label_seq_id=(str(label_seq) if label_seq is not None else False),
                         pdbx_PDB_ins_code=ins, Cartn_x='%.3f' % xyz[0], Cartn_y='%.3f' % xyz[1], Cartn_z='%.3f' % xyz[2],
                         occupancy='%.2f' % occ, B_iso_or_equiv='%.2f' % (10 + serial[0] % 7), pdbx_formal_charge=charge,
                         auth_seq_id=str(seq), auth_asym_id=chain, pdbx_PDB_model_num=str(model)))

    def residue(comp, chain, seq, origin, ins=None, group='ATOM', alt_ca=False, break_before=False):
        o = np.asarray(origin, float)
        add(group, 'N', 'N', None, comp, chain, seq, ins, o)
        if alt_ca:
            add(group, 'C', 'CA', 'A', comp, chain, seq, ins, o + [1.46, 0, 0], occ=0.4)
            add(group, 'C', 'CA', 'B', comp, chain, seq, ins, o + [1.40, 0.3, 0], occ=0.6)
        else:
            add(group, 'C', 'CA', None, comp, chain, seq, ins, o + [1.46, 0, 0])
        add(group, 'C', 'C', None, comp, chain, seq, ins, o + [2.0, 1.4, 0])
        add(group, 'O', 'O', None, comp, chain, seq, ins, o + [1.4, 2.4, 0])
        if comp != 'GLY':
            add(group, 'C', 'CB', None, comp, chain, seq, ins, o + [2.0, -1.2, 0.8])
        return o + [3.3, 1.5, 0]          # where the next N goes (1.33 A from C)

    names = ['ALA', 'GLY', 'SER', 'PHE', 'HIS', 'ASN', 'GLN', 'MET', 'TRP', 'LYS']
    nxt = np.zeros(3)
    seq = 10
    for k in range(6):
        nxt = residue(names[(k + variant) % len(names)], 'A', seq, nxt, alt_ca=(k == 2))
        seq += 1
    nxt = residue('MSE', 'A', seq, nxt, group='HETATM')                          # modified residue inside the chain
    seq += 1
    nxt = residue(names[(7 + variant) % len(names)], 'A', seq, nxt, ins='A')    # insertion code, same number as ...
    nxt = residue(names[(8 + variant) % len(names)], 'A', seq, nxt, ins='B')    # ... this one
    seq += 4
    nxt = residue('LEU', 'A', seq, nxt + [8.0, 0, 0])                           # chain break: C-N = 9.3 A
    nxt = residue('VAL', 'A', seq + 1, nxt)
    o = np.array([0.0, 12.0, 5.0])
    for k in range(4):                                                          # second chain
        o = residue(names[(k + 3 + variant) % len(names)], 'B', 1 + k, o)
    lig = np.array([5.0, 5.0, 8.0])
    for k, (el, nm) in enumerate((('C', "C1'"), ('O', "O5'"), ('N', 'N 1'), ('P', 'PA'), ('C', 'C"2'), ('FE', 'FE'))):
        add('HETATM', el, nm, None, 'LIG', 'A', 301, None, lig + [1.4 * k, 0.2 * k, 0], charge=('2' if el == 'FE' else None))
    add('HETATM', 'ZN', 'ZN', None, 'ZN', 'A', 302, None, [9.0, 9.0, 9.0], charge='2')
    add('HETATM', 'CA', 'CA', None, 'CA', 'B', 303, None, [12.0, 9.0, 9.0])
    for k in range(5):
        add('HETATM', 'O', 'O', None, ('HOH' if k % 2 == 0 else 'WAT'), ('A' if k < 3 else 'B'), 401 + k, None, rng.random(3) * 20)
    if variant == 1:          # a second model: the reference drops it, the call sequence still shows it
        add('ATOM', 'N', 'N', None, 'ALA', 'A', 10, None, [0.5, 0.5, 0.5], model=2)
        add('ATOM', 'C', 'CA', None, 'ALA', 'A', 10, None, [1.9, 0.5, 0.5], model=2)
    cols = {k: [r[k] for r in rows] for k in rows[0]}
    if variant == 2:          # the PDBe flavour: pdbe_label_seq_id wins over auth_seq_id (P:16-21), no B factors (P:24-27)
        cols['pdbe_label_seq_id'] = [str(1000 + int(s)) for s in cols['auth_seq_id']]
        del cols['B_iso_or_equiv']
        del cols['pdbx_formal_charge']
    return cols


def aromatic_atom_site():
    """A two-chain peptide with complete aromatic and amide side chains and explicit hydrogens: what the ring / amide loops of
    run_arpeggio (I:938-1382) need from a FILE — PHE / TYR / TRP / HIS rings stacked at 3.6 A (parallel, tilted and on edge),
    ASN / GLN side-chain amides, a methionine sulphur and a lysine nitrogen over rings.  Idealised, not energy-minimised:
    consecutive residues are 3.6 A apart along the chain (C - N = 1.33 A), every ring a regular polygon of radius 1.39 A."""
    rows = []
    serial = [0]

    def add(el, name, comp, chain, seq, xyz):
        serial[0] += 1
        rows.append(dict(group_PDB='ATOM', id=str(serial[0]), type_symbol=el, label_atom_id=name, label_alt_id=None, label_comp_id=comp,
                         label_asym_id=chain, label_entity_id='1', label_seq_id=str(seq), pdbx_PDB_ins_code=None,
                         Cartn_x='%.3f' % xyz[0], Cartn_y='%.3f' % xyz[1], Cartn_z='%.3f' % xyz[2], occupancy='1.00',
                         B_iso_or_equiv='%.2f' % (10 + serial[0] % 7), pdbx_formal_charge=None, auth_seq_id=str(seq), auth_asym_id=chain,
                         pdbx_PDB_model_num='1'))

    def polygon(centre, normal, start, names, elements, comp, chain, seq, hydrogens=()):
        n = np.asarray(normal, float) / np.linalg.norm(normal)
        u = np.cross(n, [0.0, 0.0, 1.0])
        if np.linalg.norm(u) < 1e-6:
            u = np.cross(n, [0.0, 1.0, 0.0])
        u /= np.linalg.norm(u)
        v = np.cross(n, u)
        k = len(names)
        pts = {}
        for j, (nm, el) in enumerate(zip(names, elements)):
            t = start + 2 * np.pi * j / k
            pts[nm] = np.asarray(centre) + 1.39 * (np.cos(t) * u + np.sin(t) * v)
            add(el, nm, comp, chain, seq, pts[nm])
        for hn, on in hydrogens:                      # a hydrogen 1.08 A outwards of its ring atom
            d = pts[on] - np.asarray(centre)
            add('H', hn, comp, chain, seq, pts[on] + 1.08 * d / np.linalg.norm(d))
        return pts

    def residue(comp, chain, seq, o, tilt=0.0):
        o = np.asarray(o, float)
        add('N', 'N', comp, chain, seq, o)
        add('C', 'CA', comp, chain, seq, o + [1.46, 0, 0])
        add('C', 'C', comp, chain, seq, o + [2.0, 1.4, 0])
        add('O', 'O', comp, chain, seq, o + [1.4, 2.4, 0])
        add('H', 'H', comp, chain, seq, o + [-0.5, -0.85, 0.0])
        cb = o + [2.0, -1.2, 0.8]
        if comp != 'GLY':
            add('C', 'CB', comp, chain, seq, cb)
        c = o + [1.8, -4.2, 1.2]                                           # centre of the side chain's ring, if it has one
        nrm = [np.cos(tilt), np.sin(tilt), 0.0]                            # tilt = 0: normal along the chain
        if comp in ('PHE', 'TYR'):
            pts = polygon(c, nrm, 0.3, ['CG', 'CD1', 'CE1', 'CZ', 'CE2', 'CD2'], 'CCCCCC', comp, chain, seq,
                          hydrogens=(('HD1', 'CD1'), ('HE1', 'CE1'), ('HE2', 'CE2'), ('HD2', 'CD2')) + ((('HZ', 'CZ'),) if comp == 'PHE' else ()))
            if comp == 'TYR':
                d = pts['CZ'] - c
                oh = pts['CZ'] + 1.36 * d / np.linalg.norm(d)
                add('O', 'OH', comp, chain, seq, oh)
                add('H', 'HH', comp, chain, seq, oh + 0.96 * d / np.linalg.norm(d))
        elif comp == 'HIS':
            polygon(c, nrm, 0.1, ['CG', 'ND1', 'CE1', 'NE2', 'CD2'], 'CNCNC', comp, chain, seq, hydrogens=(('HD1', 'ND1'), ('HE1', 'CE1'), ('HD2', 'CD2')))
        elif comp == 'TRP':
            polygon(c, nrm, 0.2, ['CG', 'CD1', 'NE1', 'CE2', 'CD2'], 'CCNCC', comp, chain, seq, hydrogens=(('HD1', 'CD1'), ('HE1', 'NE1')))
            n_ = np.asarray(nrm)
            u = np.cross(n_, [0.0, 0.0, 1.0]); u /= np.linalg.norm(u)
            polygon(c + 2.3 * u, nrm, 0.5, ['CZ2', 'CH2', 'CZ3', 'CE3'], 'CCCC', comp, chain, seq, hydrogens=(('HZ2', 'CZ2'), ('HH2', 'CH2')))
        elif comp in ('ASN', 'GLN'):
            names = {'ASN': ('CG', 'OD1', 'ND2'), 'GLN': ('CD', 'OE1', 'NE2')}[comp]
            if comp == 'GLN':
                add('C', 'CG', comp, chain, seq, o + [1.9, -2.6, 1.0])
            add('C', names[0], comp, chain, seq, c + [0.0, 0.3, 0.0])
            add('O', names[1], comp, chain, seq, c + [0.0, -0.4, 1.0])
            add('N', names[2], comp, chain, seq, c + [0.0, -0.5, -1.1])
            add('H', 'H' + names[2][1:] + '1', comp, chain, seq, c + [0.0, -1.5, -1.2])
        elif comp == 'MET':
            add('C', 'CG', comp, chain, seq, o + [1.9, -2.6, 1.0])
            add('S', 'SD', comp, chain, seq, c)
            add('C', 'CE', comp, chain, seq, c + [0.3, -1.7, 0.4])
        elif comp == 'LYS':
            add('C', 'CG', comp, chain, seq, o + [1.9, -2.6, 1.0])
            add('C', 'CD', comp, chain, seq, o + [1.9, -3.4, 2.2])
            add('C', 'CE', comp, chain, seq, c + [0.0, 1.0, 0.6])
            add('N', 'NZ', comp, chain, seq, c)
            for j in range(3):
                add('H', f'HZ{j + 1}', comp, chain, seq, c + [0.95 * np.cos(2.1 * j), -0.3, 0.95 * np.sin(2.1 * j)])
        return o + [3.3, 1.5, 0]

    nxt = np.zeros(3)
    for k, (comp, tilt) in enumerate((('PHE', 0.0), ('TYR', 0.0), ('TRP', 0.6), ('HIS', 1.3), ('ASN', 0.0), ('PHE', 0.3), ('GLN', 0.0), ('MET', 0.0),
                                      ('PHE', 0.0), ('LYS', 0.0), ('TYR', 0.9), ('ALA', 0.0))):
        nxt = residue(comp, 'A', 20 + k, nxt, tilt)
    o = np.array([0.6, -8.6, 2.4])                                          # second chain: its side chains face the first one's
    for k, (comp, tilt) in enumerate((('TYR', 0.2), ('ASN', 0.0), ('GLN', 0.0), ('TRP', 0.0), ('HIS', 0.5), ('GLY', 0.0))):
        nxt_b = residue(comp, 'B', 5 + k, [o[0] + 3.3 * k, o[1] + 1.5 * k - 0.0, o[2]], tilt)
    return {k: [r[k] for r in rows] for k in rows[0]}


def chem_comp(variant):
    rows = [('ALA', 'L-peptide linking', 'ALANINE'), ('GLY', 'peptide linking', 'GLYCINE'), ('MSE', 'L-PEPTIDE LINKING', 'SELENOMETHIONINE'),
            ('HOH', 'non-polymer', 'WATER'), ('WAT', 'NON-POLYMER', 'water'), ('LIG', 'non-polymer', 'SOME LIGAND'), ('ZN', 'non-polymer', 'ZINC ION'),
            ('DA', 'DNA linking', "2'-DEOXYADENOSINE-5'-MONOPHOSPHATE"), ('U', 'RNA linking', "URIDINE-5'-MONOPHOSPHATE"),
            ('NAG', 'D-saccharide, beta linking', '2-acetamido-2-deoxy-beta-D-glucopyranose'), ('XYZ', 'other', 'SOMETHING'),
            ('UNK', None, 'UNKNOWN'), ('UNL', False, 'UNKNOWN LIGAND'), ('PEP', 'peptide-like', 'A PEPTIDE-LIKE THING')]
    if variant:
        rows = rows[::-1]
    return {'id': [r[0] for r in rows], 'type': [r[1] for r in rows], 'name': [r[2] for r in rows],
            'formula': [('C3 H7 N O2' if k % 2 else None) for k in range(len(rows))]}


# ---- recording builder and holders ---------------------------------------------------------------------------------
class PDBConstructionException(Exception):
    pass


class Recorder:
    def __init__(self):
        self.calls = []

    def init_model(self, n):
        self.calls.append(['model', n])

    def init_seg(self, s):
        self.calls.append(['seg', s])

    def init_chain(self, c):
        self.calls.append(['chain', c])

    def init_residue(self, name, field, seq, icode):
        self.calls.append(['residue', name, field, seq, icode])

    def init_atom(self, name, coord, b, occ, alt, fullname, serial, element):
        assert coord.dtype == np.float32
        self.calls.append(['atom', name, [float(x) for x in coord], b, occ, alt, fullname, serial, element])


def main():
    rng = np.random.default_rng(8)
    config = load_by_path('ref_config', os.path.join(REF, 'config.py'))
    PDB = types.SimpleNamespace(PDBExceptions=types.SimpleNamespace(PDBConstructionException=PDBConstructionException))
    state = {}

    class _Block:
        def get_mmcif_category_names(self):
            return list(state['cats'])

        def get_mmcif_category(self, name):
            return state['cats'][name]

    gemmi = types.SimpleNamespace(cif=types.SimpleNamespace(read=lambda path: types.SimpleNamespace(sole_block=lambda: _Block())))
    fake_os = types.SimpleNamespace(path=types.SimpleNamespace(isfile=lambda p: True, basename=os.path.basename))
    ns = {'numpy': np, 'PDB': PDB, 'gemmi': gemmi, 'os': fake_os, 'config': config}
    P = compile_functions(os.path.join(REF, 'protein_reader.py'),
                          ['_get_res_id', '_get_b_factor', '_get_ins_code', '_format_formal_charge', '_get_hetero_flag', '_init_biopython_atom',
                           '_parse_atom_site_biopython', 'get_component_types'], ns)
    cases = []
    for variant in range(3):
        cols = peptide_atom_site(rng, variant)
        rec = Recorder()
        P['_parse_atom_site_biopython'](cols, rec)
        cc = chem_comp(variant)
        state['cats'] = {'_chem_comp.': cc}
        comp = P['get_component_types']('made_up.cif')
        text = cif_text(f'CASE{variant}', [('_atom_site.', cols), ('_chem_comp.', cc)], singles=[('_entry.id', f'CASE{variant}')])
        cases.append(dict(name=f'peptide{variant}', text=text, atom_site=cols, chem_comp=cc, builder_calls=rec.calls, component_types=comp))
    # a file without _chem_comp: ValueError('Missing _chem_comp. category in mmcif')
    state['cats'] = {'_atom_site.': {}}
    try:
        P['get_component_types']('x.cif')
        raise SystemExit('expected ValueError')
    except ValueError as e:
        missing = str(e)
    hetero = [[f, r, P['_get_hetero_flag'](f, r)] for f in ('ATOM', 'HETATM', 'hetatm', '') for r in ('HOH', 'WAT', 'ALA', 'hoh')]

    # ---- _handle_chains_residues_and_breaks on given polypeptides
    import collections
    import logging
    import operator
    from functools import reduce

    class Chain:
        def __init__(self, cid):
            self.id = cid

    class Res:
        def __init__(self, idx, chain):
            self.idx, self.chain = idx, chain

        def get_parent(self):
            return self.chain

        def __hash__(self):
            return self.idx

    class PPB:
        pps = []

        def build_peptides(self, structure, aa_only=True):
            assert aa_only is False
            return PPB.pps

    ins = {'collections': collections, 'logging': logging, 'operator': operator, 'reduce': reduce, 'PPBuilder': PPB}
    IC = compile_class(os.path.join(REF, 'interactions.py'), 'InteractionComplex', ['_handle_chains_residues_and_breaks'], ins)
    logging.disable(logging.CRITICAL)
    book = []
    for layout in ([[0, 1, 2, 3], [6, 7]], [[0, 1], [3, 4, 5], [8, 9, 10, 11]], []):
        ca, cb = Chain('A'), Chain('B')
        residues = [Res(k, ca if k < 8 else cb) for k in range(13)]
        PPB.pps = [[residues[k] for k in pp] for pp in layout]
        obj = IC.__new__(IC)
        obj.biopython_str = types.SimpleNamespace(get_chains=lambda: iter([ca, cb]))
        obj._handle_chains_residues_and_breaks()
        book.append(dict(polypeptides=layout,
                         residues=[dict(is_polypeptide=getattr(r, 'is_polypeptide', False), has_links=hasattr(r, 'prev_residue') and hasattr(r, 'next_residue'),
                                        prev=(r.prev_residue.idx if getattr(r, 'prev_residue', None) is not None else -1),
                                        next=(r.next_residue.idx if getattr(r, 'next_residue', None) is not None else -1),
                                        is_terminal=getattr(r, 'is_terminal', None), is_chain_break=getattr(r, 'is_chain_break', None))
                                   for r in residues],
                         polypeptide_residues=sorted(r.idx for r in obj.polypeptide_residues)))
    out = dict(cases=cases, missing_chem_comp=missing, hetero_flag=hetero, bookkeeping=book)
    json.dump(out, open(os.path.join(OUT, 'reader.json'), 'w'), indent=1)
    print('reader.json:', len(cases), 'cases,', sum(len(c['builder_calls']) for c in cases), 'builder calls')


if __name__ == '__main__':
    main()
