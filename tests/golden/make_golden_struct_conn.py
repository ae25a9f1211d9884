#!/usr/bin/env python3
"""Fixture for the `_struct_conn` part of the mmCIF reader: tests/golden/struct_conn.json.
Build container only (needs /root/reference).

Executes ``parse_struct_conn_bonds``, ``__process_struct_conn`` and ``__add_bond_to_openbabel``
(protein_reader.py:139-212) — AST-extracted where they lie, nothing copied — on holders: gemmi's block as a dict of
categories (item -> list of str), OpenBabel's molecule as adjacency lists behind GetAtomById / NewBond / OBAtomAtomIter.
Stored: the two categories, the mmCIF text of the same content, the bonds the molecule had before and the adjacency
lists the reference left behind.

    python tests/golden/make_golden_struct_conn.py
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get('ARP_GOLDEN_OUT', HERE)      # (tests/test_fixture_freshness.py regenerates into a scratch directory)
sys.path.insert(0, HERE)
from make_golden_core import REF, compile_functions   # noqa: E402
from make_golden_reader import cif_text                # noqa: E402


class OBAtom:
    def __init__(self, oid):
        self.oid, self.nbrs = oid, []

    def GetId(self):
        return self.oid


class OBBond:
    def __init__(self, mol):
        self.mol, self.a, self.b, self.order = mol, None, None, None

    def SetBegin(self, a):
        self.a = a

    def SetEnd(self, b):
        self.b = b

    def SetBondOrder(self, o):          # the last call of __add_bond_to_openbabel: the bond is complete here
        self.order = o
        self.a.nbrs.append(self.b)
        self.b.nbrs.append(self.a)
        self.mol.new_bonds.append((self.a.oid, self.b.oid, o))


class OBMol:
    def __init__(self, ids, bonds):
        self.atoms = {i: OBAtom(i) for i in ids}
        self.new_bonds = []
        for a, b in bonds:
            self.atoms[a].nbrs.append(self.atoms[b])
            self.atoms[b].nbrs.append(self.atoms[a])

    def GetAtomById(self, i):
        return self.atoms[i]

    def NewBond(self):
        return OBBond(self)


class Block:
    def __init__(self, cats):
        self.cats = cats

    def get_mmcif_category_names(self):
        return list(self.cats)

    def get_mmcif_category(self, name):
        return self.cats[name]


def case(seed):
    rs = np.random.RandomState(seed)
    # atom_site: two chains; residue numbers reused across chains; an atom name that occurs twice in a residue (alt locs);
    # insertion-coded residues sharing auth_seq_id (the reference takes the first row that matches)
    rows = []

    def add(chain, seq, comp, name, alt=None, ins=None, el='C'):
        rows.append(dict(group_PDB='ATOM', id=str(len(rows) + 1), type_symbol=el, label_atom_id=name, label_alt_id=alt, label_comp_id=comp,
                         label_asym_id=chain, label_entity_id='1', label_seq_id=str(seq), pdbx_PDB_ins_code=ins,
                         Cartn_x='%.3f' % rs.uniform(0, 30), Cartn_y='%.3f' % rs.uniform(0, 30), Cartn_z='%.3f' % rs.uniform(0, 30),
                         occupancy='1.00', B_iso_or_equiv='10.00', pdbx_formal_charge=None, auth_seq_id=str(seq), auth_asym_id=chain,
                         pdbx_PDB_model_num='1'))

    for chain in 'AB':
        for seq in (10, 11, 12):
            for name in ('N', 'CA', 'C', 'O', 'SG'):
                add(chain, seq, 'CYS', name, el=name[0])
    add('A', 12, 'CYS', 'CB', alt='A'); add('A', 12, 'CYS', 'CB', alt='B')
    add('A', 13, 'SER', 'OG', ins='A', el='O'); add('A', 13, 'THR', 'OG', ins='B', el='O')
    add('A', 301, 'HEM', 'FE', el='FE'); add('A', 301, 'HEM', "C1'"); add('A', 302, 'ZN', 'ZN', el='ZN')
    add('B', 401, 'HOH', 'O', el='O')
    atom_site = {k: [r[k] for r in rows] for k in rows[0]}
    conn = []

    def link(kind, a, b):
        conn.append(dict(id=f'{kind}{len(conn) + 1}', conn_type_id=kind, ptnr1_auth_asym_id=a[0], ptnr1_auth_seq_id=str(a[1]), ptnr1_label_atom_id=a[2],
                         ptnr2_auth_asym_id=b[0], ptnr2_auth_seq_id=str(b[1]), ptnr2_label_atom_id=b[2], pdbx_dist_value='2.03'))

    link('disulf', ('A', 10, 'SG'), ('B', 10, 'SG'))
    link('disulf', ('A', 11, 'SG'), ('A', 12, 'SG'))
    link('covale', ('A', 12, 'CB'), ('A', 301, "C1'"))        # alt locs: the first CB row
    link('metalc', ('A', 301, 'FE'), ('A', 11, 'SG'))
    link('metalc', ('A', 302, 'ZN'), ('B', 401, 'O'))
    link('hydrog', ('A', 13, 'OG'), ('B', 12, 'O'))           # insertion codes A / B share auth_seq_id 13: the first row
    link('covale', ('A', 10, 'C'), ('A', 11, 'N'))            # a bond the molecule has already (both directions below)
    link('covale', ('A', 11, 'N'), ('A', 10, 'C'))
    link('covale', ('A', 99, 'XX'), ('A', 10, 'N'))           # partner 1 not in the table
    link('covale', ('B', 10, 'CA'), ('C', 10, 'CA'))          # partner 2 not in the table
    link('disulf', ('B', 10, 'SG'), ('A', 10, 'SG'))          # the first link again, reversed
    struct_conn = {k: [r[k] for r in conn] for k in conn[0]}
    ids = [int(x) for x in atom_site['id']]
    by = {(c, s, n): int(i) for c, s, n, i in reversed(list(zip(atom_site['auth_asym_id'], atom_site['auth_seq_id'], atom_site['label_atom_id'], atom_site['id'])))}
    existing = [(by[('A', '10', 'C')], by[('A', '11', 'N')])]
    return atom_site, struct_conn, ids, existing


def main():
    import logging
    logging.disable(logging.CRITICAL)
    out = []
    for seed in (1, 2):
        atom_site, struct_conn, ids, existing = case(seed)
        mol = OBMol(ids, existing)
        ob = types.SimpleNamespace(OBAtomAtomIter=lambda a: iter(list(a.nbrs)))
        ns = {'ob': ob}
        F = compile_functions(os.path.join(REF, 'protein_reader.py'),
                              ['parse_struct_conn_bonds', '__process_struct_conn', '__add_bond_to_openbabel'], ns)
        block = Block({'_atom_site.': atom_site, '_struct_conn.': struct_conn})
        F['parse_struct_conn_bonds'](mol, block)                       # protein_reader.py:139-151, executed
        text = cif_text('conn%d' % seed, [('_atom_site.', atom_site), ('_struct_conn.', struct_conn)], singles=[('_entry.id', 'CONN')])
        out.append({'atom_site': atom_site, 'struct_conn': struct_conn, 'text': text, 'existing_bonds': existing,
                    'new_bonds': [list(b) for b in mol.new_bonds],
                    'neighbours': {str(i): [n.oid for n in mol.atoms[i].nbrs] for i in ids if mol.atoms[i].nbrs}})
        assert len(mol.new_bonds) == 6, mol.new_bonds
    json.dump(out, open(os.path.join(OUT, 'struct_conn.json'), 'w'), indent=0)
    print([c['new_bonds'] for c in out])


if __name__ == '__main__':
    main()
