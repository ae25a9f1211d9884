#!/usr/bin/env python3
"""Golden vectors for the CORE of the hot path, produced by EXECUTING the reference's own methods.

Run in the build container only (needs /root/reference; the GPU box never sees it):

    python tests/golden/make_golden_core.py

What is executed (function bodies compiled with ``ast`` from the files where they lie under
/root/reference, assembled into a class called InteractionComplex so that name mangling of the
``__private`` methods works as in the original):

  interactions.py  run_arpeggio, _make_selection, _calculate_atom_contacts, __get_contact_type,
                   _calculate_ring_contacts, __calculate_atom_plane_contacts, __calculate_plane_plane_contacts,
                   _calculate_group_contacts, __calculate_group_group_contacts, __calculate_group_plane_contacts,
                   get_contacts, _prepare_plane_plane_contact_for_export, _prepare_atom_plane_contact_for_export,
                   _initialize_atom_sift, _initialize_residue_sift, _calc_residue_sifts, structure_checks,
                   write_contacts, __write_contact_file, write_atom_types, write_atom_sifts, __write_atom_sifts,
                   write_residue_sifts, write_polar_matching
  utils.py         get_angle, group_angle, group_group_angle, is_hbond, is_weak_hbond, is_halogen_weak_hbond,
                   is_xbond, get_single_bond_neighbour, update_atom_sift / _fsift / _integer_sift, selection_parser,
                   is_digit, make_pymol_json, make_pymol_string, get_residue_name

The third-party objects those bodies touch (BioPython Atom / Residue / Chain, the OpenBabel atom / bond iterators,
Bio.PDB.NeighborSearch) are absent from this image.  They are replaced by DATA HOLDERS that carry no arithmetic of
the path: stored attributes, stored neighbour lists, stored bond orders.  The one exception is the NeighborSearch
holder, whose ``search_all`` / ``search`` enumerate pairs by brute force with the KD-tree's membership test as
recalled from BioPython's C source (float64, ``dx*dx+dy*dy+dz*dz <= r*r``): that third-party semantic stays
"recalled", everything the reference itself computes per pair is executed reference code.

Delivery order / orientation of ``search_all`` is an artefact of the KD-tree and of Python's hash seed in the
reference; the holder offers three modes: 'canonical' (pairs sorted by (i, j), bgn = lower packed index — the order
this project defines results on), 'reversed' (bgn = higher index) and 'random' (seeded shuffle of order and
orientation).  The HIP path is compared with the canonical runs, the C oracle with all of them.

Only inputs (packed arrays) and outputs are written; no reference source text is stored anywhere in this repository.
"""
import ast
import collections
import csv
import gzip
import hashlib
import importlib.util
import io
import json
import logging
import operator
import os
import sys
import tempfile
import time
import types
from functools import reduce

import numpy as np

REF = '/root/reference/arpeggio/core'
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from arpeggio_amd import synth                      # noqa: E402
from arpeggio_amd.core import config as pconfig     # noqa: E402  (bit positions of the packed arrays only)
from arpeggio_amd.core.packed import PackedComplex  # noqa: E402

SIFT_SLOTS = ('', '_inter_only', '_intra_only', '_water_only')
COUNTERS = ('actual_hbonds', 'actual_hbonds_intra_only', 'actual_hbonds_inter_only', 'actual_hbonds_water_only',
            'actual_polars', 'actual_polars_intra_only', 'actual_polars_inter_only', 'actual_polars_water_only')


# ------------------------------------------------------------------------------------------------ code extraction
def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def function_nodes(path, names, in_class=None):
    tree = ast.parse(open(path).read(), filename=path)
    body = tree.body
    if in_class:
        body = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == in_class).body
    found = {n.name: n for n in body if isinstance(n, ast.FunctionDef) and n.name in names}
    missing = set(names) - set(found)
    if missing:
        raise RuntimeError(f'not found in {path}: {missing}')
    return [found[n] for n in names]


def compile_functions(path, names, namespace):
    for node in function_nodes(path, names):
        exec(compile(ast.Module(body=[node], type_ignores=[]), path, 'exec'), namespace)
    return {n: namespace[n] for n in names}


def compile_class(path, class_name, names, namespace):
    """The named methods of `class_name` as a class of the same name (private-name mangling preserved)."""
    cls = ast.ClassDef(name=class_name, bases=[], keywords=[], body=function_nodes(path, names, in_class=class_name),
                       decorator_list=[])
    if sys.version_info >= (3, 12):
        cls.type_params = []
    mod = ast.Module(body=[cls], type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, path, 'exec'), namespace)
    return namespace[class_name]


def compile_assignments(path, names, namespace):
    tree = ast.parse(open(path).read(), filename=path)
    for node in tree.body:
        if isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id in names for t in node.targets):
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, 'exec'), namespace)


# ------------------------------------------------------------------------------------------------ data holders
class Chain:
    def __init__(self, cid):
        self.id = cid


class Residue:
    """BioPython Residue stand-in: attributes only."""

    def __init__(self, idx, chain, resname, het, seq, icode):
        self.idx, self.chain, self.resname, self.id, self.child_list = idx, chain, resname, (het, seq, icode), []

    def get_parent(self):
        return self.chain

    def get_resname(self):
        return self.resname

    def __hash__(self):
        return self.idx

    def __repr__(self):
        return f'<Residue {self.idx} {self.resname}>'


class Atom:
    """BioPython Atom stand-in: attributes only (hash / equality by identity like an object in a set)."""

    def __init__(self, idx, res, name, element, coord, serial, water):
        self.idx, self.res, self.name, self.element, self.coord, self.serial_number = idx, res, name, element, coord, serial
        self.water = water
        res.child_list.append(self)

    def get_parent(self):
        return self.res

    def get_id(self):
        return self.name

    def get_full_id(self):
        r = self.res
        return ('s', 0, r.chain.id, ('W' if self.water else r.id[0], r.id[1], r.id[2]), (self.name, ' '))

    def __hash__(self):
        return self.idx

    def __repr__(self):
        return f'<Atom {self.idx} {self.name}>'


class OBAtom:
    def __init__(self, oid, atomic_num):
        self.oid, self.atomic_num, self.nbrs, self.bonds = oid, atomic_num, [], []

    def GetId(self):
        return self.oid

    def GetAtomicNum(self):
        return self.atomic_num


class OBBond:
    def __init__(self, a, b, order, aromatic):
        self.a, self.b, self.order, self.aromatic = a, b, order, aromatic

    def GetBondOrder(self):
        return self.order

    def IsAromatic(self):
        return self.aromatic

    def GetNbrAtom(self, x):
        return self.b if x is self.a else self.a


class OBMol:
    def __init__(self, atoms):
        self.atoms = atoms

    def GetAtomById(self, oid):
        return self.atoms[oid]


ob = types.SimpleNamespace(OBAtomAtomIter=lambda a: iter(a.nbrs), OBAtomBondIter=lambda a: iter(a.bonds))


class NeighborSearch:
    """Holder for Bio.PDB.NeighborSearch: brute-force enumeration with the KD-tree's inclusive float64 test (recalled)."""
    mode = 'canonical'
    seed = 0

    def __init__(self, atom_list, bucket_size=10):
        self.atoms = sorted(atom_list, key=lambda a: a.idx)
        self.xyz = np.array([a.coord for a in self.atoms], np.float64).reshape(-1, 3)

    def search_all(self, radius, level='A'):
        x, n, out = self.xyz, len(self.atoms), []
        r2 = float(radius) * float(radius)
        for i in range(n - 1):
            d = x[i + 1:] - x[i]
            d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
            for j in np.nonzero(d2 <= r2)[0].tolist():
                out.append((self.atoms[i], self.atoms[i + 1 + j]))
        if self.mode == 'reversed':
            out = [(b, a) for a, b in out]
        elif self.mode == 'random':
            rng = np.random.default_rng(NeighborSearch.seed)
            flip = rng.random(len(out)) < 0.5
            out = [(b, a) if f else (a, b) for (a, b), f in zip(out, flip)]
            out = [out[k] for k in rng.permutation(len(out))]
        return out

    def search(self, center, radius, level='A'):
        c = np.asarray(center, np.float64)
        d = self.xyz - c
        d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
        hits = [self.atoms[k] for k in np.nonzero(d2 <= float(radius) * float(radius))[0].tolist()]
        if self.mode == 'reversed':
            hits = hits[::-1]
        elif self.mode == 'random':
            hits = [hits[k] for k in np.random.default_rng(NeighborSearch.seed + 1).permutation(len(hits))]
        return hits


class Structure:
    def __init__(self, residues):
        self.residues, self.rings, self.amides = residues, collections.OrderedDict(), collections.OrderedDict()

    def get_residues(self):
        return iter(self.residues)


# ------------------------------------------------------------------------------------------------ reference namespace
def reference_namespace():
    logging.disable(logging.CRITICAL)
    config = load_by_path('ref_config', os.path.join(REF, 'config.py'))
    exceptions = load_by_path('ref_exceptions', os.path.join(REF, 'exceptions.py'))
    uns = {'np': np, 'logging': logging, 'config': config, 'collections': collections, 'ob': ob,
           'Atom': Atom, 'Residue': Residue, 'SelectionError': exceptions.SelectionError}
    U = compile_functions(os.path.join(REF, 'utils.py'),
                          ['get_angle', 'group_angle', 'group_group_angle', 'is_hbond', 'is_weak_hbond',
                           'is_halogen_weak_hbond', 'is_xbond', 'get_single_bond_neighbour', 'update_atom_sift',
                           'update_atom_fsift', 'update_atom_integer_sift', 'is_digit', 'selection_parser',
                           'make_pymol_json', 'make_pymol_string', 'get_residue_name'], uns)
    utils = types.SimpleNamespace(**U)
    ins = {'np': np, 'logging': logging, 'config': config, 'collections': collections, 'utils': utils, 'ob': ob,
           'csv': csv, 'os': os, 'reduce': reduce, 'operator': operator, 'NeighborSearch': NeighborSearch,
           'AtomSerialError': exceptions.AtomSerialError}
    compile_assignments(os.path.join(REF, 'interactions.py'),
                        ['PlanePlaneContact', 'AtomPlaneContact', 'AtomAtomContact', 'Parameters'], ins)
    IC = compile_class(os.path.join(REF, 'interactions.py'), 'InteractionComplex',
                       ['run_arpeggio', '_make_selection', '_calculate_atom_contacts', '__get_contact_type',
                        '_calculate_ring_contacts', '__calculate_atom_plane_contacts', '__calculate_plane_plane_contacts',
                        '_calculate_group_contacts', '__calculate_group_group_contacts', '__calculate_group_plane_contacts',
                        'get_contacts', '_prepare_plane_plane_contact_for_export', '_prepare_atom_plane_contact_for_export',
                        '_initialize_atom_sift', '_initialize_residue_sift', '_calc_residue_sifts', 'structure_checks',
                        'write_contacts', '__write_contact_file', 'write_atom_types', 'write_atom_sifts',
                        '__write_atom_sifts', 'write_residue_sifts', 'write_polar_matching'], ins)
    return IC, utils, config, exceptions


# ------------------------------------------------------------------------------------------------ pack -> holders
_ATOMIC_NUMBER = {'H': 1, 'D': 1, 'C': 6, 'N': 7, 'O': 8, 'S': 16, 'CL': 17, 'FE': 26, 'ZN': 30}


def extras_for(pc):
    """Deterministic stand-ins for what _handle_hydrogens (I:1885-1921) reads from OpenBabel: atomic number by element,
    explicit valence 1..4 and formal charge -1..1 by index (only _initialize_atom_sift's potential counts read them)."""
    i = np.arange(pc.n_atoms)
    return {'atomic_number': np.array([_ATOMIC_NUMBER.get(e.strip().upper(), 6) for e in pc.element]),
            'bond_order': 1 + (i * 7) % 4, 'formal_charge': np.array([-1, 0, 0, 1, 0])[i % 5]}


def holders_from_pack(IC, pc, bond_order=None, bond_aromatic=None, extras=None):
    """An object shaped like the reference's InteractionComplex after initialize(), built from a PackedComplex.

    bond_order / bond_aromatic (aligned with pc.bond_idx): when given, every bond is ONE object seen from both ends
    (get_single_bond_neighbour then runs on real bond data); when absent, each atom's bond iterator delivers decoys
    (a double bond, an aromatic bond, a single bond to a hydrogen) followed by one single bond to pc.sb_nbr[i].
    extras: optional per-atom ints for _initialize_atom_sift (atomic_number, bond_order, formal_charge, num_hydrogens).
    """
    pc.ensure_labels()
    extras = extras_for(pc)
    n, nr = pc.n_atoms, pc.n_residues
    chains = {}
    residues = []
    for r in range(nr):
        cid = pc.res_chain[r]
        chains.setdefault(cid, Chain(cid))
        residues.append(Residue(r, chains[cid], pc.res_name[r], ' ', int(pc.res_seq[r]), pc.res_icode[r]))
    atoms = []
    names = pconfig.ATOM_TYPE_NAMES
    for i in range(n):
        f = int(pc.flags[i])
        a = Atom(i, residues[int(pc.res_id[i])], pc.atom_name[i], pc.element[i], pc.xyz[i].copy(), int(pc.serial[i]),
                 bool(f & pconfig.F_WATER))
        a.atom_types = {names[b] for b in range(12) if (int(pc.type_mask[i]) >> b) & 1}
        a.is_metal = bool(f & pconfig.F_METAL)
        a.is_halogen = bool(f & pconfig.F_HALOGEN)
        a.vdw_radius = float(pc.vdw[i])
        a.cov_radius = float(pc.cov[i])
        a.h_coords = [np.array(pc.h_xyz[k], np.float64) for k in range(pc.h_off[i], pc.h_off[i + 1])]
        ex = extras or {}
        a.atomic_number = int(ex['atomic_number'][i]) if 'atomic_number' in ex else 6
        a.bond_order = int(ex['bond_order'][i]) if 'bond_order' in ex else 4
        a.formal_charge = int(ex['formal_charge'][i]) if 'formal_charge' in ex else 0
        a.num_hydrogens = len(a.h_coords)
        atoms.append(a)
    # element / resname consistency of the flag bits the kernels read instead of strings
    for i, a in enumerate(atoms):
        f = int(pc.flags[i])
        assert (a.element.strip() == 'H') == bool(f & pconfig.F_HYDROGEN), (i, a.element)
        assert (a.element == 'C') == bool(f & pconfig.F_ELEM_C), (i, a.element)
        assert (a.element == 'S') == bool(f & pconfig.F_ELEM_S), (i, a.element)
        if a.element == 'S':      # the only place the residue name is read on the path (I:1023)
            assert (a.res.resname == 'MET') == bool(f & pconfig.F_RES_MET), (i, a.res.resname)
    # OpenBabel side
    obatoms = [OBAtom(1000 + i, 1 if (int(pc.flags[i]) & pconfig.F_HYDROGEN) else 6) for i in range(n)]
    bond_objs = {}
    for i in range(n):
        for k in range(pc.bond_off[i], pc.bond_off[i + 1]):
            j = int(pc.bond_idx[k])
            obatoms[i].nbrs.append(obatoms[j])
            if bond_order is not None:
                key = (min(i, j), max(i, j))
                if key not in bond_objs:
                    bond_objs[key] = OBBond(obatoms[key[0]], obatoms[key[1]], int(bond_order[k]), bool(bond_aromatic[k]))
                obatoms[i].bonds.append(bond_objs[key])
        if bond_order is None:
            me = obatoms[i]
            decoy_heavy, decoy_h = OBAtom(-1, 6), OBAtom(-2, 1)
            me.bonds = [OBBond(me, decoy_heavy, 2, False), OBBond(me, decoy_heavy, 1, True), OBBond(me, decoy_h, 1, False)]
            if pc.sb_nbr[i] >= 0:
                me.bonds.append(OBBond(me, obatoms[int(pc.sb_nbr[i])], 1, False))
    for r in range(nr):
        res = residues[r]
        fl = int(pc.res_flags[r])
        if fl & pconfig.R_HAS_SEQ:
            res.prev_residue = residues[pc.res_prev[r]] if pc.res_prev[r] >= 0 else None
            res.next_residue = residues[pc.res_next[r]] if pc.res_next[r] >= 0 else None
    st = Structure(residues)
    self_ = IC.__new__(IC)
    self_.id = pc.id
    self_.s_atoms = atoms
    self_.biopython_str = st
    self_.ob_mol = OBMol({a.oid: a for a in obatoms})
    self_.bio_to_ob = {a: obatoms[a.idx].oid for a in atoms}
    self_.ob_to_bio = {obatoms[a.idx].oid: a for a in atoms}
    self_.component_types = dict(pc.component_types)
    # what initialize() does with reference code: sift initialisation (I:1735-1883)
    self_._initialize_atom_sift()
    self_._initialize_residue_sift()
    self_.polypeptide_residues = set()
    for r in range(nr):
        if int(pc.res_flags[r]) & pconfig.R_POLYPEPTIDE:
            residues[r].is_polypeptide = True
            self_.polypeptide_residues.add(residues[r])
    for r in range(pc.n_rings):
        st.rings[r] = {'ring_id': r, 'center': np.array(pc.ring_center[r], np.float64), 'normal': np.array(pc.ring_normal[r], np.float64),
                       'atoms': [atoms[int(k)] for k in (pc.ring_atoms[r] if pc.ring_atoms else [])],
                       'residue': residues[int(pc.ring_res[r])] if pc.ring_res[r] >= 0 else None}
    for e in range(pc.n_amides):
        st.amides[e] = {'amide_id': e, 'center': np.array(pc.amide_center[e], np.float32), 'normal': np.array(pc.amide_normal[e], np.float32),
                        'atoms': [atoms[int(k)] for k in pc.amide_atoms[e] if k >= 0],
                        'residue': residues[int(pc.amide_res[e])]}
    return self_


# ------------------------------------------------------------------------------------------------ result dumping
def bits(lst):
    return int(sum(int(bool(b)) << k for k, b in enumerate(lst)))


def dump_run(self_, pc, config):
    ct_code = {nm: k for k, nm in enumerate(pconfig.CONTACT_TYPE_NAMES)}
    ac = self_.atom_contacts
    out = {
        'selection': np.array(sorted(a.idx for a in self_.selection), np.int32),
        'selection_plus': np.array(sorted(a.idx for a in self_.selection_plus), np.int32),
        'selection_plus_residues': np.array(sorted(r.idx for r in self_.selection_plus_residues), np.int32),
        'selection_ring_ids': np.array(sorted(self_.selection_ring_ids), np.int32),
        'selection_plus_ring_ids': np.array(sorted(self_.selection_plus_ring_ids), np.int32),
        'selection_amide_ids': np.array(sorted(self_.selection_amide_ids), np.int32),
        'selection_plus_amide_ids': np.array(sorted(self_.selection_plus_amide_ids), np.int32),
        'aa_bgn': np.array([c.bgn_atom.idx for c in ac], np.int32),
        'aa_end': np.array([c.end_atom.idx for c in ac], np.int32),
        'aa_sift': np.array([bits(c.sifts) for c in ac], np.uint16),
        'aa_ctype': np.array([ct_code[c.contact_type] for c in ac], np.uint8),
        'aa_dist': np.array([c.distance for c in ac], np.float32),
    }
    assert all(np.asarray(c.distance).dtype == np.float32 for c in ac)
    assert all(all(v in (0, 1) for v in c.sifts) for c in ac)
    n = pc.n_atoms
    A = self_.s_atoms
    out['atom_sift'] = np.array([[bits(getattr(a, 'sift' + s)) for s in SIFT_SLOTS] for a in A], np.uint16).reshape(n, 4)
    out['atom_fsift'] = np.array([[bits(getattr(a, 'actual_fsift' + s)) for s in SIFT_SLOTS] for a in A], np.uint16).reshape(n, 4)
    out['atom_integer_sift'] = np.array([[getattr(a, 'integer_sift' + s) for s in SIFT_SLOTS] for a in A], np.uint8).reshape(n, 4, 15)
    out['atom_counts'] = np.array([[getattr(a, c) for c in COUNTERS] for a in A], np.int32).reshape(n, 8)
    out['atom_potential_fsift'] = np.array([bits(a.potential_fsift) for a in A], np.uint16)
    ap = self_.atom_plane_contacts
    ring_index = {id(self_.biopython_str.rings[k]): k for k in self_.biopython_str.rings}
    # the record does not hold the ring id: recover it from the (residue, sorted atom names, distance) it was built from
    ap_ring = []
    for c in ap:
        cands = [k for k, rg in self_.biopython_str.rings.items()
                 if rg['residue'] is c.end_res and sorted(a.get_id() for a in rg['atoms']) == c.end_res_atoms
                 and np.linalg.norm(c.bgn_atom.coord - rg['center']) == c.distance]
        assert len(cands) == 1, cands
        ap_ring.append(cands[0])
    del ring_index
    ap_names = pconfig.ATOM_PLANE_NAMES
    out['ap_atom'] = np.array([c.bgn_atom.idx for c in ap], np.int32)
    out['ap_ring'] = np.array(ap_ring, np.int32)
    out['ap_dist'] = np.array([c.distance for c in ap], np.float64)
    out['ap_mask'] = np.array([sum(1 << ap_names.index(s) for s in c.sifts) for c in ap], np.uint8)
    out['ap_ctype'] = np.array([ct_code[c.text] for c in ap], np.uint8)
    assert all(c.sifts == sorted(c.sifts) for c in ap)
    pp_names = pconfig.PLANE_PLANE_NAMES

    def planes(lst, pref, with_types):
        out[pref + '_bgn'] = np.array([c.bgn_id for c in lst], np.int32)
        out[pref + '_end'] = np.array([c.end_id for c in lst], np.int32)
        out[pref + '_dist'] = np.array([c.distance for c in lst], np.float64)
        out[pref + '_ctype'] = np.array([ct_code[c.text] for c in lst], np.uint8)
        if with_types:
            out[pref + '_type1'] = np.array([pp_names.index(c.contact_type[0]) for c in lst], np.uint8)
            out[pref + '_type2'] = np.array([pp_names.index(c.contact_type[1]) if len(c.contact_type) > 1 else 255 for c in lst], np.uint8)

    planes(self_.plane_plane_contacts, 'pp', True)
    planes(self_.group_group_contacts, 'gg', False)
    planes(self_.group_plane_contacts, 'gp', False)
    R = self_.biopython_str.residues
    for nm, w in (('ring_ring_inter_integer_sift', 9), ('ring_atom_inter_integer_sift', 5), ('atom_ring_inter_integer_sift', 5),
                  ('mc_atom_ring_inter_integer_sift', 5), ('sc_atom_ring_inter_integer_sift', 5),
                  ('amide_amide_inter_integer_sift', 1), ('amide_ring_inter_integer_sift', 1), ('ring_amide_inter_integer_sift', 1)):
        out['res_' + nm] = np.array([getattr(r, nm) for r in R], np.int32).reshape(len(R), w)
    return out


def run_case(IC, config, pc, selectors=None, sel_idx=None, cutoff=5.0, comp=0.1, seq_adj=False, mode='canonical', seed=0,
             bond_order=None, bond_aromatic=None, with_export=False, extras=None):
    NeighborSearch.mode, NeighborSearch.seed = mode, seed
    self_ = holders_from_pack(IC, pc, bond_order, bond_aromatic, extras)
    self_.structure_checks()
    if sel_idx is not None:      # an explicit atom list instead of selector strings: patch the parser call of I:1395
        chosen = [self_.s_atoms[int(k)] for k in sel_idx]
        saved = IC._make_selection.__globals__['utils'].selection_parser
        IC._make_selection.__globals__['utils'].selection_parser = lambda selections, entity: list(chosen)
        try:
            with np.errstate(all='ignore'):
                self_.run_arpeggio(['<explicit atom list>'], cutoff, comp, seq_adj)
        finally:
            IC._make_selection.__globals__['utils'].selection_parser = saved
    else:
        with np.errstate(all='ignore'):
            self_.run_arpeggio(list(selectors or []), cutoff, comp, seq_adj)
    out = dump_run(self_, pc, config)
    export = None
    if with_export:
        export = {'get_contacts': json.dumps(self_.get_contacts(), sort_keys=True)}
        with tempfile.TemporaryDirectory() as wd:
            self_.write_contacts(self_.selection if (selectors or sel_idx is not None) else [], wd)
            self_.write_atom_types(wd)
            self_.write_atom_sifts(wd)
            self_.write_residue_sifts(wd)
            self_.write_polar_matching(wd)
            for fn in sorted(os.listdir(wd)):
                export[fn] = open(os.path.join(wd, fn), newline='').read()
    return out, export


def pack_arrays(pc, prefix):
    return pc.to_arrays(prefix)


# ------------------------------------------------------------------------------------------------ the packs
def repoint_hydrogen_neighbours(pc):
    """get_single_bond_neighbour never returns a hydrogen (U:630): a stored neighbour that is one moves to the next heavy atom."""
    is_h = (pc.flags & pconfig.F_HYDROGEN) != 0
    heavy = np.nonzero(~is_h)[0]
    for i in np.nonzero((pc.sb_nbr >= 0) & is_h[np.maximum(pc.sb_nbr, 0)])[0].tolist():
        k = np.searchsorted(heavy, pc.sb_nbr[i])
        pc.sb_nbr[i] = heavy[k % len(heavy)] if heavy[k % len(heavy)] != i else heavy[(k + 1) % len(heavy)]


def soup_with_rings(seed, n_uniform=420, box=15.0, n_rings=26, n_amides=24):
    """Dense soup: every type / flag combination on the uniform atoms, hexagonal rings and amide groups between them."""
    pc = synth.make_synthetic(n_uniform, seed=seed, box=(box, box, box), n_rings=n_rings, n_amides=n_amides, water_frac=0.08,
                              atoms_per_residue=5, residues_per_chain=12, id=f'soup{seed}')
    rng = np.random.default_rng(1000 + seed)
    nu = n_uniform
    tm = pc.type_mask.copy()
    for b in range(12):
        flip = rng.random(nu) < 0.22
        tm[:nu] = np.where(flip, tm[:nu] ^ np.uint16(1 << b), tm[:nu])
    pc.type_mask = tm
    fl = pc.flags.copy()
    heavy = np.arange(nu)
    fl[heavy[rng.random(nu) < 0.04]] |= pconfig.F_METAL
    fl[heavy[rng.random(nu) < 0.08]] |= pconfig.F_HALOGEN
    # a few explicit hydrogens and deuteriums among the atoms (I:712: 'H' is skipped, 'D' is not)
    hyd = heavy[rng.random(nu) < 0.04]
    fl[hyd] = (fl[hyd] & ~np.uint16(pconfig.F_ELEM_C | pconfig.F_ELEM_S)) | pconfig.F_HYDROGEN
    pc.flags = fl
    pc.ensure_labels()
    el = list(pc.element)
    for i in hyd.tolist():
        el[i] = 'H'
    deut = [i for i in heavy[rng.random(nu) < 0.02].tolist() if el[i] not in ('C', 'S', 'H')]
    for i in deut:
        el[i] = 'D'
    pc.element = el
    # xbond donors need a single-bond neighbour in the reference (U:173 dereferences None); leave two without to be
    # handled by the dedicated known-answer pack, give the others one
    xd = (pc.type_mask & pconfig.ATOM_TYPE_BIT['xbond donor']) != 0
    lonely = np.nonzero(xd & (pc.sb_nbr < 0))[0]
    pc.sb_nbr[lonely] = (lonely + 7) % pc.n_atoms
    # some halogens / donors deliberately WITHOUT a neighbour: is_halogen_weak_hbond returns 0 for them (U:139-141)
    hal = np.nonzero(((pc.flags & pconfig.F_HALOGEN) != 0) & ~xd)[0]
    pc.sb_nbr[hal[::3]] = -1
    repoint_hydrogen_neighbours(pc)
    # ring 1 has no residue (I:1479): never in any ring id set
    pc.ring_res[1] = -1
    # residues: some polypeptide flags off, some sequence links cut
    nres = pc.n_residues
    cut = rng.random(nres) < 0.1
    pc.res_next[cut] = -1
    pc.validate()
    return pc


def protein_bond_orders(pc):
    """Bond orders for synth.proteinlike(): ring bonds aromatic (order 1), C=O double, everything else single."""
    n = pc.n_atoms
    ring_pairs = set()
    for ids in pc.ring_atoms:
        ids = [int(x) for x in ids]
        for q in range(len(ids)):
            ring_pairs.add(frozenset((ids[q], ids[(q + 1) % len(ids)])))
    order = np.ones(len(pc.bond_idx), np.uint8)
    arom = np.zeros(len(pc.bond_idx), np.uint8)
    for i in range(n):
        for k in range(pc.bond_off[i], pc.bond_off[i + 1]):
            j = int(pc.bond_idx[k])
            if frozenset((i, j)) in ring_pairs:
                arom[k] = 1
            names = {pc.atom_name[i], pc.atom_name[j]}
            if names == {'C', 'O'} or names == {'CG', 'OD1'}:
                order[k] = 2
    return order, arom


def finish_geometry(rpc):
    """read_mmcif leaves the centres / normals / residues of its template rings and amide groups to the GPU (initialize()).  A
    fixture must not depend on a GPU: the amide groups go through the EXECUTED reference (_perceive_amide_groups, I:1531-1589, and
    _assign_aromatic_rings_to_residues, I:1453-1492 — make_golden_prepare.run), the ring centres / normals through the restatement
    of OpenBabel's OBRing::findCenterAndNormal (third-party, recalled: oracle/ref_py.ring_geometry)."""
    if not getattr(rpc, 'plane_geometry_pending', False):
        return
    import copy
    import make_golden_prepare as prep
    from oracle import ref_py
    rc, rn = ref_py.ring_geometry(rpc.xyz, rpc.ring_atoms)
    geo = {}
    prep.run(copy.deepcopy(rpc), [list(map(int, q)) for q in rpc.amide_atoms], rc, 'g', geo)
    rpc.ring_center, rpc.ring_normal, rpc.ring_res = np.asarray(rc, np.float64).reshape(-1, 3), np.asarray(rn, np.float64).reshape(-1, 3), geo['g/ring_res']
    rpc.amide_center, rpc.amide_normal, rpc.amide_res = geo['g/amide_center'], geo['g/amide_normal'], geo['g/amide_res']
    rpc.plane_geometry_pending = False


def reader_section(IC, config, add, arrays, utils):
    # ---- D. a structure that came through the mmCIF reader (core/protein_reader.read_mmcif): alternative locations (the B
    # child wins on occupancy), insertion codes 17A / 17B, a modified residue in the chain, a chain break, hetero groups, waters
    # of two names, and a heavy water (element D: not a hydrogen at I:712, but its deuterons are hydrogens to OpenBabel, I:1524)
    import tempfile
    from make_golden_reader import peptide_atom_site, chem_comp, cif_text
    from arpeggio_amd.core import protein_reader
    cols = peptide_atom_site(np.random.default_rng(5), 0)
    nrow = len(cols['id'])

    def add_row(**kv):
        for k in cols:
            cols[k].append(kv.get(k, cols[k][nrow - 1]))
    ox = np.array([float(cols['Cartn_x'][3]) + 2.9, float(cols['Cartn_y'][3]) + 0.4, float(cols['Cartn_z'][3]) + 0.3])
    for nm, el, d in (('O', 'O', (0, 0, 0)), ('D1', 'D', (0.76, 0.59, 0.0)), ('D2', 'D', (-0.76, 0.59, 0.0))):
        add_row(group_PDB='HETATM', id=str(len(cols['id']) + 1), type_symbol=el, label_atom_id=nm, label_alt_id=None, label_comp_id='DOD',
                label_asym_id='A', label_seq_id=False, pdbx_PDB_ins_code=None, Cartn_x='%.3f' % (ox[0] + d[0]), Cartn_y='%.3f' % (ox[1] + d[1]),
                Cartn_z='%.3f' % (ox[2] + d[2]), occupancy='1.00', pdbx_formal_charge=None, auth_seq_id='450', auth_asym_id='A', pdbx_PDB_model_num='1')
    cc = chem_comp(0)
    cc['id'].append('DOD'); cc['type'].append('NON-POLYMER'); cc['name'].append('DEUTERATED WATER')
    for k in cc:
        if len(cc[k]) < len(cc['id']):
            cc[k].append(None)
    text = cif_text('READER', [('_atom_site.', cols), ('_chem_comp.', cc)], singles=[('_entry.id', 'READER')])
    with tempfile.TemporaryDirectory() as td:
        fn = os.path.join(td, 'reader_h.cif')
        open(fn, 'w').write(text)
        rpc = protein_reader.read_mmcif(fn)
    rpc.id = 'reader'
    finish_geometry(rpc)
    assert (rpc.amide_res >= 0).all() and np.abs(rpc.amide_center).max() > 0 and rpc.n_amides > 0
    arrays['reader/cif_text'] = np.array(text)
    for tag, kw in (('whole', dict()), ('chain_b', dict(selectors=['/B//']))):
        out, _ = run_case(IC, config, rpc, **kw)
        add(f'reader:{tag}', rpc, out, pack='reader', selectors=kw.get('selectors'), sel=None, cutoff=5.0, comp=0.1, seq_adj=False, mode='canonical')
    # ---- D2. a file with complete aromatic / amide side chains and explicit hydrogens (make_golden_reader.aromatic_atom_site): the
    # ring and amide loops on a structure that came from a FILE — template rings / amides, template bonds, hydrogens by distance
    from make_golden_reader import aromatic_atom_site
    cols2 = aromatic_atom_site()
    cc2 = {'id': sorted(set(cols2['label_comp_id'])), 'type': None, 'name': None}
    cc2['type'] = ['peptide linking' if c == 'GLY' else 'L-peptide linking' for c in cc2['id']]
    cc2['name'] = [c + ' RESIDUE' for c in cc2['id']]
    text2 = cif_text('RINGS', [('_atom_site.', cols2), ('_chem_comp.', cc2)], singles=[('_entry.id', 'RINGS')])
    with tempfile.TemporaryDirectory() as td:
        fn = os.path.join(td, 'rings_h.cif')
        open(fn, 'w').write(text2)
        rpc2 = protein_reader.read_mmcif(fn)
    rpc2.id = 'reader_rings'
    finish_geometry(rpc2)
    assert rpc2.n_rings >= 10 and rpc2.n_amides >= 18 and (rpc2.ring_res >= 0).all()
    arrays['reader_rings/cif_text'] = np.array(text2)
    # the bonds inside the standard residues come from read_mmcif's residue templates (protein_reader.RESIDUE_BONDS): holders
    # carrying those bonds with their orders and aromatic flags, get_single_bond_neighbour (U:612-635) EXECUTED for every atom
    arrays['reader_rings/bond_order'], arrays['reader_rings/bond_aromatic'] = rpc2.bond_order, rpc2.bond_aromatic
    h2 = holders_from_pack(IC, rpc2, rpc2.bond_order, rpc2.bond_aromatic)
    sb2 = np.full(rpc2.n_atoms, -1, np.int32)
    for a_ in h2.s_atoms:
        nb = utils.get_single_bond_neighbour(h2.ob_mol.GetAtomById(h2.bio_to_ob[a_]))
        if nb is not None:
            sb2[a_.idx] = h2.ob_to_bio[nb.GetId()].idx
    arrays['reader_rings/sb_nbr_reference'] = sb2
    assert np.array_equal(sb2, rpc2.sb_nbr), 'read_mmcif: single-bond neighbours differ from the executed get_single_bond_neighbour'
    for tag, kw in (('whole', dict()), ('tyr', dict(selectors=['RESNAME:TYR'])), ('a25', dict(selectors=['/A/25/']))):
        out, _ = run_case(IC, config, rpc2, bond_order=rpc2.bond_order, bond_aromatic=rpc2.bond_aromatic, **kw)
        add(f'reader_rings:{tag}', rpc2, out, pack='reader_rings', selectors=kw.get('selectors'), sel=None, cutoff=5.0, comp=0.1, seq_adj=False, mode='canonical')
        assert len(out['ap_atom']) and (tag != 'whole' or (len(out['pp_bgn']) and len(out['gg_bgn']) and len(out['gp_bgn']))), tag


def check_reader():
    """Re-runs section D alone and compares every array it makes with the committed core_cases.npz (the CPU suite calls this: a
    fixture that no longer is what its generator makes must not go unnoticed again).  Returns the number of arrays compared."""
    IC, utils, config, exceptions = reference_namespace()
    arrays, seen = {}, []

    def add(name, pc, out, **params):
        if f'{params["pack"]}/xyz' not in arrays:
            ex = extras_for(pc)
            pc.lone_pair_electrons = (np.asarray(config.VALENCE)[ex['atomic_number']] - ex['bond_order'] - ex['formal_charge']).astype(np.int32)
            arrays.update(pack_arrays(pc, params['pack'] + '/'))
        for k, v in out.items():
            arrays[f'{name}/{k}'] = v
        seen.append(name)

    reader_section(IC, config, add, arrays, utils)
    z = np.load(os.path.join(HERE, 'core_cases.npz'), allow_pickle=False)
    bad = [k for k, v in arrays.items() if k not in z.files or np.asarray(v).dtype != z[k].dtype or np.asarray(v).shape != z[k].shape
           or np.asarray(v).tobytes() != z[k].tobytes()]
    stale = [k for k in z.files if k.split('/')[0] in set(seen) | {'reader'} and k not in arrays]
    assert not bad and not stale, (bad[:10], stale[:10])
    return len(arrays)


def main():
    IC, utils, config, exceptions = reference_namespace()
    from helpers import known_answer_packs, random_dense_pack
    arrays, meta, exports = {}, [], {}

    def add(name, pc, out, **params):
        pre = f'{name}/'
        if f'{params["pack"]}/xyz' not in arrays:
            ex = extras_for(pc)                # the operands of I:1804, from the same stand-in values the holders carry
            pc.lone_pair_electrons = (np.asarray(config.VALENCE)[ex['atomic_number']] - ex['bond_order'] - ex['formal_charge']).astype(np.int32)
            arrays.update(pack_arrays(pc, params['pack'] + '/'))
        for k, v in out.items():
            arrays[pre + k] = v
        meta.append(dict(name=name, **params))
        print(f'{name}: {len(out["aa_bgn"])} atom-atom, {len(out["ap_atom"])} atom-plane, {len(out["pp_bgn"])} plane-plane, '
              f'{len(out["gg_bgn"])} group-group, {len(out["gp_bgn"])} group-plane; |selection_plus| = {len(out["selection_plus"])}')

    # ---- A. the hand-built branch packs of tests/helpers.py, both orientations; the crash case of U:173 separately
    for nm, pc in known_answer_packs():
        pc.id = 'ka_' + nm
        pc.ensure_labels()
        for mode in ('canonical', 'reversed'):
            out, _ = run_case(IC, config, pc, mode=mode)
            add(f'ka_{nm}:{mode}', pc, out, pack='ka_' + nm, selectors=None, sel=None, cutoff=5.0, comp=0.1, seq_adj=False, mode=mode)
    # ---- A2. pairs ON every distance threshold and one float32 ulp to either side (tests/helpers.threshold_edge_pack)
    from helpers import threshold_edge_pack
    edge = threshold_edge_pack()
    edge.id = 'edge'
    edge.ensure_labels()
    for mode in ('canonical', 'reversed'):
        out, _ = run_case(IC, config, edge, mode=mode)
        add(f'edge:{mode}', edge, out, pack='edge', selectors=None, sel=None, cutoff=5.0, comp=0.1, seq_adj=False, mode=mode)
    sel6 = edge.expansion_probe[0::2]           # the A atoms of the pairs at 6.0 A -/0/+ one ulp: which B atoms join selection_plus
    arrays['edge/sel_idx'] = sel6.astype(np.int32)
    out, _ = run_case(IC, config, edge, sel_idx=sel6)
    assert sorted(out['selection_plus'].tolist()) == sorted(edge.expansion_probe[[0, 1, 2, 3, 4]].tolist()), out['selection_plus']
    add('edge:sel6', edge, out, pack='edge', selectors=None, sel='edge/sel_idx', cutoff=5.0, comp=0.1, seq_adj=False, mode='canonical')
    T = pconfig.ATOM_TYPE_BIT
    from helpers import tiny_complex
    crash = tiny_complex([[0, 0, 0], [3.0, 0, 0]], type_mask=[T['xbond donor'], T['xbond acceptor']], sb_nbr=[-1, -1])
    crash.ensure_labels()
    try:
        run_case(IC, config, crash)
        raise SystemExit('is_xbond without a neighbour was expected to raise')
    except AttributeError as e:
        arrays.update(pack_arrays(crash, 'xbond_crash/'))
        meta.append(dict(name='xbond_crash', pack='xbond_crash', raises='AttributeError', message=str(e)))

    # ---- B. dense soups
    for seed in (1, 2):
        pc = random_dense_pack(seed)
        pc.id = f'dense{seed}'
        pc.ensure_labels()
        # flag bits consistent with the string tables the reference reads (element, resname)
        fl = pc.flags.copy()
        both = ((fl & pconfig.F_ELEM_C) != 0) & ((fl & pconfig.F_ELEM_S) != 0)
        fl[both] &= ~np.uint16(pconfig.F_ELEM_S)
        hyd = (fl & pconfig.F_HYDROGEN) != 0
        fl[hyd] &= ~np.uint16(pconfig.F_ELEM_C | pconfig.F_ELEM_S)
        met_res = np.zeros(pc.n_residues, bool)
        met_res[pc.res_id[(fl & pconfig.F_RES_MET) != 0]] = True
        met_res &= (np.arange(pc.n_residues) % 3 == 0)
        fl = np.where(met_res[pc.res_id], fl | pconfig.F_RES_MET, fl & ~np.uint16(pconfig.F_RES_MET)).astype(np.uint16)
        pc.flags = fl
        pc.element = ['H' if f & pconfig.F_HYDROGEN else ('C' if f & pconfig.F_ELEM_C else ('S' if f & pconfig.F_ELEM_S else 'X')) for f in fl.tolist()]
        pc.res_name = ['MET' if m else 'UNK' for m in met_res.tolist()]
        pc.component_types = {'MET': 'P', 'UNK': 'P'}
        repoint_hydrogen_neighbours(pc)
        rng = np.random.default_rng(50 + seed)
        sel_res = rng.random(pc.n_residues) < 0.3
        sel_idx = np.nonzero(sel_res[pc.res_id])[0]
        for (tag, kw) in (('whole', dict()), ('sel', dict(sel_idx=sel_idx)), ('sel_seqadj', dict(sel_idx=sel_idx, seq_adj=True)),
                          ('sel_comp', dict(sel_idx=sel_idx, comp=0.25, cutoff=4.5)), ('sel_random', dict(sel_idx=sel_idx, mode='random', seed=seed))):
            out, _ = run_case(IC, config, pc, **kw)
            add(f'dense{seed}:{tag}', pc, out, pack=f'dense{seed}', selectors=None,
                sel=(None if 'sel_idx' not in kw else f'dense{seed}/sel_idx'), cutoff=kw.get('cutoff', 5.0), comp=kw.get('comp', 0.1),
                seq_adj=kw.get('seq_adj', False), mode=kw.get('mode', 'canonical'))
        arrays[f'dense{seed}/sel_idx'] = sel_idx.astype(np.int32)

    for seed in (3, 4):
        pc = soup_with_rings(seed)
        rng = np.random.default_rng(70 + seed)
        sel_res = rng.random(pc.n_residues) < 0.25
        sel_idx = np.nonzero(sel_res[pc.res_id])[0]
        arrays[f'soup{seed}/sel_idx'] = sel_idx.astype(np.int32)
        for (tag, kw) in (('whole', dict(with_export=(seed == 3))), ('sel', dict(sel_idx=sel_idx, with_export=(seed == 3))),
                          ('sel_reversed', dict(sel_idx=sel_idx, mode='reversed')),
                          ('sel_seqadj_random', dict(sel_idx=sel_idx, seq_adj=True, mode='random', seed=seed))):
            out, exp = run_case(IC, config, pc, **kw)
            name = f'soup{seed}:{tag}'
            add(name, pc, out, pack=f'soup{seed}', selectors=None, sel=(None if 'sel_idx' not in kw else f'soup{seed}/sel_idx'),
                cutoff=5.0, comp=0.1, seq_adj=kw.get('seq_adj', False), mode=kw.get('mode', 'canonical'))
            if exp:
                exports[name] = exp

    # ---- C. the 1tqn_h stand-in (BASELINE configs[0]/[1]): real bond orders, selector strings, exports
    pc = synth.proteinlike()
    order, arom = protein_bond_orders(pc)
    arrays['proteinlike/bond_order'], arrays['proteinlike/bond_aromatic'] = order, arom
    # get_single_bond_neighbour (U:612-635) executed for every atom on real bond data
    h0 = holders_from_pack(IC, pc, order, arom)
    sb_ref = np.full(pc.n_atoms, -1, np.int32)
    for a in h0.s_atoms:
        nb = utils.get_single_bond_neighbour(h0.ob_mol.GetAtomById(h0.bio_to_ob[a]))
        if nb is not None:
            sb_ref[a.idx] = h0.ob_to_bio[nb.GetId()].idx
    arrays['proteinlike/sb_nbr_reference'] = sb_ref
    pc.sb_nbr = sb_ref.copy()        # the pack carries what the reference's own function returns
    for tag, selectors in (('lig508', ['/A/508/']), ('resname_phe', ['RESNAME:PHE']), ('ligands', ['LIGANDS']),
                           ('two', ['/A/508/FE', '/A/100/'])):
        out, exp = run_case(IC, config, pc, selectors=selectors, bond_order=order, bond_aromatic=arom, with_export=(tag in ('lig508', 'two')))
        name = f'proteinlike:{tag}'
        add(name, pc, out, pack='proteinlike', selectors=selectors, sel=None, cutoff=5.0, comp=0.1, seq_adj=False, mode='canonical')
        if exp:
            exports[name] = exp
    # the WHOLE stand-in (BASELINE configs[1]: no selectors = every atom, I:1395) — I:709 rebuilds set(self.selection) for every
    # neighbour pair, ~1e9 hash inserts here: minutes, once — and the same with sequence-adjacent residues included
    for tag, kw in (('whole', dict()), ('whole_seqadj', dict(seq_adj=True)), ('lig508_seqadj_comp', dict(selectors=['/A/508/'], seq_adj=True, comp=0.3, cutoff=4.5))):
        t_run = time.time()
        out, _ = run_case(IC, config, pc, bond_order=order, bond_aromatic=arom, **kw)
        add(f'proteinlike:{tag}', pc, out, pack='proteinlike', selectors=kw.get('selectors'), sel=None, cutoff=kw.get('cutoff', 5.0),
            comp=kw.get('comp', 0.1), seq_adj=kw.get('seq_adj', False), mode='canonical')
        print(f'   ({time.time() - t_run:.0f} s)')
    reader_section(IC, config, add, arrays, utils)
    small = synth.proteinlike(n_res=110, n_waters=70, id='proteinlike_small')
    o2, a2 = protein_bond_orders(small)
    out, exp = run_case(IC, config, small, bond_order=o2, bond_aromatic=a2, with_export=True)
    add('proteinlike_small:whole', small, out, pack='proteinlike_small', selectors=None, sel=None, cutoff=5.0, comp=0.1, seq_adj=False,
        mode='canonical')
    exports['proteinlike_small:whole'] = exp

    OUT = os.environ.get('ARP_GOLDEN_OUT', HERE)      # (tests/test_fixture_freshness.py regenerates into a scratch directory)
    np.savez_compressed(os.path.join(OUT, 'core_cases.npz'), **arrays)
    json.dump(meta, open(os.path.join(OUT, 'core_cases.json'), 'w'), indent=1)
    # exports: JSON / CSV text, gzip (mtime 0: reproducible bytes)
    buf = io.BytesIO()
    with gzip.GzipFile(fileobj=buf, mode='wb', mtime=0) as gz:
        gz.write(json.dumps(exports, sort_keys=True).encode())
    open(os.path.join(OUT, 'core_exports.json.gz'), 'wb').write(buf.getvalue())
    print('sha256 of exports:', hashlib.sha256(buf.getvalue()).hexdigest()[:16], len(buf.getvalue()), 'bytes')
    print(len(meta), 'cases written')


if __name__ == '__main__':
    main()
