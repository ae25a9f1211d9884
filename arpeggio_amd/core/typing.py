"""Atom typing of standard residues by table, and element radii.

What the reference does in ``initialize()`` for the atoms of the twenty standard amino acids needs no cheminformatics:
``_ob_atom_typing`` ends by throwing away whatever SMARTS matching said about them and reading their types from a
dictionary keyed by ``<RESNAME><ATOMNAME>`` (interactions.py:1966-1983; table config.py:150-590), waters are donors and
acceptors by decree (interactions.py:1953-1956), and radii are per-element constants (interactions.py:1494-1511).  With
these three a protein-only structure can be typed without OpenBabel; ligand atoms still need its SMARTS engine and keep
whatever mask the packer gives them.

The table is data of the drop-in contract and is kept as data (``data/prot_atom_types.json``, regenerated from the
reference's config.py by ``tools/make_typing_table.py``), accidents included: two missing commas in the reference fuse
'GLNOE1' + 'GLNNE2' and 'TRPCD1' + 'TRPCE3' into keys that match no atom, so those atoms LACK the corresponding type, and
'TRYCB' is spelt that way.  Pinned against the executed reference code by tests/golden/typing.json.
"""
from __future__ import annotations

import json
import os

import numpy as np

from . import config

_HERE = os.path.dirname(os.path.abspath(__file__))
_TABLE = None


def table():
    global _TABLE
    if _TABLE is None:
        _TABLE = json.load(open(os.path.join(_HERE, 'data', 'prot_atom_types.json')))
    return _TABLE


def _mask_of(names):
    return sum(config.ATOM_TYPE_BIT[n] for n in names)


def key_masks(use_ambiguities=False):
    """'<RESNAME><ATOMNAME>' -> 12-bit type mask.  ``use_ambiguities``: the table after ``address_ambiguities()``
    (interactions.py:120-133), which strikes the ASN / GLN / HIS side-chain keys from four acceptor / donor lists."""
    t = table()
    out = {}
    for key, names in t['keys'].items():
        if use_ambiguities:
            names = [n for n in names if key not in t['ambiguous'].get(n, ())]
        out[key] = _mask_of(names)
    return out


def table_type_mask():
    """Bits of the types the dictionary covers: for a standard residue they are CLEARED before the dictionary is read
    (interactions.py:1973-1975); 'xbond donor' is not among them and survives."""
    return _mask_of(table()['table_types'])


def apply_protein_typing(pc, use_ambiguities=False, waters=True):
    """New ``type_mask`` for a PackedComplex: atoms of standard residues from the dictionary, water atoms as hbond donor +
    acceptor, every other atom as it was (interactions.py:1952-1983).  Returns a uint16 array; the pack is not modified."""
    pc.ensure_labels()
    t = table()
    std = set(t['std_res'])
    masks = key_masks(use_ambiguities)
    covered = np.uint16(table_type_mask())
    out = pc.type_mask.copy()
    if use_ambiguities:
        # I:124-127: the four ambiguous SMARTS patterns are struck as well, which changes the types of ligand / hetero
        # atoms.  Their matches are OpenBabel's: a pack made where it is available carries the re-typed masks.
        alt = getattr(pc, 'type_mask_ambiguities', None)
        if alt is not None:
            out = np.asarray(alt, np.uint16).copy()
        else:
            t_ = table()
            nonstd = np.array([rn not in set(t_['std_res']) for rn in pc.res_name], bool)[pc.res_id]
            affected = np.uint16(config.ATOM_TYPE_BIT['hbond acceptor'] | config.ATOM_TYPE_BIT['hbond donor'] |
                                 config.ATOM_TYPE_BIT['xbond acceptor'] | config.ATOM_TYPE_BIT['weak hbond acceptor'])
            water = (pc.flags & config.F_WATER) != 0
            k = int((nonstd & ~water & ((pc.type_mask & affected) != 0)).sum())
            if k:
                import logging
                logging.warning('address_ambiguities: %d atoms of non-standard residues keep the SMARTS types they were packed with '
                                '(the pack has no type_mask_ambiguities: terminal-amide N / O of ligands are typed as without -a)', k)
    if waters:      # interactions.py:1953-1956 (before the dictionary override, so a standard residue flagged W loses them again)
        w = (pc.flags & config.F_WATER) != 0
        out[w] |= np.uint16(config.ATOM_TYPE_BIT['hbond acceptor'] | config.ATOM_TYPE_BIT['hbond donor'])
    res_std = np.array([rn in std for rn in pc.res_name], bool)
    rows = np.nonzero(res_std[pc.res_id])[0]
    stripped = [rn.strip() for rn in pc.res_name]
    for i in rows.tolist():
        out[i] = (out[i] & ~covered) | np.uint16(masks.get(stripped[pc.res_id[i]] + pc.atom_name[i].strip(), 0))
    return out


# Element radii.  The reference asks OpenBabel: ob.GetVdwRad(Z), ob.GetCovalentRad(Z) (interactions.py:1501, 1509).
# OpenBabel is not part of the reference tree and not installed here; the values below are restated from the element
# table of OpenBabel 3.1 (src/elementtable.h: covalent radii of Cordero et al. 2008, van der Waals radii of Mantina et al.
# 2009 / the Bondi-style extensions it ships) FROM MEMORY — "recalled", to be re-verified wherever OpenBabel is
# available.  A structure packed with OpenBabel present carries OpenBabel's own numbers in PackedComplex.vdw / .cov;
# this table only serves packers without it.
ELEMENT_RADII = {  # symbol: (vdw, cov)
    'H': (1.10, 0.31), 'D': (1.10, 0.31), 'HE': (1.40, 0.28), 'LI': (1.81, 1.28), 'BE': (1.53, 0.96), 'B': (1.92, 0.84),
    'C': (1.70, 0.76), 'N': (1.55, 0.71), 'O': (1.52, 0.66), 'F': (1.47, 0.57), 'NE': (1.54, 0.58), 'NA': (2.27, 1.66),
    'MG': (1.73, 1.41), 'AL': (1.84, 1.21), 'SI': (2.10, 1.11), 'P': (1.80, 1.07), 'S': (1.80, 1.05), 'CL': (1.75, 1.02),
    'AR': (1.88, 1.06), 'K': (2.75, 2.03), 'CA': (2.31, 1.76), 'SC': (2.30, 1.70), 'TI': (2.15, 1.60), 'V': (2.05, 1.53),
    'CR': (2.05, 1.39), 'MN': (2.05, 1.39), 'FE': (2.05, 1.32), 'CO': (2.00, 1.26), 'NI': (2.00, 1.24), 'CU': (2.00, 1.32),
    'ZN': (2.10, 1.22), 'GA': (1.87, 1.22), 'GE': (2.11, 1.20), 'AS': (1.85, 1.19), 'SE': (1.90, 1.20), 'BR': (1.83, 1.20),
    'KR': (2.02, 1.16), 'RB': (3.03, 2.20), 'SR': (2.49, 1.95), 'MO': (2.10, 1.54), 'AG': (2.10, 1.45), 'CD': (2.20, 1.44),
    'I': (1.98, 1.39), 'XE': (2.16, 1.40), 'CS': (3.43, 2.44), 'BA': (2.68, 2.15), 'PT': (2.05, 1.36), 'AU': (2.10, 1.36),
    'HG': (2.05, 1.32), 'PB': (2.02, 1.46),
}


def element_radii(elements):
    """(vdw, cov) float64 arrays for a list of element symbols (upper-cased, stripped); KeyError for an element the table
    does not hold — a silent default would change contact classes."""
    sym = [e.strip().upper() for e in elements]
    missing = sorted({s for s in sym if s not in ELEMENT_RADII})
    if missing:
        raise KeyError(f'no radii for element(s) {missing}: pass them in PackedComplex.vdw / .cov')
    r = np.array([ELEMENT_RADII[s] for s in sym], np.float64).reshape(-1, 2)
    return r[:, 0].copy(), r[:, 1].copy()


def element_flags(elements, res_names=None, res_id=None):
    """The flag bits a packer derives from strings (interactions.py:1990-1991, 712, 1009, 1023): metal, halogen, hydrogen,
    carbon, sulphur and — with the residue names — 'residue is MET'."""
    f = np.zeros(len(elements), np.uint16)
    for i, e in enumerate(elements):
        u = e.upper()
        if u in config.METALS:
            f[i] |= config.F_METAL
        if u in config.HALOGENS:
            f[i] |= config.F_HALOGEN
        if e.strip() == 'H':
            f[i] |= config.F_HYDROGEN
        if e == 'C':
            f[i] |= config.F_ELEM_C
        if e == 'S':
            f[i] |= config.F_ELEM_S
    if res_names is not None and res_id is not None:
        met = np.array([rn == 'MET' for rn in res_names], bool)[np.asarray(res_id)]
        f[met] |= config.F_RES_MET
    return f
