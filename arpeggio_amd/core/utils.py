"""Host-side helpers of the drop-in API: the selection mini-language and entity labels.

Behavioural mirror of the reference's ``utils.selection_parser`` (arpeggio/core/utils.py:396-526),
``make_pymol_json`` (utils.py:530-564), ``make_pymol_string`` (utils.py:567-609) and
``get_residue_name`` (utils.py:748-767), operating on the string tables of a
PackedComplex instead of BioPython objects.  Pure string logic; nothing here is on the
GPU path.  Pinned against the executed reference function by
tests/golden/selection_parser.json.
"""
from __future__ import annotations

import logging
import os
import platform

import numpy as np

from . import config
from .exceptions import SelectionError


def max_mem_usage():
    """utils.py:55-67."""
    try:
        import resource
        base = 1024.0 if platform.system() == 'Linux' else 1048576.0
        return str(round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / base, 2)) + ' MB'
    except Exception as err:  # pragma: no cover
        return 'Resource usage information not available {}'.format(str(err))


def is_digit(x):
    try:
        int(x)
        return True
    except ValueError:
        return False


class _Tables:
    """String columns of a PackedComplex as object arrays (built once per parser call)."""

    def __init__(self, pc):
        pc.ensure_labels()
        self.pc = pc
        self.res_of = pc.res_id
        self.res_name = np.asarray(pc.res_name, dtype=object)
        self.res_chain = np.asarray(pc.res_chain, dtype=object)
        self.res_seq = np.asarray(pc.res_seq)
        self.res_icode = np.asarray(pc.res_icode, dtype=object)
        self.atom_name = np.asarray(pc.atom_name, dtype=object)


def _select_resname(t, text, original):
    name = text.replace('RESNAME:', '').strip()
    if len(name) > 3:
        raise SelectionError(original)
    return (np.array([r.strip() for r in t.res_name], dtype=object) == name)[t.res_of]


def _select_ligands(t):
    pc, nres = t.pc, t.pc.n_residues
    size = np.bincount(t.res_of, minlength=nres)
    with_carbon = np.zeros(nres, bool)
    with_carbon[t.res_of[np.asarray(pc.element, dtype=object) == 'C']] = True
    names = [r.strip().upper() for r in t.res_name]
    excluded = config.COMMON_SOLVENTS | config.STANDARD_NUCLEOTIDES
    ligand = np.fromiter((not (pc.res_flags[r] & config.R_POLYPEPTIDE) and 5 <= size[r] <= 100 and with_carbon[r]
                          and names[r] not in excluded and not t.res_name[r].startswith('+') for r in range(nres)), bool, nres)
    return ligand[t.res_of]


def _residue_field(text, original):
    """'<number>[<insertion letter>]' -> (number, insertion code); anything else is a bad selector."""
    if is_digit(text):
        return int(text), ' '
    if text.isalnum() and text[-1].isalpha() and is_digit(text[:-1]):
        return int(text[:-1]), text[-1]
    raise SelectionError(original)


def _select_path(t, text, original):
    fields = text.lstrip('/').split('/')
    if len(fields) != 3:
        raise SelectionError(original)
    chain, residue, name = fields
    mask = np.ones(t.pc.n_atoms, bool)
    number, icode = _residue_field(residue, original) if residue else (None, ' ')
    if name and not (name.isalnum() or "'" in name):
        raise SelectionError(original)
    if chain:
        mask &= (t.res_chain == chain)[t.res_of]
    if number:          # a residue number of 0 is falsy, hence silently ignored — as in the reference (utils.py:508)
        mask &= ((t.res_seq == number) & (t.res_icode == icode))[t.res_of]
    if name:
        mask &= t.atom_name == name
    return mask


def selection_parser(selection_list, pc):
    """Atoms selected by a list of selectors (their union); returns a sorted int array of packed atom indices.

    Behavioural counterpart of ``utils.selection_parser`` (arpeggio/core/utils.py:396-526), pinned against the executed
    reference function by tests/golden/selection_parser.json.  Selector forms: ``/<chain>/<resnum>[<inscode>]/<atom_name>``
    with exactly three fields (any may be empty), ``RESNAME:<up to 3 chars>`` and ``LIGANDS``; anything else, and an empty
    result, raise ``SelectionError``.
    """
    t = _Tables(pc)
    chosen = np.zeros(pc.n_atoms, bool)
    for original in selection_list:
        text = original.strip()
        if text.startswith('RESNAME:'):
            chosen |= _select_resname(t, text, original)
        elif text.startswith('LIGANDS'):
            chosen |= _select_ligands(t)
        elif text.startswith('/'):
            chosen |= _select_path(t, text, original)
        else:
            raise SelectionError(original)
    if not chosen.any():
        logging.error('Selection was empty.')
        raise SelectionError('entity not found')
    return np.nonzero(chosen)[0].astype(np.int64)


def make_pymol_json(pc, atom=None, residue=None):
    """utils.py:530-564 for an atom index or a residue index of a PackedComplex."""
    if atom is not None:
        r = int(pc.res_id[atom])
        return {
            'label_comp_id': pc.res_name[r],
            'auth_seq_id': int(pc.res_seq[r]),
            'auth_asym_id': pc.res_chain[r],
            'auth_atom_id': pc.atom_name[atom],
            'pdbx_PDB_ins_code': pc.res_icode[r],
        }
    if residue is not None:
        r = int(residue)
        return {
            'label_comp_id': pc.res_name[r],
            'auth_seq_id': int(pc.res_seq[r]),
            'auth_asym_id': pc.res_chain[r],
            'pdbx_PDB_ins_code': pc.res_icode[r],
        }
    raise TypeError('Cannot make a json object from non-Atom/Residue object.')


def make_pymol_string(pc, atom=None, residue=None):
    """utils.py:567-609: chain/resnum[icode]/atom-name."""
    if atom is not None:
        r, atom_name = int(pc.res_id[atom]), pc.atom_name[atom]
    elif residue is not None:
        r, atom_name = int(residue), ''
    else:
        raise TypeError('Cannot make a PyMOL string from a non-Atom or Residue object.')
    res_num = int(pc.res_seq[r])
    if pc.res_icode[r] != ' ':
        res_num = str(res_num) + pc.res_icode[r]
    return '{}/{}/{}'.format(pc.res_chain[r], res_num, atom_name)


def get_residue_name(pc, atom=None, residue=None):
    """utils.py:748-767."""
    if atom is not None:
        return pc.res_name[int(pc.res_id[atom])]
    if residue is not None:
        return pc.res_name[int(residue)]
    raise TypeError('Cannot return Residue from from non-Atom/non-Residue object.')
