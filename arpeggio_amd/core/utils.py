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


def selection_parser(selection_list, pc):
    """Atoms selected by a list of selectors; returns a sorted int array of packed atom indices.

    Selector forms (additive): ``/<chain>/<resnum>[<inscode>]/<atom_name>`` with exactly
    three fields (any may be empty), ``RESNAME:<up to 3 chars>`` and ``LIGANDS``.
    Quirks kept from the reference: a residue number of 0 is ignored (utils.py:508), atom
    names must be alphanumeric or contain an apostrophe (utils.py:499), an empty result
    raises ``SelectionError('entity not found')`` (utils.py:522-524).
    """
    pc.ensure_labels()
    n = pc.n_atoms
    res_of = pc.res_id
    res_name = np.asarray(pc.res_name, dtype=object)
    res_chain = np.asarray(pc.res_chain, dtype=object)
    res_seq = np.asarray(pc.res_seq)
    res_icode = np.asarray(pc.res_icode, dtype=object)
    atom_name = np.asarray(pc.atom_name, dtype=object)
    final = np.zeros(n, bool)

    for selection in selection_list:
        residue_number = None
        insertion_code = ' '
        chain = None
        name = None
        original_selection = selection
        selection = selection.strip()
        current = np.ones(n, bool)

        if selection.startswith('RESNAME:'):
            selection = selection.replace('RESNAME:', '').strip()
            if len(selection) > 3:  # RESNAMES ARE MAX LENGTH 3
                raise SelectionError(original_selection)
            stripped = np.array([r.strip() for r in res_name], dtype=object)
            final |= (stripped == selection)[res_of]

        elif selection.startswith('LIGANDS'):
            nres = pc.n_residues
            natoms = np.bincount(res_of, minlength=nres)
            elem = np.asarray(pc.element, dtype=object)
            has_c = np.zeros(nres, bool)
            has_c[res_of[elem == 'C']] = True
            upper = [r.strip().upper() for r in res_name]
            ok = np.array([
                not (pc.res_flags[r] & config.R_POLYPEPTIDE)             # MUST NOT BE POLYPEPTIDE
                and 5 <= natoms[r] <= 100                                 # MIN / MAX NUMBER OF ATOMS
                and has_c[r]                                              # MUST CONTAIN CARBON
                and upper[r] not in config.COMMON_SOLVENTS                # MUST NOT BE COMMON SOLVENT
                and upper[r] not in config.STANDARD_NUCLEOTIDES           # MUST NOT BE NUCLEOTIDE
                and not res_name[r].startswith('+')                       # MUST NOT BE MODIFIED NUCLEOTIDE
                for r in range(nres)], bool) if nres else np.zeros(0, bool)
            final |= ok[res_of]

        elif selection.startswith('/'):
            fields = selection.lstrip('/').split('/')
            if len(fields) != 3:
                raise SelectionError(original_selection)
            if fields[0]:
                chain = fields[0]
            if fields[1]:
                if is_digit(fields[1]):
                    residue_number = int(fields[1])
                elif fields[1].isalnum():
                    if fields[1][-1].isalpha() and is_digit(fields[1][:-1]):
                        residue_number = int(fields[1][:-1])
                        insertion_code = fields[1][-1]
                    else:
                        raise SelectionError(original_selection)
                else:
                    raise SelectionError(original_selection)
            if fields[2]:
                if not fields[2].isalnum() and "'" not in fields[2]:
                    raise SelectionError(original_selection)
                name = fields[2]
            if chain:
                current &= (res_chain == chain)[res_of]
            if residue_number:   # 0 is falsy: residue number 0 is silently ignored (utils.py:508)
                current &= ((res_seq == residue_number) & (res_icode == insertion_code))[res_of]
            if name:
                current &= (atom_name == name)
            final |= current

        else:
            raise SelectionError(original_selection)

    if not final.any():
        logging.error('Selection was empty.')
        raise SelectionError('entity not found')
    return np.nonzero(final)[0].astype(np.int64)


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
