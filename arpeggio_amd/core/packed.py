"""PackedComplex — the struct-of-arrays input contract of the hot path.

It holds exactly what the reference's ``InteractionComplex.initialize()``
(arpeggio/core/interactions.py:288-327) leaves on its BioPython/OpenBabel objects
and what ``run_arpeggio`` (interactions.py:329-347) reads, as flat NumPy arrays
that can be handed to the C ABI (include/arpeggio_hip.h) without conversion.

Numeric arrays go to the GPU; the string tables stay on the host and are used
only by the selection parser and the JSON export.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from . import config


def _arr(x, dtype, shape_tail=()):
    a = np.ascontiguousarray(x, dtype=dtype)
    if shape_tail:
        a = a.reshape((-1,) + tuple(shape_tail))
    return a


def single_bond_neighbours(bond_off, bond_idx, bond_order, bond_aromatic, is_hydrogen):
    """utils.get_single_bond_neighbour (utils.py:612-635) for every atom: the first bond, in the bond graph's own
    order, that has order 1, is not aromatic and leads to a non-hydrogen atom; -1 where there is none (the reference
    returns None).  ``bond_order`` / ``bond_aromatic`` are aligned with ``bond_idx`` (CSR)."""
    bond_off = np.asarray(bond_off, np.int64)
    bond_idx = np.asarray(bond_idx, np.int64)
    n = len(bond_off) - 1
    ok = (np.asarray(bond_order) == 1) & (np.asarray(bond_aromatic) == 0) & ~np.asarray(is_hydrogen, bool)[bond_idx]
    pos = np.nonzero(ok)[0]
    owner = np.searchsorted(bond_off, pos, side='right') - 1
    out = np.full(n, -1, np.int32)
    first = np.ones(len(pos), bool)
    first[1:] = owner[1:] != owner[:-1]
    out[owner[first]] = bond_idx[pos[first]]
    return out


@dataclass
class PackedComplex:
    # ---- atoms: self.s_atoms (interactions.py:54-60) ----
    xyz: np.ndarray            # f32 [N,3]   atom.coord (protein_reader.py:327)
    vdw: np.ndarray            # f64 [N]     atom.vdw_radius (interactions.py:1501)
    cov: np.ndarray            # f64 [N]     atom.cov_radius (interactions.py:1509)
    type_mask: np.ndarray      # u16 [N]     atom.atom_types, bit = config.ATOM_TYPE_NAMES index
    flags: np.ndarray          # u16 [N]     config.F_* bits
    res_id: np.ndarray         # i32 [N]     index into the residue table
    # ---- residues ----
    res_flags: np.ndarray      # u8  [NR]    config.R_* bits (interactions.py:1671,1860)
    res_prev: np.ndarray       # i32 [NR]    prev_residue, -1 = None (interactions.py:1687-1693)
    res_next: np.ndarray       # i32 [NR]
    # ---- OpenBabel bond graph, CSR over atom indices (interactions.py:750) ----
    bond_off: np.ndarray       # i32 [N+1]
    bond_idx: np.ndarray       # i32 [M]
    # ---- atom.h_coords, CSR (interactions.py:1513-1529) ----
    h_off: np.ndarray          # i32 [N+1]
    h_xyz: np.ndarray          # f64 [NH,3]
    # ---- utils.get_single_bond_neighbour (utils.py:612-635), -1 = None ----
    sb_nbr: np.ndarray         # i32 [N]
    # ---- rings (interactions.py:1697-1733, 1453-1492) ----
    ring_center: np.ndarray    # f64 [R,3]
    ring_normal: np.ndarray    # f64 [R,3]
    ring_res: np.ndarray       # i32 [R]     -1 = residue None
    ring_atoms: List[np.ndarray] = field(default_factory=list)   # atom indices per ring
    # ---- amides (interactions.py:1531-1589) ----
    amide_center: np.ndarray = None   # f32 [A,3]
    amide_normal: np.ndarray = None   # f32 [A,3]
    amide_res: np.ndarray = None      # i32 [A]
    amide_atoms: np.ndarray = None    # i32 [A,4]  N, C, O, C
    # ---- host-side string tables (JSON export / selection parser only) ----
    atom_name: Optional[List[str]] = None
    element: Optional[List[str]] = None
    serial: Optional[np.ndarray] = None          # i32 [N] atom.serial_number
    res_name: Optional[List[str]] = None         # residue.resname
    res_seq: Optional[np.ndarray] = None         # i32 [NR] residue.id[1]
    res_icode: Optional[List[str]] = None        # residue.id[2]
    res_chain: Optional[List[str]] = None        # chain.id
    component_types: Optional[Dict[str, str]] = None   # interactions.py:70
    # ---- config.VALENCE[atomic_number] - atom.bond_order - atom.formal_charge (interactions.py:1804): what the
    #      potential hbond / polar counts of acceptors are made of (write_polar_matching only) ----
    lone_pair_electrons: Optional[np.ndarray] = None   # i32 [N]
    # ---- the SMARTS types of every atom with the four ambiguous patterns struck (address_ambiguities, I:120-133: 'NH2
    #      terminal amide' as hbond / xbond / weak hbond acceptor, 'oxygen amide term' as hbond donor); produced by the packer
    #      where OpenBabel is available; None: unknown ----
    type_mask_ambiguities: Optional[np.ndarray] = None   # u16 [N]
    id: str = 'packed'

    def __post_init__(self):
        self.xyz = _arr(self.xyz, np.float32, (3,))
        n = self.xyz.shape[0]
        self.vdw = _arr(self.vdw, np.float64)
        self.cov = _arr(self.cov, np.float64)
        self.type_mask = _arr(self.type_mask, np.uint16)
        self.flags = _arr(self.flags, np.uint16)
        self.res_id = _arr(self.res_id, np.int32)
        self.res_flags = _arr(self.res_flags, np.uint8)
        self.res_prev = _arr(self.res_prev, np.int32)
        self.res_next = _arr(self.res_next, np.int32)
        self.bond_off = _arr(self.bond_off, np.int32)
        self.bond_idx = _arr(self.bond_idx, np.int32)
        self.h_off = _arr(self.h_off, np.int32)
        self.h_xyz = _arr(self.h_xyz, np.float64, (3,))
        self.sb_nbr = _arr(self.sb_nbr, np.int32)
        self.ring_center = _arr(self.ring_center, np.float64, (3,))
        self.ring_normal = _arr(self.ring_normal, np.float64, (3,))
        self.ring_res = _arr(self.ring_res, np.int32)
        if self.amide_center is None:
            self.amide_center = np.zeros((0, 3), np.float32)
            self.amide_normal = np.zeros((0, 3), np.float32)
            self.amide_res = np.zeros((0,), np.int32)
        self.amide_center = _arr(self.amide_center, np.float32, (3,))
        self.amide_normal = _arr(self.amide_normal, np.float32, (3,))
        self.amide_res = _arr(self.amide_res, np.int32)
        if self.amide_atoms is None:
            self.amide_atoms = np.full((self.amide_res.shape[0], 4), -1, np.int32)
        self.amide_atoms = _arr(self.amide_atoms, np.int32, (4,))
        self.validate()
        assert n == self.n_atoms

    # ---- sizes ----
    @property
    def n_atoms(self):
        return int(self.xyz.shape[0])

    @property
    def n_residues(self):
        return int(self.res_flags.shape[0])

    @property
    def n_rings(self):
        return int(self.ring_res.shape[0])

    @property
    def n_amides(self):
        return int(self.amide_res.shape[0])

    def validate(self):
        n, nr = self.n_atoms, self.n_residues
        for name in ('vdw', 'cov', 'type_mask', 'flags', 'res_id', 'sb_nbr'):
            if getattr(self, name).shape != (n,):
                raise ValueError(f'{name}: expected shape ({n},), got {getattr(self, name).shape}')
        if self.res_prev.shape != (nr,) or self.res_next.shape != (nr,):
            raise ValueError('res_prev/res_next must have one entry per residue')
        if n and (self.res_id.min() < 0 or self.res_id.max() >= nr):
            raise ValueError('res_id out of range')
        for off, idx, what in ((self.bond_off, self.bond_idx, 'bond'), (self.h_off, self.h_xyz, 'h')):
            if off.shape != (n + 1,) or off[0] != 0 or np.any(np.diff(off) < 0) or off[-1] != idx.shape[0]:
                raise ValueError(f'{what}_off is not a valid CSR offset array')
        if self.bond_idx.size and (self.bond_idx.min() < 0 or self.bond_idx.max() >= n):
            raise ValueError('bond_idx out of range')
        if n and (self.sb_nbr.min() < -1 or self.sb_nbr.max() >= n):
            raise ValueError('sb_nbr out of range')
        if self.ring_normal.shape != self.ring_center.shape or self.ring_center.shape[0] != self.n_rings:
            raise ValueError('ring arrays disagree in length')
        if self.amide_normal.shape != self.amide_center.shape or self.amide_center.shape[0] != self.n_amides:
            raise ValueError('amide arrays disagree in length')
        for arr, what in ((self.ring_res, 'ring_res'), (self.amide_res, 'amide_res')):
            if arr.size and (arr.min() < -1 or arr.max() >= nr):
                raise ValueError(f'{what} out of range')

    # ---- string tables with defaults (synthetic packs carry none) ----
    def rings_not_in_path_order(self):
        """Indices of the rings whose ``ring_atoms`` form a cycle of the bond graph (every atom has two bonded partners in
        the ring) but are not LISTED along it (some consecutive pair, the last and the first included, is not bonded).
        Rings whose atoms are not bonded among themselves (packs without that part of the bond graph) cannot be judged and
        are not reported."""
        if not self.ring_atoms or self.bond_idx.shape[0] == 0:
            return []
        bad = []
        for r, atoms in enumerate(self.ring_atoms):
            a = [int(x) for x in atoms]
            if len(a) < 3:
                continue
            members = set(a)
            nbrs = {i: members.intersection(self.bond_idx[self.bond_off[i]:self.bond_off[i + 1]].tolist()) for i in a}
            if any(len(v) < 2 for v in nbrs.values()):
                continue
            if any(a[(k + 1) % len(a)] not in nbrs[a[k]] for k in range(len(a))):
                bad.append(r)
        return bad

    def ensure_labels(self):
        n, nr = self.n_atoms, self.n_residues
        if self.res_name is None:
            self.res_name = ['UNK'] * nr
        if self.res_seq is None:
            self.res_seq = np.arange(1, nr + 1, dtype=np.int32)
        if self.res_icode is None:
            self.res_icode = [' '] * nr
        if self.res_chain is None:
            self.res_chain = ['A'] * nr
        if self.atom_name is None:
            self.atom_name = [f'X{i}' for i in range(n)]
        if self.element is None:
            self.element = ['H' if f & config.F_HYDROGEN else ('C' if f & config.F_ELEM_C else ('S' if f & config.F_ELEM_S else 'X'))
                            for f in self.flags.tolist()]
        if self.serial is None:
            self.serial = np.arange(1, n + 1, dtype=np.int32)
        if self.component_types is None:
            self.component_types = {}
        if self.lone_pair_electrons is not None:      # (None stays None: the polar-matching writer says so instead of writing zeros)
            self.lone_pair_electrons = _arr(self.lone_pair_electrons, np.int32)
        if self.type_mask_ambiguities is not None:
            self.type_mask_ambiguities = _arr(self.type_mask_ambiguities, np.uint16)
        for rn in sorted(set(self.res_name)):      # (sorted: the order of a set of strings changes from process to process)
            self.component_types.setdefault(rn, 'P')
        return self

    def type_names(self, i):
        m = int(self.type_mask[i])
        return {name for b, name in enumerate(config.ATOM_TYPE_NAMES) if m >> b & 1}

    # ---- persistence (a packed structure is the "file" this implementation reads) ----
    _ARRAYS = ('xyz', 'vdw', 'cov', 'type_mask', 'flags', 'res_id', 'res_flags', 'res_prev', 'res_next', 'bond_off',
               'bond_idx', 'h_off', 'h_xyz', 'sb_nbr', 'ring_center', 'ring_normal', 'ring_res', 'amide_center',
               'amide_normal', 'amide_res', 'amide_atoms', 'serial', 'res_seq', 'lone_pair_electrons')
    _LISTS = ('atom_name', 'element', 'res_name', 'res_icode', 'res_chain')

    def to_arrays(self, prefix=''):
        """Every field as a NumPy array under ``prefix + name`` (what ``save`` writes)."""
        self.ensure_labels()
        d = {prefix + k: getattr(self, k) for k in self._ARRAYS if getattr(self, k) is not None}
        if self.type_mask_ambiguities is not None:
            d[prefix + 'type_mask_ambiguities'] = self.type_mask_ambiguities
        for k in self._LISTS:
            d[prefix + k] = np.array(getattr(self, k), dtype=np.str_)
        ro = np.concatenate([[0], np.cumsum([len(a) for a in self.ring_atoms])]).astype(np.int32) if self.ring_atoms \
            else np.zeros(self.n_rings + 1, np.int32)
        d[prefix + 'ring_atoms_off'] = ro
        d[prefix + 'ring_atoms_idx'] = (np.concatenate(self.ring_atoms).astype(np.int32) if self.ring_atoms and ro[-1]
                                        else np.zeros(0, np.int32))
        d[prefix + 'component_types_keys'] = np.array(list(self.component_types.keys()), dtype=np.str_)
        d[prefix + 'component_types_vals'] = np.array(list(self.component_types.values()), dtype=np.str_)
        d[prefix + 'id'] = np.array(self.id)
        return d

    @classmethod
    def from_arrays(cls, z, prefix=''):
        kw = {k: z[prefix + k] for k in cls._ARRAYS + ('type_mask_ambiguities',) if (prefix + k) in z}
        for k in cls._LISTS:
            kw[k] = [str(x) for x in z[prefix + k]]
        ro, ri = z[prefix + 'ring_atoms_off'], z[prefix + 'ring_atoms_idx']
        kw['ring_atoms'] = [ri[ro[r]:ro[r + 1]] for r in range(len(ro) - 1)]
        kw['component_types'] = {str(k): str(v) for k, v in zip(z[prefix + 'component_types_keys'], z[prefix + 'component_types_vals'])}
        kw['id'] = str(z[prefix + 'id'])
        return cls(**kw)

    def save(self, path):
        """Write the pack to a ``.npz`` file (numeric arrays + string tables)."""
        np.savez_compressed(path, **self.to_arrays())

    @classmethod
    def load(cls, path):
        return cls.from_arrays(np.load(path, allow_pickle=False))
