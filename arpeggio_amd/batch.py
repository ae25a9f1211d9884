"""Several structures in ONE pass (include/arpeggio_hip.h: arp_set_batch).

The reference runs one structure per process; its production use is the weekly PDBe release — ~10^5 entries of a few
thousand atoms each (/root/reference README.md:4) — and a protein-sized structure cannot fill an MI355X: a 5.9 k-atom
pass is four launches of fixed cost.  Here B structures are uploaded as one concatenated PackedComplex (atom, residue,
ring and amide indices shifted by the structure's offsets; coordinates UNCHANGED, so every distance is the one the single
run computes) and the library keeps them apart in the grid: every structure's cells sit at an integer offset with an
empty cell around them.  One `run_launch` then evaluates all of them; the results come back as one set of bags whose ids
are split per structure here.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np

from .core.packed import PackedComplex


def concat_complexes(pcs: Sequence[PackedComplex]):
    """One PackedComplex holding all of ``pcs`` + the offsets / boxes arp_set_batch needs."""
    B = len(pcs)
    a_off = np.zeros(B + 1, np.int64); r_off = np.zeros(B + 1, np.int64)
    g_off = np.zeros(B + 1, np.int64); m_off = np.zeros(B + 1, np.int64)
    h_cnt = np.zeros(B + 1, np.int64); b_cnt = np.zeros(B + 1, np.int64)
    for k, pc in enumerate(pcs):
        a_off[k + 1] = a_off[k] + pc.n_atoms; r_off[k + 1] = r_off[k] + pc.n_residues
        g_off[k + 1] = g_off[k] + pc.n_rings; m_off[k + 1] = m_off[k] + pc.n_amides
        h_cnt[k + 1] = h_cnt[k] + len(pc.h_xyz); b_cnt[k + 1] = b_cnt[k] + len(pc.bond_idx)

    def shift(arr, off):          # indices with -1 = none
        arr = np.asarray(arr)
        return np.where(arr >= 0, arr + off, arr).astype(np.int32)

    def cat(name, dtype=None):
        return np.concatenate([np.asarray(getattr(pc, name)) for pc in pcs]) if B else np.zeros(0, dtype)

    boxes = np.zeros((B, 6), np.float64)
    for k, pc in enumerate(pcs):
        pts = [pc.xyz.astype(np.float64)]
        if pc.n_rings:
            pts.append(pc.ring_center)
        if pc.n_amides:
            pts.append(pc.amide_center.astype(np.float64))
        allp = np.concatenate(pts) if sum(len(p) for p in pts) else np.zeros((1, 3))
        boxes[k, :3] = allp.min(axis=0)
        boxes[k, 3:] = allp.max(axis=0)
    # float32 atoms must lie inside the box after the library's float32 -> float64 conversion: min / max of the same values do
    big = PackedComplex(
        xyz=cat('xyz'), vdw=cat('vdw'), cov=cat('cov'), type_mask=cat('type_mask'), flags=cat('flags'),
        res_id=np.concatenate([pc.res_id + r_off[k] for k, pc in enumerate(pcs)]).astype(np.int32),
        res_flags=cat('res_flags'),
        res_prev=np.concatenate([shift(pc.res_prev, r_off[k]) for k, pc in enumerate(pcs)]),
        res_next=np.concatenate([shift(pc.res_next, r_off[k]) for k, pc in enumerate(pcs)]),
        bond_off=np.concatenate([[0]] + [pc.bond_off[1:].astype(np.int64) + b_cnt[k] for k, pc in enumerate(pcs)]).astype(np.int32),
        bond_idx=np.concatenate([pc.bond_idx.astype(np.int64) + a_off[k] for k, pc in enumerate(pcs)]).astype(np.int32),
        h_off=np.concatenate([[0]] + [pc.h_off[1:].astype(np.int64) + h_cnt[k] for k, pc in enumerate(pcs)]).astype(np.int32),
        h_xyz=np.concatenate([pc.h_xyz.reshape(-1, 3) for pc in pcs]),
        sb_nbr=np.concatenate([shift(pc.sb_nbr, a_off[k]) for k, pc in enumerate(pcs)]),
        ring_center=np.concatenate([pc.ring_center.reshape(-1, 3) for pc in pcs]),
        ring_normal=np.concatenate([pc.ring_normal.reshape(-1, 3) for pc in pcs]),
        ring_res=np.concatenate([shift(pc.ring_res, r_off[k]) for k, pc in enumerate(pcs)]),
        amide_center=np.concatenate([pc.amide_center.reshape(-1, 3) for pc in pcs]),
        amide_normal=np.concatenate([pc.amide_normal.reshape(-1, 3) for pc in pcs]),
        amide_res=np.concatenate([shift(pc.amide_res, r_off[k]) for k, pc in enumerate(pcs)]),
        id='batch[%d]' % B)
    return big, dict(atom=a_off, residue=r_off, ring=g_off, amide=m_off, boxes=boxes)


def _split(res: Dict[str, np.ndarray], key: str, off: np.ndarray, shifts: Dict[str, np.ndarray]) -> List[Dict[str, np.ndarray]]:
    """Cut one bag into its structures: records are assigned by the structure of column ``key`` (both ends of a record
    belong to the same structure by construction) and ids become structure-local (``shifts``: column -> offsets)."""
    sid = np.searchsorted(off, res[key], side='right') - 1
    order = np.argsort(sid, kind='stable')
    sid_sorted = sid[order]
    bounds = np.searchsorted(sid_sorted, np.arange(len(off)))
    out = []
    for s in range(len(off) - 1):
        sl = order[bounds[s]:bounds[s + 1]]
        d = {k: v[sl] for k, v in res.items()}
        for col, o in shifts.items():
            d[col] = (d[col] - o[s]).astype(np.int32)
        out.append(d)
    return out


def split_atom_contacts(res, offsets):
    return _split(res, 'i', offsets['atom'], {'i': offsets['atom'], 'j': offsets['atom']})


def split_bag(name, res, offsets):
    a, r, m = offsets['atom'], offsets['ring'], offsets['amide']
    if name == 'atom_plane':
        return _split(res, 'ring', r, {'atom': a, 'ring': r})
    if name == 'plane_plane':
        return _split(res, 'bgn', r, {'bgn': r, 'end': r})
    if name == 'group_group':
        return _split(res, 'bgn', m, {'bgn': m, 'end': m})
    if name == 'group_plane':
        return _split(res, 'amide', m, {'amide': m, 'ring': r})
    raise KeyError(name)
