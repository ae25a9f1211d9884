"""Shared helpers for the test-suite (oracle-side utilities; never imported by the product)."""
import numpy as np

from arpeggio_amd.core.packed import PackedComplex


def planes_only_complex(ring_center, ring_normal, ring_res, amide_center, amide_normal, amide_res, nres):
    """A PackedComplex with no atoms, just rings and amides (golden plane fixtures)."""
    z = np.zeros
    return PackedComplex(
        xyz=z((0, 3), np.float32), vdw=z(0), cov=z(0), type_mask=z(0, np.uint16), flags=z(0, np.uint16),
        res_id=z(0, np.int32), res_flags=z(int(nres), np.uint8), res_prev=np.full(int(nres), -1, np.int32),
        res_next=np.full(int(nres), -1, np.int32), bond_off=z(1, np.int32), bond_idx=z(0, np.int32),
        h_off=z(1, np.int32), h_xyz=z((0, 3)), sb_nbr=z(0, np.int32),
        ring_center=ring_center, ring_normal=ring_normal, ring_res=ring_res,
        amide_center=amide_center, amide_normal=amide_normal, amide_res=amide_res)


def tiny_complex(xyz, vdw=1.7, cov=0.76, type_mask=0, flags=0, res_id=None, res_flags=None, res_prev=None,
                 res_next=None, bonds=(), h=None, sb_nbr=None, rings=None, amides=None):
    """Hand-built pack for known-answer tests.  bonds: iterable of (i, j); h: dict atom -> list of xyz."""
    xyz = np.asarray(xyz, np.float32).reshape(-1, 3)
    n = xyz.shape[0]

    def per_atom(v, dt):
        a = np.empty(n, dt)
        a[:] = v
        return a

    res_id = np.arange(n, dtype=np.int32) if res_id is None else np.asarray(res_id, np.int32)
    nres = int(res_id.max()) + 1 if n else 0
    res_flags = np.zeros(nres, np.uint8) if res_flags is None else np.asarray(res_flags, np.uint8)
    res_prev = np.full(nres, -1, np.int32) if res_prev is None else np.asarray(res_prev, np.int32)
    res_next = np.full(nres, -1, np.int32) if res_next is None else np.asarray(res_next, np.int32)
    adj = [[] for _ in range(n)]
    for i, j in bonds:
        adj[i].append(j)
        adj[j].append(i)
    bond_off = np.concatenate([[0], np.cumsum([len(a) for a in adj])]).astype(np.int32)
    bond_idx = np.array([j for a in adj for j in a], np.int32)
    h = h or {}
    hl = [np.asarray(h.get(i, []), np.float64).reshape(-1, 3) for i in range(n)]
    h_off = np.concatenate([[0], np.cumsum([x.shape[0] for x in hl])]).astype(np.int32)
    h_xyz = np.concatenate(hl, axis=0) if n else np.zeros((0, 3))
    if sb_nbr is None:
        sb_nbr = np.array([a[0] if a else -1 for a in adj], np.int32)
    rc, rn, rr = rings if rings is not None else (np.zeros((0, 3)), np.zeros((0, 3)), np.zeros(0, np.int32))
    ac, an, ar = amides if amides is not None else (np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), np.zeros(0, np.int32))
    return PackedComplex(
        xyz=xyz, vdw=per_atom(vdw, np.float64), cov=per_atom(cov, np.float64), type_mask=per_atom(type_mask, np.uint16),
        flags=per_atom(flags, np.uint16), res_id=res_id, res_flags=res_flags, res_prev=res_prev, res_next=res_next,
        bond_off=bond_off, bond_idx=bond_idx, h_off=h_off, h_xyz=h_xyz, sb_nbr=np.asarray(sb_nbr, np.int32),
        ring_center=rc, ring_normal=rn, ring_res=rr, amide_center=ac, amide_normal=an, amide_res=ar)


def deg_close(a, b, tol=1e-4):
    """Angles equal within tol degrees, NaN == NaN."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    return np.all(both_nan | (np.abs(a - b) <= tol))
