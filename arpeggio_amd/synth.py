"""Deterministic synthetic PackedComplex generators (BASELINE.json configs 2-5).

The reference ships no input files or fixtures, so every workload
here is generated from a counter-based splitmix64 stream: the same (seed, index)
gives the same value on every machine, independent of NumPy's RNG.

Nothing here is on the hot path; it only manufactures what
``InteractionComplex.initialize()`` (interactions.py:288-327) would leave behind.
"""
from __future__ import annotations

import numpy as np

from .core import config
from .core.packed import PackedComplex

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed: int, idx) -> np.ndarray:
    """splitmix64 of (seed + (idx+1) * golden gamma); idx is an integer array."""
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over='ignore'):
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + (idx + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def u01(seed: int, stream: int, idx) -> np.ndarray:
    """Uniform [0,1) with 24 random bits (exactly representable in float32)."""
    s = (seed * 0x1000003 + stream * 0x632BE5AB) & 0xFFFFFFFFFFFFFFFF
    return (splitmix64(s, idx) >> np.uint64(40)).astype(np.float64) / float(1 << 24)


def _unit_vectors(seed, stream, idx):
    """Uniform unit vectors (float64) from two uniforms."""
    z = 2.0 * u01(seed, stream, idx) - 1.0
    phi = 2.0 * np.pi * u01(seed, stream + 1, idx)
    r = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    return np.stack([r * np.cos(phi), r * np.sin(phi), z], axis=1)


def _rotation_matrices(seed, stream, idx):
    """Random rotations: columns (e1, e2, e3) orthonormal, float64 [K,3,3]."""
    e3 = _unit_vectors(seed, stream, idx)
    t = _unit_vectors(seed, stream + 2, idx)
    e1 = np.cross(e3, t)
    nrm = np.linalg.norm(e1, axis=1, keepdims=True)
    bad = nrm[:, 0] < 1e-6
    if np.any(bad):
        e1[bad] = np.cross(e3[bad], np.array([1.0, 0.0, 0.0]))
        nrm = np.linalg.norm(e1, axis=1, keepdims=True)
    e1 /= nrm
    e2 = np.cross(e3, e1)
    return np.stack([e1, e2, e3], axis=2)


# element table: (symbol, cdf weight, vdw, cov) — radii are OpenBabel-style element
# table values (synthetic constants here; the real ones come from I:1501,1509)
_ELEMENTS = [
    ('C', 0.62, 1.70, 0.76),
    ('N', 0.17, 1.55, 0.71),
    ('O', 0.19, 1.52, 0.66),
    ('S', 0.01, 1.80, 1.05),
    ('CL', 0.005, 1.75, 1.02),
    ('ZN', 0.005, 1.39, 1.22),
]
_EL_C, _EL_N, _EL_O, _EL_S, _EL_CL, _EL_ZN = range(6)

T = config.ATOM_TYPE_BIT


def _close_pairs(xyz32, radius):
    """All pairs (i<j) with float64 distance < radius; sorted, deterministic."""
    from scipy.spatial import cKDTree
    if xyz32.shape[0] < 2:
        return np.zeros((0, 2), np.int64)
    tree = cKDTree(xyz32.astype(np.float64))
    pairs = tree.query_pairs(radius, output_type='ndarray')
    if pairs.size == 0:
        return np.zeros((0, 2), np.int64)
    pairs = np.sort(pairs, axis=1)
    order = np.lexsort((pairs[:, 1], pairs[:, 0]))
    return pairs[order].astype(np.int64)


def _uniform_atoms(seed, idx, box, origin, water_frac):
    """Coordinates (float32), water flag, element and type mask of the uniform atoms with indices ``idx`` — every quantity a
    function of (seed, index) alone, so any subset of a structure can be generated without the rest."""
    idx = np.asarray(idx, dtype=np.uint64)
    xyz_u = np.stack([u01(seed, 10 + a, idx) * box[a] + origin[a] for a in range(3)], axis=1).astype(np.float32) if idx.size else np.zeros((0, 3), np.float32)
    is_water = u01(seed, 20, idx) < water_frac
    ue = u01(seed, 21, idx)
    cdf = np.cumsum([e[1] for e in _ELEMENTS])
    cdf[-1] = 1.0 + 1e-9
    elem_u = np.searchsorted(cdf, ue, side='right').astype(np.int64)
    elem_u[is_water] = _EL_O

    def p(stream):
        return u01(seed, stream, idx)

    tm = np.zeros(idx.size, np.uint32)
    o, nn, c, s_, cl = (elem_u == _EL_O), (elem_u == _EL_N), (elem_u == _EL_C), (elem_u == _EL_S), (elem_u == _EL_CL)
    acc = (o & (p(30) < 0.9)) | (nn & (p(30) < 0.2)) | (s_ & (p(30) < 0.3))
    don = (o & (p(31) < 0.3)) | (nn & (p(31) < 0.7))
    tm[acc] |= T['hbond acceptor'] | T['weak hbond acceptor']
    tm[acc & (p(32) < 0.9)] |= T['xbond acceptor']
    tm[don] |= T['hbond donor']
    tm[o & (p(33) < 0.5)] |= T['carbonyl oxygen']
    tm[o & (p(34) < 0.1)] |= T['neg ionisable']
    tm[nn & (p(34) < 0.1)] |= T['pos ionisable']
    wdon = c & (p(35) < 0.6)
    tm[wdon] |= T['weak hbond donor']
    tm[(c & (p(36) < 0.4)) | (s_ & (p(36) < 0.5)) | (cl & (p(36) < 0.5))] |= T['hydrophobe']
    tm[c & (p(37) < 0.15)] |= T['aromatic']
    tm[c & (p(38) < 0.2)] |= T['carbonyl carbon']
    tm[cl] |= T['weak hbond acceptor']
    # waters are donors and acceptors (interactions.py:1953-1956)
    tm[is_water] = T['hbond acceptor'] | T['hbond donor']
    return xyz_u, is_water, elem_u, tm


def _ring_atoms(seed, ridx, box, origin):
    """float32 [K, 6, 3]: the six atoms of the rings with indices ``ridx`` (regular hexagons, radius 1.39)."""
    ridx = np.asarray(ridx, dtype=np.uint64)
    if ridx.size == 0:
        return np.zeros((0, 6, 3), np.float32)
    rc = np.stack([u01(seed, 50 + a, ridx) * box[a] + origin[a] for a in range(3)], axis=1)
    rrot = _rotation_matrices(seed, 53, ridx)
    ang = np.arange(6) * (np.pi / 3.0)
    hexa = 1.39 * (np.cos(ang)[None, :, None] * rrot[:, None, :, 0] + np.sin(ang)[None, :, None] * rrot[:, None, :, 1])
    return (rc[:, None, :] + hexa).astype(np.float32)


def _amide_atoms(seed, aidx, box, origin):
    """float32 [K, 4, 3]: N, C, O, CA of the amide groups with indices ``aidx`` (planar)."""
    aidx = np.asarray(aidx, dtype=np.uint64)
    if aidx.size == 0:
        return np.zeros((0, 4, 3), np.float32)
    ac = np.stack([u01(seed, 60 + a, aidx) * box[a] + origin[a] for a in range(3)], axis=1)
    arot = _rotation_matrices(seed, 63, aidx)
    local = np.array([[1.33 * np.cos(np.deg2rad(120.0)), 1.33 * np.sin(np.deg2rad(120.0))],   # N
                      [0.0, 0.0],                                                                # C
                      [1.23 * np.cos(np.deg2rad(-120.0)), 1.23 * np.sin(np.deg2rad(-120.0))],  # O
                      [1.52, 0.0]])                                                              # CA
    am = local[None, :, 0, None] * arot[:, None, :, 0] + local[None, :, 1, None] * arot[:, None, :, 1]
    return (ac[:, None, :] + am).astype(np.float32)


def make_synthetic(n_uniform: int, seed: int = 3, density: float = 0.05, box=None,
                   origin=(0.0, 0.0, 0.0), n_rings: int = 0, n_amides: int = 0,
                   water_frac: float = 0.05, bond_radius: float = 1.7,
                   atoms_per_residue: int = 8, residues_per_chain: int = 300,
                   id: str = 'synthetic') -> PackedComplex:
    """Uniform random atoms (+ optional hexagonal rings and amide groups) in a box.

    box: (Lx, Ly, Lz) in Angstrom; default = cube with n_total / density volume.
    Config 3 = make_synthetic(100_000 - ring/amide atoms, seed=3, ...) with L=126.
    """
    n_ring_atoms, n_amide_atoms = 6 * n_rings, 4 * n_amides
    n = n_uniform + n_ring_atoms + n_amide_atoms
    if box is None:
        L = (max(n, 1) / density) ** (1.0 / 3.0)
        box = (L, L, L)
    box = np.asarray(box, dtype=np.float64)
    origin = np.asarray(origin, dtype=np.float64)
    idx = np.arange(n_uniform, dtype=np.uint64)

    # ---------------- uniform atoms ----------------
    xyz_u, is_water, elem_u, tm = _uniform_atoms(seed, idx, box, origin, water_frac)

    # residues of the uniform block: runs of `atoms_per_residue` non-water atoms, then waters
    prot = ~is_water
    ordinal = np.cumsum(prot) - 1
    n_prot_res = int((int(prot.sum()) + atoms_per_residue - 1) // atoms_per_residue)
    res_u = np.where(prot, ordinal // atoms_per_residue, n_prot_res + np.cumsum(is_water) - 1).astype(np.int64)
    n_wat = int(is_water.sum())
    nres_u = n_prot_res + n_wat

    # ---------------- rings: regular hexagons, radius 1.39; amides: N, C, O, CA planar ----------------
    xyz_r = _ring_atoms(seed, np.arange(n_rings, dtype=np.uint64), box, origin).reshape(-1, 3)
    xyz_a = _amide_atoms(seed, np.arange(n_amides, dtype=np.uint64), box, origin).reshape(-1, 3)

    # ---------------- assemble atoms ----------------
    xyz = np.concatenate([xyz_u, xyz_r, xyz_a], axis=0) if n else np.zeros((0, 3), np.float32)
    elem = np.concatenate([elem_u, np.full(n_ring_atoms, _EL_C), np.tile([_EL_N, _EL_C, _EL_O, _EL_C], n_amides)]).astype(np.int64)
    tmask = np.concatenate([
        tm,
        np.full(n_ring_atoms, T['aromatic'] | T['hydrophobe'], np.uint32),
        np.tile(np.array([T['hbond donor'],
                          T['carbonyl carbon'],
                          T['hbond acceptor'] | T['weak hbond acceptor'] | T['xbond acceptor'] | T['carbonyl oxygen'],
                          T['weak hbond donor']], np.uint32), n_amides),
    ]).astype(np.uint16)
    water_all = np.concatenate([is_water, np.zeros(n_ring_atoms + n_amide_atoms, bool)])
    res_id = np.concatenate([res_u,
                             nres_u + np.repeat(np.arange(n_rings), 6),
                             nres_u + n_rings + np.repeat(np.arange(n_amides), 4)]).astype(np.int32)
    nres = nres_u + n_rings + n_amides

    vdw_t = np.array([e[2] for e in _ELEMENTS])
    cov_t = np.array([e[3] for e in _ELEMENTS])
    vdw, cov = vdw_t[elem], cov_t[elem]

    flags = np.zeros(n, np.uint16)
    flags[elem == _EL_ZN] |= config.F_METAL
    flags[elem == _EL_CL] |= config.F_HALOGEN
    flags[water_all] |= config.F_WATER
    flags[elem == _EL_C] |= config.F_ELEM_C
    flags[elem == _EL_S] |= config.F_ELEM_S

    # residue table: protein-like residues are polypeptide with prev/next inside a chain
    res_flags = np.zeros(nres, np.uint8)
    res_prev = np.full(nres, -1, np.int32)
    res_next = np.full(nres, -1, np.int32)
    pr = np.arange(n_prot_res)
    res_flags[:n_prot_res] = config.R_POLYPEPTIDE | config.R_HAS_SEQ
    has_prev = (pr % residues_per_chain) != 0
    res_prev[:n_prot_res][has_prev] = pr[has_prev] - 1
    has_next = ((pr % residues_per_chain) != residues_per_chain - 1) & (pr + 1 < n_prot_res)
    res_next[:n_prot_res][has_next] = pr[has_next] + 1
    # 1 % of protein residues are "MET": mark their S atoms' residue flag
    met = u01(seed, 70, np.arange(n_prot_res, dtype=np.uint64)) < 0.05
    if n_uniform:
        is_met_atom = np.zeros(n, bool)
        is_met_atom[:n_uniform] = prot & met[np.minimum(res_u, max(n_prot_res - 1, 0))] if n_prot_res else False
        flags[is_met_atom] |= config.F_RES_MET

    # ---------------- bonds ----------------
    bp = []
    # ring bonds (aromatic) and amide bonds
    if n_rings:
        base = n_uniform + 6 * np.arange(n_rings)[:, None]
        k = np.arange(6)[None, :]
        bp.append(np.stack([(base + k).ravel(), (base + (k + 1) % 6).ravel()], axis=1))
    if n_amides:
        base = n_uniform + n_ring_atoms + 4 * np.arange(n_amides)
        for a_, b_ in ((0, 1), (1, 2), (1, 3)):
            bp.append(np.stack([base + a_, base + b_], axis=1))
    # proximity bonds between any two atoms closer than bond_radius (inter-residue ones
    # are what exercises the covalent branch, I:748-757)
    bp.append(_close_pairs(xyz, bond_radius))
    bp = np.concatenate(bp, axis=0) if bp else np.zeros((0, 2), np.int64)
    if bp.size:
        bp = np.unique(np.sort(bp, axis=1), axis=0)
    both = np.concatenate([bp, bp[:, ::-1]], axis=0)
    order = np.lexsort((both[:, 1], both[:, 0]))
    both = both[order]
    deg = np.bincount(both[:, 0], minlength=n) if n else np.zeros(0, np.int64)
    bond_off = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    bond_idx = both[:, 1].astype(np.int32)

    # single-bond heavy neighbour = first bonded atom (utils.py:612-635), -1 if none
    sb_nbr = np.full(n, -1, np.int32)
    has = deg > 0
    sb_nbr[has] = bond_idx[bond_off[:-1][has]]
    # xbond donors: bonded chlorines only (the reference dereferences None otherwise, U:173)
    tmask[(elem == _EL_CL) & has] |= T['xbond donor']

    # ---------------- hydrogens: donors / weak donors get 1-3 H at 1.0 A ----------------
    aall = np.arange(n, dtype=np.uint64)
    wants_h = (tmask & (T['hbond donor'] | T['weak hbond donor'])) != 0
    nh = np.where(wants_h, 1 + (u01(seed, 80, aall) * 3).astype(np.int64), 0)
    nh[water_all] = 2
    h_off = np.concatenate([[0], np.cumsum(nh)]).astype(np.int32)
    owner = np.repeat(np.arange(n), nh)
    slot = np.arange(int(nh.sum())) - np.repeat(h_off[:-1], nh)
    hdir = _unit_vectors(seed, 81, owner.astype(np.uint64) * np.uint64(4) + slot.astype(np.uint64))
    h_xyz = xyz[owner].astype(np.float64) + 1.0 * hdir if owner.size else np.zeros((0, 3))

    # ---------------- ring / amide geometry ----------------
    if n_rings:
        rp = xyz_r.astype(np.float64).reshape(n_rings, 6, 3)
        ring_center = rp.mean(axis=1)
        v = rp - ring_center[:, None, :]
        nrm = np.cross(v, np.roll(v, -1, axis=1)).sum(axis=1)
        ring_normal = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
    else:
        ring_center = np.zeros((0, 3)); ring_normal = np.zeros((0, 3))
    ring_res = (nres_u + np.arange(n_rings)).astype(np.int32)
    ring_atoms = [np.arange(n_uniform + 6 * r, n_uniform + 6 * r + 6, dtype=np.int32) for r in range(n_rings)]
    if n_amides:
        ap = xyz_a.reshape(n_amides, 4, 3)
        amide_center = ((ap[:, 1, :] + ap[:, 0, :]) / np.float32(2.0)).astype(np.float32)   # mean(C, N), float32
        nv = np.cross((ap[:, 2, :] - ap[:, 1, :]).astype(np.float64), (ap[:, 0, :] - ap[:, 1, :]).astype(np.float64))
        amide_normal = (nv / np.linalg.norm(nv, axis=1, keepdims=True)).astype(np.float32)
    else:
        amide_center = np.zeros((0, 3), np.float32); amide_normal = np.zeros((0, 3), np.float32)
    amide_res = (nres_u + n_rings + np.arange(n_amides)).astype(np.int32)
    amide_atoms = (n_uniform + n_ring_atoms + 4 * np.arange(n_amides)[:, None] + np.arange(4)[None, :]).astype(np.int32)

    # ---------------- labels ----------------
    sym = [e[0] for e in _ELEMENTS]
    element = [sym[e] for e in elem.tolist()]
    atom_name = ([f'{sym[e][0]}{i % 97}' for i, e in enumerate(elem_u.tolist())]
                 + [f'C{k + 1}' for _ in range(n_rings) for k in range(6)]
                 + ['N', 'C', 'O', 'CA'] * n_amides)
    res_name = (['MET' if m else 'ALA' for m in met.tolist()] + ['HOH'] * n_wat + ['BNZ'] * n_rings + ['AMD'] * n_amides)
    res_chain = ([chr(ord('A') + (r // residues_per_chain) % 26) for r in range(n_prot_res)]
                 + ['W'] * n_wat + ['R'] * n_rings + ['G'] * n_amides)
    res_seq = np.concatenate([pr % residues_per_chain + 1, np.arange(n_wat) + 1, np.arange(n_rings) + 1,
                              np.arange(n_amides) + 1]).astype(np.int32)
    comp = {'ALA': 'P', 'MET': 'P', 'HOH': 'W', 'BNZ': 'B', 'AMD': 'B'}

    return PackedComplex(
        xyz=xyz, vdw=vdw, cov=cov, type_mask=tmask, flags=flags, res_id=res_id,
        res_flags=res_flags, res_prev=res_prev, res_next=res_next,
        bond_off=bond_off, bond_idx=bond_idx, h_off=h_off, h_xyz=h_xyz, sb_nbr=sb_nbr,
        ring_center=ring_center, ring_normal=ring_normal, ring_res=ring_res, ring_atoms=ring_atoms,
        amide_center=amide_center, amide_normal=amide_normal, amide_res=amide_res, amide_atoms=amide_atoms,
        atom_name=atom_name, element=element, serial=np.arange(1, n + 1, dtype=np.int32),
        res_name=res_name, res_seq=res_seq, res_icode=[' '] * nres, res_chain=res_chain,
        component_types=comp, id=id)


def config3(n: int = 100_000, seed: int = 3) -> PackedComplex:
    """BASELINE.json configs[2]: 100k random-coordinate atoms, rho = 0.05 / A^3 (L = 126 A)."""
    n_rings, n_amides = n // 100, n // 50
    L = (n / 0.05) ** (1.0 / 3.0)
    return make_synthetic(n - 6 * n_rings - 4 * n_amides, seed=seed, box=(L, L, L), n_rings=n_rings,
                          n_amides=n_amides, id=f'synthetic_{n}')


def config5(n_rings: int = 10_000, n_amides: int = 10_000, seed: int = 5, L: float = 100.0) -> PackedComplex:
    """BASELINE.json configs[4]: 10k aromatic rings (+10k amide groups) in a 100 A cube."""
    return make_synthetic(0, seed=seed, box=(L, L, L), n_rings=n_rings, n_amides=n_amides, id='rings_10k')


def slab_box(n_total: int, box_slabs: int):
    """Box of the slab family: ``box_slabs`` cubes of n_total / box_slabs atoms (rho = 0.05 / A^3) side by side along x."""
    L = ((n_total / box_slabs) / 0.05) ** (1.0 / 3.0)
    return (L * box_slabs, L, L)


def slab_config(n_per_slab: int, n_slabs: int, seed: int = 4, box_slabs: int = None) -> PackedComplex:
    """BASELINE.json configs[3] family: n_slabs * n_per_slab atoms, box elongated along x
    so that each slab is the config-3 cube (weak scaling).  ``box_slabs``: the box is that of ``box_slabs`` cubes whatever
    ``n_slabs`` is — ONE structure (configs[3] itself: 2 M atoms, box_slabs = 8) cut into 1, 2, 4 or 8 slabs (strong scaling)."""
    n = n_per_slab * n_slabs
    n_rings, n_amides = n // 100, n // 50
    return make_synthetic(n - 6 * n_rings - 4 * n_amides, seed=seed, box=slab_box(n, box_slabs or n_slabs), n_rings=n_rings,
                          n_amides=n_amides, id=f'slab_{n_slabs}x{n_per_slab}')



def slab_home_records(n_per_slab: int, n_slabs: int, rank: int, seed: int = 4, bond_radius: float = 1.7,
                      water_frac: float = 0.05, atoms_per_residue: int = 8, residues_per_chain: int = 300, box_slabs: int = None):
    """What rank ``rank`` of ``n_slabs`` owns of ``slab_config(n_per_slab, n_slabs, seed)`` — the records ``sharding.pack_records``
    would cut out of the whole structure, bit for bit — WITHOUT building the whole structure: every per-atom quantity of the
    synthetic model is a function of (seed, index), so a rank evaluates them for its slab (plus the 1.7 A of neighbours its
    proximity bonds can reach) and computes for all atoms only the three cheap columns the partition needs (x, the water flag
    behind the residue numbering, ring / amide positions).  Returns (records, book) for ``sharding.shard_records_to_device``."""
    from . import sharding
    n = n_per_slab * n_slabs
    n_rings, n_amides = n // 100, n // 50
    n_uniform = n - 6 * n_rings - 4 * n_amides
    box = np.asarray(slab_box(n, box_slabs or n_slabs), np.float64)      # (box_slabs: see slab_config)
    origin = np.zeros(3)
    n_ring_atoms = 6 * n_rings
    # ---- columns of ALL atoms the partition and the residue numbering need
    idx_all = np.arange(n_uniform, dtype=np.uint64)
    x_u = (u01(seed, 10, idx_all) * box[0] + origin[0]).astype(np.float32)
    water_all = u01(seed, 20, idx_all) < water_frac
    prot_all = ~water_all
    n_prot = int(prot_all.sum())
    n_prot_res = (n_prot + atoms_per_residue - 1) // atoms_per_residue
    n_wat = int(water_all.sum())
    nres_u = n_prot_res + n_wat
    nres = nres_u + n_rings + n_amides
    res_u_all = np.where(prot_all, (np.cumsum(prot_all) - 1) // atoms_per_residue, n_prot_res + np.cumsum(water_all) - 1).astype(np.int64)
    xyz_r_all = _ring_atoms(seed, np.arange(n_rings, dtype=np.uint64), box, origin)        # [R, 6, 3] float32 (1 % of the atoms)
    xyz_a_all = _amide_atoms(seed, np.arange(n_amides, dtype=np.uint64), box, origin)      # [A, 4, 3]
    x_all = np.concatenate([x_u, xyz_r_all[:, :, 0].ravel(), xyz_a_all[:, :, 0].ravel()])
    edges = sharding.slab_edges(float(x_all.min()), float(x_all.max()), n_slabs)
    a_own = sharding.owner_of(x_all, edges)
    # rings / amides belong to the rank of their centre (float64 mean of the float32 atoms / float32 mean of C and N)
    ring_center_all = xyz_r_all.astype(np.float64).mean(axis=1) if n_rings else np.zeros((0, 3))
    amide_center_all = ((xyz_a_all[:, 1, :] + xyz_a_all[:, 0, :]) / np.float32(2.0)).astype(np.float32) if n_amides else np.zeros((0, 3), np.float32)
    r_own = sharding.owner_of(ring_center_all[:, 0], edges)
    m_own = sharding.owner_of(amide_center_all[:, 0], edges)
    home = np.nonzero(a_own == rank)[0]                       # global atom ids, ascending
    # ---- the margin set: home atoms and everything a proximity bond of a home atom can reach
    lo = float(x_all[home].min()) - bond_radius - 1e-3 if home.size else 0.0
    hi = float(x_all[home].max()) + bond_radius + 1e-3 if home.size else 0.0
    marg = np.nonzero((x_all >= lo) & (x_all <= hi))[0]
    mu = marg[marg < n_uniform]
    mr = marg[(marg >= n_uniform) & (marg < n_uniform + n_ring_atoms)] - n_uniform
    ma = marg[marg >= n_uniform + n_ring_atoms] - n_uniform - n_ring_atoms
    xyz_mu, water_mu, elem_mu, tm_mu = _uniform_atoms(seed, mu.astype(np.uint64), box, origin, water_frac)
    xyz_m = np.concatenate([xyz_mu, xyz_r_all.reshape(-1, 3)[mr], xyz_a_all.reshape(-1, 3)[ma]], axis=0)
    elem_m = np.concatenate([elem_mu, np.full(mr.size, _EL_C), np.array([_EL_N, _EL_C, _EL_O, _EL_C])[ma % 4]]).astype(np.int64)
    tmask_m = np.concatenate([tm_mu, np.full(mr.size, T['aromatic'] | T['hydrophobe'], np.uint32),
                              np.array([T['hbond donor'], T['carbonyl carbon'],
                                        T['hbond acceptor'] | T['weak hbond acceptor'] | T['xbond acceptor'] | T['carbonyl oxygen'],
                                        T['weak hbond donor']], np.uint32)[ma % 4]]).astype(np.uint16)
    water_m = np.concatenate([water_mu, np.zeros(mr.size + ma.size, bool)])
    res_m = np.concatenate([res_u_all[mu], nres_u + mr // 6, nres_u + n_rings + ma // 4]).astype(np.int64)
    # ---- bonds with at least one end in the margin set: ring / amide bonds by construction, proximity bonds by distance
    pos_of = np.full(n, -1, np.int64)
    pos_of[marg] = np.arange(marg.size)
    bp = []
    if n_rings:
        rings_here = np.unique(mr // 6)
        base = n_uniform + 6 * rings_here[:, None]
        k = np.arange(6)[None, :]
        bp.append(np.stack([(base + k).ravel(), (base + (k + 1) % 6).ravel()], axis=1))
    if n_amides:
        amides_here = np.unique(ma // 4)
        base = n_uniform + n_ring_atoms + 4 * amides_here
        for a_, b_ in ((0, 1), (1, 2), (1, 3)):
            bp.append(np.stack([base + a_, base + b_], axis=1))
    cp = _close_pairs(xyz_m, bond_radius)
    bp.append(marg[cp])
    bp = np.concatenate(bp, axis=0) if bp else np.zeros((0, 2), np.int64)
    if bp.size:
        bp = np.unique(np.sort(bp, axis=1), axis=0)
    both = np.concatenate([bp, bp[:, ::-1]], axis=0)
    is_home = np.zeros(n, bool)
    is_home[home] = True
    both = both[is_home[both[:, 0]]]                                   # the lists of home atoms only
    both = both[np.lexsort((both[:, 1], both[:, 0]))]
    hpos = pos_of[home]
    deg = np.bincount(np.searchsorted(home, both[:, 0]), minlength=home.size).astype(np.int32) if home.size else np.zeros(0, np.int32)
    bond_gid = both[:, 1].astype(np.int32)
    first = np.concatenate([[0], np.cumsum(deg)])[:-1]
    has = deg > 0
    sb_xyz = np.zeros((home.size, 3), np.float32)
    sb_xyz[has] = xyz_m[pos_of[bond_gid[first[has]]]]                  # first bonded atom = lowest id (utils.py:612-635)
    tmask_h = tmask_m[hpos].copy()
    elem_h = elem_m[hpos]
    tmask_h[(elem_h == _EL_CL) & has] |= T['xbond donor']
    # ---- hydrogens of the home atoms
    wants_h = (tmask_h & (T['hbond donor'] | T['weak hbond donor'])) != 0
    nh = np.where(wants_h, 1 + (u01(seed, 80, home.astype(np.uint64)) * 3).astype(np.int64), 0)
    nh[water_m[hpos]] = 2
    owner = np.repeat(home, nh)
    slot = np.arange(int(nh.sum())) - np.repeat(np.concatenate([[0], np.cumsum(nh)])[:-1], nh)
    hdir = _unit_vectors(seed, 81, owner.astype(np.uint64) * np.uint64(4) + slot.astype(np.uint64))
    h_xyz = xyz_m[pos_of[owner]].astype(np.float64) + 1.0 * hdir if owner.size else np.zeros((0, 3))
    # ---- flags, radii, residue rows
    vdw_t = np.array([e[2] for e in _ELEMENTS])
    cov_t = np.array([e[3] for e in _ELEMENTS])
    flags = np.zeros(home.size, np.uint16)
    flags[elem_h == _EL_ZN] |= config.F_METAL
    flags[elem_h == _EL_CL] |= config.F_HALOGEN
    flags[water_m[hpos]] |= config.F_WATER
    flags[elem_h == _EL_C] |= config.F_ELEM_C
    flags[elem_h == _EL_S] |= config.F_ELEM_S
    res_h = res_m[hpos]
    uni_prot = (home < n_uniform) & ~water_m[hpos]
    if n_prot_res:
        met_h = u01(seed, 70, np.minimum(res_h, n_prot_res - 1).astype(np.uint64)) < 0.05
        flags[uni_prot & met_h] |= config.F_RES_MET
    is_prot_res = res_h < n_prot_res
    res_flags = np.where(is_prot_res, config.R_POLYPEPTIDE | config.R_HAS_SEQ, 0).astype(np.uint8)
    res_prev = np.where(is_prot_res & (res_h % residues_per_chain != 0), res_h - 1, -1).astype(np.int32)
    res_next = np.where(is_prot_res & (res_h % residues_per_chain != residues_per_chain - 1) & (res_h + 1 < n_prot_res), res_h + 1, -1).astype(np.int32)
    # ---- rings / amides of this rank
    rh, mh = np.nonzero(r_own == rank)[0], np.nonzero(m_own == rank)[0]
    rp = xyz_r_all[rh].astype(np.float64)
    if rh.size:
        v = rp - ring_center_all[rh][:, None, :]
        nrm = np.cross(v, np.roll(v, -1, axis=1)).sum(axis=1)
        ring_normal = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
    else:
        ring_normal = np.zeros((0, 3))
    ap = xyz_a_all[mh]
    if mh.size:
        nv = np.cross((ap[:, 2, :] - ap[:, 1, :]).astype(np.float64), (ap[:, 0, :] - ap[:, 1, :]).astype(np.float64))
        amide_normal = (nv / np.linalg.norm(nv, axis=1, keepdims=True)).astype(np.float32)
    else:
        amide_normal = np.zeros((0, 3), np.float32)
    rec = {
        'gid': home.astype(np.int32), 'xyz': xyz_m[hpos], 'vdw': vdw_t[elem_h], 'cov': cov_t[elem_h], 'tmask': tmask_h, 'flags': flags,
        'res_gid': res_h.astype(np.int32), 'res_flags': res_flags, 'res_prev': res_prev, 'res_next': res_next,
        'sel': np.ones(home.size, np.uint8), 'sb_xyz': sb_xyz, 'sb_has': has.astype(np.uint8), 'h_cnt': nh.astype(np.int32), 'h_xyz': h_xyz,
        'bond_cnt': deg, 'bond_gid': bond_gid,
        'ring_gid': rh.astype(np.int32), 'ring_center': ring_center_all[rh], 'ring_normal': ring_normal, 'ring_res': (nres_u + rh).astype(np.int32),
        'amide_gid': mh.astype(np.int32), 'amide_center': amide_center_all[mh], 'amide_normal': amide_normal,
        'amide_res': (nres_u + n_rings + mh).astype(np.int32),
    }
    book = dict(edges=edges, n_res_global=int(nres), n_atoms_global=int(n), home_x=x_all[home].astype(np.float64))
    return rec, book


def proteinlike(n_res: int = 480, seed: int = 2, n_waters: int = 300, id: str = 'proteinlike') -> PackedComplex:
    """Stand-in for BASELINE configs[0]/[1] (`1tqn_h.cif`, ~7.7 k atoms incl. hydrogens; the file itself is not
    available): one polypeptide chain 'A' laid out as a compact self-avoiding walk, explicit hydrogens present BOTH as
    atoms (element H, like a hydrogenated mmCIF) and as `h_coords` of their parents, backbone + side-chain bonds,
    phenyl rings on ~8 % of the residues, one amide per peptide bond, a haem-like ligand residue 508 (metal + four
    five-membered rings), and waters (chain 'A', het flag W).  Deterministic (splitmix64)."""
    rs = np.random.RandomState(splitmix64(seed, [0])[0] % (2 ** 31))   # only for the walk; everything is seeded
    atoms = []      # dict(xyz, el, name, res, tmask, flags, hpar)
    bonds = []
    residues = []   # dict(name, seq, chain, poly)
    rings = []      # (atom indices, residue)
    amides = []     # (N, C, O, CA) atom indices, residue

    def unit(v):
        return v / np.linalg.norm(v)

    def add_atom(xyz, el, name, res, tm=0, fl=0):
        atoms.append(dict(xyz=np.asarray(xyz, float), el=el, name=name, res=res, tm=tm, fl=fl, h=[]))
        return len(atoms) - 1

    def add_h(parent, direction, name):
        p = atoms[parent]
        hx = p['xyz'] + 1.0 * unit(direction)
        i = add_atom(hx, 'H', name, p['res'], 0, config.F_HYDROGEN)
        p['h'].append(i)
        bonds.append((parent, i))
        return i

    # compact walk of CA positions: steps of 3.8 A biased towards the centroid
    ca = [np.zeros(3)]
    for k in range(1, n_res):
        for _ in range(200):
            d = unit(rs.normal(size=3))
            cand = ca[-1] + 3.8 * d - 0.02 * (ca[-1] - np.mean(ca, axis=0))
            cand = ca[-1] + 3.8 * unit(cand - ca[-1])
            if all(np.linalg.norm(cand - c) > 4.2 for c in ca[-40:-1]) and np.linalg.norm(cand - np.mean(ca, axis=0)) < 4.0 * n_res ** (1 / 3) + 6:
                break
        ca.append(cand)
    ca = np.array(ca)
    prev_c = None
    for k in range(n_res):
        aromatic = (splitmix64(seed, [1000 + k])[0] % 100) < 8
        charged = (splitmix64(seed, [2000 + k])[0] % 100)
        rname = 'PHE' if aromatic else ('MET' if charged < 4 else ('LYS' if charged < 12 else ('ASP' if charged < 20 else 'ALA')))
        residues.append(dict(name=rname, seq=k + 1, chain='A', poly=True))
        fwd = unit(ca[min(k + 1, n_res - 1)] - ca[max(k - 1, 0)] + 1e-3)
        side = unit(np.cross(fwd, rs.normal(size=3)))
        up = np.cross(fwd, side)
        met = config.F_RES_MET if rname == 'MET' else 0
        n_ = add_atom(ca[k] - 1.2 * fwd + 0.5 * side, 'N', 'N', k, T['hbond donor'], met)
        a_ = add_atom(ca[k], 'C', 'CA', k, T['weak hbond donor'], config.F_ELEM_C | met)
        c_ = add_atom(ca[k] + 1.25 * fwd + 0.55 * side, 'C', 'C', k, T['carbonyl carbon'], config.F_ELEM_C | met)
        o_ = add_atom(atoms[c_]['xyz'] + 1.23 * unit(side + 0.3 * up), 'O', 'O', k,
                      T['hbond acceptor'] | T['weak hbond acceptor'] | T['xbond acceptor'] | T['carbonyl oxygen'], met)
        bonds.extend([(n_, a_), (a_, c_), (c_, o_)])
        if prev_c is not None:
            bonds.append((prev_c, n_))
            amides.append(((n_, prev_c, prev_o, prev_ca), k - 1))
        add_h(n_, -fwd + up, 'H')
        add_h(a_, up - side, 'HA')
        cb = add_atom(ca[k] - 1.5 * unit(side - 0.4 * up), 'C', 'CB', k, T['hydrophobe'] | T['weak hbond donor'], config.F_ELEM_C | met)
        bonds.append((a_, cb))
        add_h(cb, up, 'HB2'); add_h(cb, -fwd, 'HB3')
        out = unit(atoms[cb]['xyz'] - ca[k])
        if aromatic:      # phenyl ring hanging off CB
            e1, e2 = out, unit(np.cross(out, up))
            cen = atoms[cb]['xyz'] + 2.9 * out
            ids = []
            for q in range(6):
                ang = q * np.pi / 3
                ids.append(add_atom(cen + 1.39 * (np.cos(ang) * -e1 + np.sin(ang) * e2), 'C', ('CG', 'CD1', 'CE1', 'CZ', 'CE2', 'CD2')[q], k,
                                    T['aromatic'] | T['hydrophobe'], config.F_ELEM_C))
            for q in range(6):
                bonds.append((ids[q], ids[(q + 1) % 6]))
                if q:
                    add_h(ids[q], atoms[ids[q]]['xyz'] - cen, 'H' + atoms[ids[q]]['name'][1:])
            bonds.append((cb, ids[0]))
            rings.append((ids, k))
        elif rname == 'MET':
            sd = add_atom(atoms[cb]['xyz'] + 1.8 * out, 'S', 'SD', k, T['hbond acceptor'] | T['hydrophobe'], config.F_ELEM_S | met)
            ce = add_atom(atoms[sd]['xyz'] + 1.8 * unit(out + up), 'C', 'CE', k, T['hydrophobe'] | T['weak hbond donor'], config.F_ELEM_C | met)
            bonds.extend([(cb, sd), (sd, ce)])
            add_h(ce, up, 'HE1'); add_h(ce, out, 'HE2')
        elif rname == 'LYS':
            nz = add_atom(atoms[cb]['xyz'] + 1.5 * out, 'N', 'NZ', k, T['hbond donor'] | T['pos ionisable'], 0)
            bonds.append((cb, nz))
            add_h(nz, out, 'HZ1'); add_h(nz, up, 'HZ2'); add_h(nz, -up, 'HZ3')
        elif rname == 'ASP':
            cg = add_atom(atoms[cb]['xyz'] + 1.5 * out, 'C', 'CG', k, T['carbonyl carbon'], config.F_ELEM_C)
            o1 = add_atom(atoms[cg]['xyz'] + 1.25 * unit(out + up), 'O', 'OD1', k, T['hbond acceptor'] | T['neg ionisable'] | T['carbonyl oxygen'] | T['weak hbond acceptor'], 0)
            o2 = add_atom(atoms[cg]['xyz'] + 1.25 * unit(out - up), 'O', 'OD2', k, T['hbond acceptor'] | T['neg ionisable'] | T['weak hbond acceptor'], 0)
            bonds.extend([(cb, cg), (cg, o1), (cg, o2)])
        prev_c, prev_o, prev_ca = c_, o_, a_
    # haem-like ligand, residue 508: metal + four pyrrole-like rings + a chlorinated tail, placed beside the chain centre
    lig = len(residues)
    residues.append(dict(name='HEM', seq=508, chain='A', poly=False))
    c0 = np.mean(ca, axis=0) + np.array([0.0, 0.0, 1.5])
    fe = add_atom(c0, 'FE', 'FE', lig, 0, config.F_METAL)
    for q in range(4):
        ang = q * np.pi / 2
        cen = c0 + 3.0 * np.array([np.cos(ang), np.sin(ang), 0.0])
        ids = []
        for w in range(5):
            a2 = ang + np.pi + w * 2 * np.pi / 5
            el, nm = ('N', 'N' + 'ABCD'[q]) if w == 0 else ('C', f'C{w}' + 'ABCD'[q])
            tm = (T['aromatic'] | T['hbond acceptor']) if w == 0 else (T['aromatic'] | T['hydrophobe'])
            ids.append(add_atom(cen + 1.15 * np.array([np.cos(a2), np.sin(a2), 0.0]), el, nm, lig, tm, config.F_ELEM_C if w else 0))
        for w in range(5):
            bonds.append((ids[w], ids[(w + 1) % 5]))
        bonds.append((fe, ids[0]))
        rings.append((ids, lig))
    cl_c = add_atom(c0 + np.array([0, 0, -2.0]), 'C', 'CBB', lig, T['hydrophobe'] | T['weak hbond donor'], config.F_ELEM_C)
    cl = add_atom(c0 + np.array([0, 0, -3.75]), 'CL', 'CL1', lig, T['xbond donor'] | T['weak hbond acceptor'] | T['hydrophobe'], config.F_HALOGEN)
    bonds.extend([(fe, cl_c), (cl_c, cl)])
    # waters around the structure
    wi = np.arange(n_waters, dtype=np.uint64)
    span = ca.max(axis=0) - ca.min(axis=0) + 8.0
    for k in range(n_waters):
        r = len(residues)
        residues.append(dict(name='HOH', seq=600 + k, chain='A', poly=False))
        pos = ca.min(axis=0) - 4.0 + span * np.array([u01(seed, 90, wi[k:k + 1])[0], u01(seed, 91, wi[k:k + 1])[0], u01(seed, 92, wi[k:k + 1])[0]])
        o = add_atom(pos, 'O', 'O', r, T['hbond acceptor'] | T['hbond donor'], config.F_WATER)
        d1 = _unit_vectors(seed, 93, wi[k:k + 1])[0]
        d2 = unit(np.cross(d1, [0.3, 0.5, 0.8]) - 0.3 * d1)
        h1 = add_h(o, d1, 'H1'); h2 = add_h(o, d2, 'H2')
        atoms[h1]['fl'] |= config.F_WATER; atoms[h2]['fl'] |= config.F_WATER

    n = len(atoms)
    xyz = np.array([a['xyz'] for a in atoms]).astype(np.float32)
    el = [a['el'] for a in atoms]
    vdw_t = {'C': 1.7, 'N': 1.55, 'O': 1.52, 'S': 1.8, 'H': 1.1, 'FE': 2.05, 'CL': 1.75}
    cov_t = {'C': 0.76, 'N': 0.71, 'O': 0.66, 'S': 1.05, 'H': 0.31, 'FE': 1.32, 'CL': 1.02}
    adj = [[] for _ in range(n)]
    for i, j in bonds:
        adj[i].append(j); adj[j].append(i)
    bond_off = np.concatenate([[0], np.cumsum([len(a) for a in adj])]).astype(np.int32)
    bond_idx = np.array([j for a in adj for j in a], np.int32)
    # single-bond heavy neighbour: first bonded non-hydrogen atom outside an aromatic ring bond
    ring_bonds = {frozenset((ids[q], ids[(q + 1) % len(ids)])) for ids, _ in rings for q in range(len(ids))}
    sb = np.full(n, -1, np.int32)
    for i in range(n):
        for j in adj[i]:
            if el[j] != 'H' and frozenset((i, j)) not in ring_bonds:
                sb[i] = j
                break
    h_lists = [[xyz[h].astype(np.float64) for h in a['h']] for a in atoms]    # hydrogens as read from the (float32) atoms
    h_off = np.concatenate([[0], np.cumsum([len(h) for h in h_lists])]).astype(np.int32)
    h_xyz = np.array([h for hs in h_lists for h in hs]).reshape(-1, 3)
    nres = len(residues)
    res_flags = np.array([(config.R_POLYPEPTIDE | config.R_HAS_SEQ) if r['poly'] else 0 for r in residues], np.uint8)
    res_prev = np.array([k - 1 if (r['poly'] and k > 0) else -1 for k, r in enumerate(residues)], np.int32)
    res_next = np.array([k + 1 if (r['poly'] and k + 1 < n_res) else -1 for k, r in enumerate(residues)], np.int32)
    rp = np.array([[xyz[i].astype(np.float64) for i in ids] for ids, _ in rings if len(ids) == 6]).reshape(-1, 6, 3)
    ring_center, ring_normal = [], []
    for ids, _ in rings:
        pts = xyz[ids].astype(np.float64)
        cen = pts.mean(axis=0)
        v = pts - cen
        nv = np.cross(v, np.roll(v, -1, axis=0)).sum(axis=0)
        ring_center.append(cen); ring_normal.append(nv / np.linalg.norm(nv))
    am_c, am_n = [], []
    for (n_i, c_i, o_i, ca_i), _ in amides:
        am_c.append(((xyz[c_i] + xyz[n_i]) / np.float32(2.0)).astype(np.float32))
        nv = np.cross((xyz[o_i] - xyz[c_i]).astype(np.float64), (xyz[n_i] - xyz[c_i]).astype(np.float64))
        am_n.append((nv / np.linalg.norm(nv)).astype(np.float32))
    del rp
    comp = {'ALA': 'P', 'PHE': 'P', 'MET': 'P', 'LYS': 'P', 'ASP': 'P', 'HEM': 'B', 'HOH': 'W'}
    return PackedComplex(
        xyz=xyz, vdw=[vdw_t[e] for e in el], cov=[cov_t[e] for e in el], type_mask=[a['tm'] for a in atoms],
        flags=[a['fl'] for a in atoms], res_id=[a['res'] for a in atoms], res_flags=res_flags, res_prev=res_prev, res_next=res_next,
        bond_off=bond_off, bond_idx=bond_idx, h_off=h_off, h_xyz=h_xyz, sb_nbr=sb,
        ring_center=np.array(ring_center).reshape(-1, 3), ring_normal=np.array(ring_normal).reshape(-1, 3),
        ring_res=[r for _, r in rings], ring_atoms=[np.array(ids, np.int32) for ids, _ in rings],
        amide_center=np.array(am_c, np.float32).reshape(-1, 3), amide_normal=np.array(am_n, np.float32).reshape(-1, 3),
        amide_res=[r for _, r in amides], amide_atoms=np.array([list(t) for t, _ in amides], np.int32).reshape(-1, 4),
        atom_name=[a['name'] for a in atoms], element=el, serial=np.arange(1, n + 1, dtype=np.int32),
        res_name=[r['name'] for r in residues], res_seq=[r['seq'] for r in residues], res_icode=[' '] * nres,
        res_chain=[r['chain'] for r in residues], component_types=comp, id=id)
