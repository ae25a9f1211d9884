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
    """Angles equal within tol degrees (the north-star bound), NaN == NaN.  float32 angles (the amide-amide loop computes
    in float32, I:1227-1300) get a tighter, derived bound: both sides run the same operation sequence on the same float32
    cosine and differ only by acosf (device library vs glibc, <= 2 ulp of a value <= pi: 4.8e-7 rad = 2.7e-5 degrees)
    carried through the degrees conversion rad * 180 / pi (two roundings of a value <= 180: 2 x 7.6e-6), i.e. 4.3e-5
    degrees at most; 6e-5 is asserted."""
    fa, fb = np.asarray(a), np.asarray(b)
    if fa.dtype == np.float32 and fb.dtype == np.float32:
        tol = min(tol, 6e-5)
    a, b = fa.astype(np.float64), fb.astype(np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    return bool(np.all(both_nan | (np.abs(a - b) <= tol)))


def known_answer_packs():
    """Tiny packs that isolate one branch each of interactions.py:693-936 (used on CPU and GPU)."""
    from arpeggio_amd.core import config
    T = config.ATOM_TYPE_BIT
    P = config.R_POLYPEPTIDE | config.R_HAS_SEQ
    da = T['hbond acceptor'] | T['hbond donor']
    acc_wd = T['hbond acceptor'] | T['weak hbond donor']
    xd, xa = T['xbond donor'], T['xbond acceptor']
    c, s = np.cos, np.sin
    r = np.deg2rad
    packs = []
    for d in (1.0, 1.52, 3.39, 3.4, 3.5, 3.51, 4.5, 4.9, 5.0):
        packs.append((f'ladder_{d}', tiny_complex([[0, 0, 0], [d, 0, 0]], type_mask=T['hydrophobe'] | T['aromatic'])))
    packs.append(('covalent', tiny_complex([[0, 0, 0], [1.0, 0, 0]], bonds=[(0, 1)], type_mask=T['hydrophobe'])))
    packs.append(('metal', tiny_complex([[0, 0, 0], [1.0, 0, 0], [2.8, 0.1, 0]], type_mask=[T['hbond acceptor'], 0, T['hbond acceptor']],
                                        flags=[0, config.F_METAL, 0])))
    packs.append(('hydrogen_atom', tiny_complex([[0, 0, 0], [3.0, 0, 0], [0, 3, 0]], flags=[config.F_HYDROGEN, 0, 0])))
    packs.append(('same_residue', tiny_complex([[0, 0, 0], [3.0, 0, 0], [0, 3, 0]], res_id=[0, 0, 1])))
    packs.append(('seq_adjacent', tiny_complex([[0, 0, 0], [3.0, 0, 0], [0, 3, 0]], res_id=[0, 1, 2], res_flags=[P, P, 0],
                                               res_prev=[-1, 0, -1], res_next=[1, -1, -1])))
    packs.append(('seq_adjacent_end_not_poly', tiny_complex([[0, 0, 0], [3.0, 0, 0]], res_id=[0, 1], res_flags=[P, config.R_HAS_SEQ],
                                                            res_prev=[-1, 0], res_next=[1, -1])))
    packs.append(('seq_adjacent_bgn_not_poly', tiny_complex([[0, 0, 0], [3.0, 0, 0]], res_id=[0, 1], res_flags=[config.R_HAS_SEQ, P],
                                                            res_prev=[-1, 0], res_next=[1, -1])))
    packs.append(('water_near', tiny_complex([[0, 0, 0], [3.4, 0, 0]], type_mask=[da, T['hbond acceptor']], flags=[config.F_WATER, 0])))
    packs.append(('water_far', tiny_complex([[0, 0, 0], [3.6, 0, 0]], type_mask=[T['hbond acceptor'], da], flags=[0, config.F_WATER])))
    packs.append(('water_water', tiny_complex([[0, 0, 0], [2.8, 0, 0]], type_mask=da, flags=config.F_WATER,
                                              h={0: [[0.96, 0, 0], [-0.3, 0.9, 0]], 1: [[3.5, 0.5, 0], [3.0, -0.9, 0]]})))
    for ang in (60, 80, 95, 180):
        packs.append((f'hbond_{ang}', tiny_complex([[0, 0, 0], [2.9, 0, 0]], type_mask=[da, da], h={0: [[c(r(ang)), s(r(ang)), 0]]})))
        packs.append((f'hbond_rev_{ang}', tiny_complex([[2.9, 0, 0], [0, 0, 0]], type_mask=[da, da], h={1: [[c(r(ang)), s(r(ang)), 0]]})))
    packs.append(('weak_overwrite', tiny_complex([[0, 0, 0], [3.0, 0, 0]], type_mask=[acc_wd, acc_wd], h={1: [[2.0, 0, 0]]})))
    packs.append(('weak_single', tiny_complex([[0, 0, 0], [3.0, 0, 0]], type_mask=[T['hbond acceptor'], T['weak hbond donor']], h={1: [[2.0, 0, 0]]})))
    packs.append(('weak_single_rev', tiny_complex([[3.0, 0, 0], [0, 0, 0]], type_mask=[T['weak hbond donor'], T['hbond acceptor']], h={0: [[2.0, 0, 0]]})))
    for ang in (100, 119, 121, 180):
        packs.append((f'xbond_{ang}', tiny_complex([[1.7 * c(r(ang)), 1.7 * s(r(ang)), 0], [0, 0, 0], [3.3, 0, 0]],
                                                   type_mask=[0, xd, xa], bonds=[(0, 1)], res_id=[0, 0, 1])))
        packs.append((f'xbond_rev_{ang}', tiny_complex([[3.3, 0, 0], [0, 0, 0], [1.7 * c(r(ang)), 1.7 * s(r(ang)), 0]],
                                                       type_mask=[xa, xd, 0], bonds=[(1, 2)], res_id=[1, 0, 0])))
    # halogen weak hydrogen bond: C-Cl ... H-N, both orientations
    for ang in (20, 40, 90, 149, 160):
        hx = [3.0 * c(r(ang)) * 0 + 2.2, 0, 0]
        nb = [-1.7 * c(r(ang)), 1.7 * s(r(ang)), 0]
        packs.append((f'halogen_weak_{ang}', tiny_complex([nb, [0, 0, 0], [3.2, 0, 0]],
                                                          type_mask=[0, T['weak hbond acceptor'], T['hbond donor']],
                                                          flags=[0, config.F_HALOGEN, 0], bonds=[(0, 1)], res_id=[0, 0, 1],
                                                          h={2: [hx]})))
        packs.append((f'halogen_weak_rev_{ang}', tiny_complex([[3.2, 0, 0], [0, 0, 0], nb],
                                                              type_mask=[T['weak hbond donor'], T['weak hbond acceptor'], 0],
                                                              flags=[0, config.F_HALOGEN, 0], bonds=[(1, 2)], res_id=[1, 0, 0],
                                                              h={0: [hx]})))
    packs.append(('halogen_no_neighbour', tiny_complex([[0, 0, 0], [3.2, 0, 0]], type_mask=[T['weak hbond acceptor'], T['hbond donor']],
                                                       flags=[config.F_HALOGEN, 0], h={1: [[2.2, 0, 0]]}, sb_nbr=[-1, -1])))
    for d in (3.6, 3.61, 4.0, 4.01, 4.5, 4.51):
        packs.append((f'features_{d}', tiny_complex([[0, 0, 0], [d, 0, 0], [0, d, 0]],
                                                    type_mask=[T['pos ionisable'] | T['carbonyl oxygen'] | T['aromatic'] | T['hydrophobe'],
                                                               T['neg ionisable'] | T['carbonyl carbon'] | T['aromatic'] | T['hydrophobe'],
                                                               T['neg ionisable'] | T['carbonyl carbon']])))
    packs.append(('collinear_angle_nan', tiny_complex([[0, 0, 0], [3, 0, 0]], type_mask=[T['hbond donor'], T['hbond acceptor']],
                                                      h={0: [[0, 0, 0]]})))   # hydrogen on top of the donor: NaN -> pi
    return packs


def random_dense_pack(seed, n=400, box=14.0):
    """Dense random soup with every atom type / flag and random bonds, hydrogens and neighbours."""
    from arpeggio_amd.core import config
    rng = np.random.default_rng(seed)
    xyz = (rng.random((n, 3)) * box).astype(np.float32)
    tm = np.zeros(n, np.uint16)
    for b in range(12):
        tm |= ((rng.random(n) < 0.3).astype(np.uint16) << b)
    fl = np.zeros(n, np.uint16)
    fl |= (rng.random(n) < 0.05).astype(np.uint16) * config.F_METAL
    fl |= (rng.random(n) < 0.10).astype(np.uint16) * config.F_HALOGEN
    fl |= (rng.random(n) < 0.10).astype(np.uint16) * config.F_WATER
    fl |= (rng.random(n) < 0.05).astype(np.uint16) * config.F_HYDROGEN
    fl |= (rng.random(n) < 0.4).astype(np.uint16) * config.F_ELEM_C
    fl |= (rng.random(n) < 0.1).astype(np.uint16) * config.F_ELEM_S
    fl |= (rng.random(n) < 0.2).astype(np.uint16) * config.F_RES_MET
    res_id = np.sort(rng.integers(0, n // 3, n)).astype(np.int32)
    res_id = np.unique(res_id, return_inverse=True)[1].astype(np.int32)
    nres = int(res_id.max()) + 1
    poly = rng.random(nres) < 0.7
    res_flags = np.where(poly, config.R_POLYPEPTIDE | config.R_HAS_SEQ, np.where(rng.random(nres) < 0.3, config.R_HAS_SEQ, 0)).astype(np.uint8)
    res_prev = np.where(np.arange(nres) > 0, np.arange(nres) - 1, -1).astype(np.int32)
    res_next = np.where(np.arange(nres) < nres - 1, np.arange(nres) + 1, -1).astype(np.int32)
    res_prev[rng.random(nres) < 0.2] = -1
    res_next[rng.random(nres) < 0.2] = -1
    d = np.linalg.norm(xyz[:, None, :].astype(np.float64) - xyz[None, :, :].astype(np.float64), axis=2)
    bi, bj = np.nonzero(np.triu(d < 1.9, 1))
    keep = rng.random(len(bi)) < 0.7
    bonds = list(zip(bi[keep].tolist(), bj[keep].tolist()))
    h = {}
    for i in range(n):
        k = int(rng.integers(0, 4)) if rng.random() < 0.5 else 0
        if k:
            v = rng.standard_normal((k, 3))
            v /= np.linalg.norm(v, axis=1, keepdims=True)
            h[i] = (xyz[i].astype(np.float64) + v * rng.uniform(0.9, 1.1)).tolist()
    pc = tiny_complex(xyz, vdw=rng.choice([1.2, 1.52, 1.55, 1.7, 1.8, 1.75, 1.39], n), cov=rng.choice([0.31, 0.66, 0.71, 0.76, 1.05, 1.02, 1.22], n),
                      type_mask=tm, flags=fl, res_id=res_id, res_flags=res_flags, res_prev=res_prev, res_next=res_next, bonds=bonds, h=h)
    # xbond donors without a neighbour would make the reference raise: give every xbond donor a neighbour
    xd = (pc.type_mask & config.ATOM_TYPE_BIT['xbond donor']) != 0
    lonely = xd & (pc.sb_nbr < 0)
    pc.sb_nbr[lonely] = (np.nonzero(lonely)[0] + 1) % n
    return pc


def boundary_sensitive_pairs(pc, contacts, comp=0.1):
    """how many emitted contacts sit within one float32 ulp of a threshold the per-pair code compares the
    float32 distance with (sum of covalent radii, sum of vdW radii, vdW sum + comp, 2.8, 3.5, 3.6, 4.0, 4.5: I:760-920;
    compared in float32, NEP 50).  These are the pairs on which a one-bit difference in the distance would flip a flag;
    the parity tests compare them like every other pair (bit-identical), this only counts them."""
    i, j, d = np.asarray(contacts['i']), np.asarray(contacts['j']), np.asarray(contacts['dist'], np.float32)
    thr = [np.asarray(pc.cov[i] + pc.cov[j]).astype(np.float32), np.asarray(pc.vdw[i] + pc.vdw[j]).astype(np.float32),
           np.asarray(pc.vdw[i] + pc.vdw[j] + comp).astype(np.float32)]
    thr += [np.full(len(d), np.float32(t), np.float32) for t in (2.8, 3.5, 3.6, 4.0, 4.5)]
    near = np.zeros(len(d), bool)
    exact = np.zeros(len(d), bool)
    for t in thr:
        near |= (d >= np.nextafter(t, np.float32(-np.inf))) & (d <= np.nextafter(t, np.float32(np.inf)))
        exact |= d == t
    return dict(contacts=int(len(d)), within_1ulp=int(near.sum()), exactly_on_threshold=int(exact.sum()))


def report_boundary_pairs(name, stats):
    """Append to gpurun_out/boundary_pairs.json (copied into profiles/ for DESIGN.md)."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, 'gpurun_out', 'boundary_pairs.json')
    os.makedirs(os.path.dirname(path), exist_ok=True)
    try:
        data = json.load(open(path))
    except (OSError, ValueError):
        data = {}
    data[name] = stats
    json.dump(data, open(path, 'w'), indent=1, sort_keys=True)


def threshold_edge_pack(comp=0.1):
    """Isolated pairs placed ON every distance threshold of I:745-921 and one float32 ulp to either side of it: atom A at
    (0, 20 k, 20 m), atom B at (d, 20 k, 20 m), so that the float32 difference is d itself and nothing else is within reach.
    Each atom is its own residue.  The thresholds: sum of covalent radii, sum of van-der-Waals radii, that + vdw_comp
    (compared as float32, I:756-773), 2.8 (metal), 3.5 (polar / weak polar), 3.6 (carbonyl), 4.0 (ionic, aromatic), 4.5
    (feature gate, hydrophobic), and the search radius 5.0 (float64 d^2 <= 25 on the float32 coordinates)."""
    from arpeggio_amd.core import config
    T = config.ATOM_TYPE_BIT
    vdw, cov = 1.7, 0.76
    thresholds = [np.float32(cov + cov), np.float32(vdw + vdw), np.float32(vdw + vdw + comp), np.float32(2.8), np.float32(3.5),
                  np.float32(3.6), np.float32(4.0), np.float32(4.5), np.float32(5.0)]
    both = 0
    for k in ('hbond acceptor', 'hbond donor', 'weak hbond acceptor', 'weak hbond donor', 'pos ionisable', 'neg ionisable',
              'hydrophobe', 'carbonyl oxygen', 'carbonyl carbon', 'aromatic', 'xbond acceptor'):
        both |= T[k]
    combos = [(both, 0, both, 0),                                           # every type on both sides (no hydrogens: hbond by geometry stays 0)
              (T['hbond acceptor'], 0, 0, config.F_METAL),                  # metal complex at 2.8
              (T['hbond acceptor'] | T['hbond donor'], config.F_WATER, T['hbond acceptor'], 0),     # water branch of the hbond block
              (T['carbonyl oxygen'] | T['neg ionisable'], 0, T['carbonyl carbon'] | T['pos ionisable'], 0)]
    xyz, tm, fl = [], [], []
    slot = 0
    for t in thresholds:
        for d in (np.nextafter(t, np.float32(0)), t, np.nextafter(t, np.float32(10))):
            for tb, fb, te, fe in combos:
                y, z = 20.0 * (slot % 12), 20.0 * (slot // 12)
                slot += 1
                xyz += [[0.0, y, z], [float(d), y, z]]
                tm += [tb, te]
                fl += [fb, fe]
    # three more pairs around the 6.0 A expansion radius of _make_selection (I:1420: float64 d^2 <= 36 on float32 coordinates);
    # pc.expansion_probe = the six atoms (A, B per pair): selecting the A atoms decides which B atoms join selection_plus
    probe = []
    for d in (np.nextafter(np.float32(6.0), np.float32(0)), np.float32(6.0), np.nextafter(np.float32(6.0), np.float32(10))):
        y, z = 20.0 * (slot % 12), 20.0 * (slot // 12)
        slot += 1
        probe += [len(xyz), len(xyz) + 1]
        xyz += [[0.0, y, z], [float(d), y, z]]
        tm += [0, 0]
        fl += [0, 0]
    pc = tiny_complex(np.array(xyz, np.float32), vdw=vdw, cov=cov, type_mask=tm, flags=fl)
    pc.expansion_probe = np.array(probe, np.int32)
    return pc


def concat_packs(p, q, shift=(0.0, 0.0, 0.0)):
    """Two packed structures side by side as one (q moved by `shift`); labels are dropped.  For shard tests that need a
    gap in the structure (an empty slab between two bodies)."""
    n, r = p.n_atoms, p.n_residues
    d = np.asarray(shift, np.float64)

    def link(v, base):
        return np.where(v >= 0, v + base, -1).astype(np.int32)

    def moved(c):
        return (c + d.astype(c.dtype)) if len(c) else c

    return PackedComplex(
        xyz=np.concatenate([p.xyz, moved(q.xyz)]), vdw=np.concatenate([p.vdw, q.vdw]), cov=np.concatenate([p.cov, q.cov]),
        type_mask=np.concatenate([p.type_mask, q.type_mask]), flags=np.concatenate([p.flags, q.flags]),
        res_id=np.concatenate([p.res_id, q.res_id + r]).astype(np.int32),
        res_flags=np.concatenate([p.res_flags, q.res_flags]),
        res_prev=np.concatenate([p.res_prev, link(q.res_prev, r)]).astype(np.int32),
        res_next=np.concatenate([p.res_next, link(q.res_next, r)]).astype(np.int32),
        bond_off=np.concatenate([p.bond_off, q.bond_off[1:] + p.bond_off[-1]]).astype(np.int32),
        bond_idx=np.concatenate([p.bond_idx, q.bond_idx + n]).astype(np.int32),
        h_off=np.concatenate([p.h_off, q.h_off[1:] + p.h_off[-1]]).astype(np.int32),
        h_xyz=np.concatenate([p.h_xyz, moved(q.h_xyz)]),
        sb_nbr=np.concatenate([p.sb_nbr, link(q.sb_nbr, n)]).astype(np.int32),
        ring_center=np.concatenate([p.ring_center, moved(q.ring_center)]),
        ring_normal=np.concatenate([p.ring_normal, q.ring_normal]),
        ring_res=np.concatenate([p.ring_res, link(q.ring_res, r)]).astype(np.int32),
        amide_center=np.concatenate([p.amide_center, moved(q.amide_center)]),
        amide_normal=np.concatenate([p.amide_normal, q.amide_normal]),
        amide_res=np.concatenate([p.amide_res, link(q.amide_res, r)]).astype(np.int32))


class GlooTransport:
    """The host-buffer transport of arpeggio_amd.sharding over torch.distributed (gloo in the CPU tests): neighbour exchange of
    uint8 arrays (sizes first) and an all-reduce (MAX).  Test infrastructure: the product moves its halos with RCCL behind the
    C ABI and never imports torch."""

    def __init__(self, dist, rank, world):
        self.dist, self.rank, self.world = dist, rank, world

    def exchange(self, payload):
        import torch
        dist, rank = self.dist, self.rank
        sides = [s for s in (-1, +1) if 0 <= rank + s < self.world]
        lens_out = {s: torch.tensor([payload[s].size if s in payload else 0], dtype=torch.int64) for s in sides}
        lens_in = {s: torch.zeros(1, dtype=torch.int64) for s in sides}
        ops = []
        for s in sides:
            ops.append(dist.P2POp(dist.isend, lens_out[s], rank + s))
            ops.append(dist.P2POp(dist.irecv, lens_in[s], rank + s))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        out = {s: torch.from_numpy(np.ascontiguousarray(payload[s], np.uint8)) if s in payload else torch.zeros(0, dtype=torch.uint8) for s in sides}
        inn = {s: torch.empty(int(lens_in[s].item()), dtype=torch.uint8) for s in sides}
        ops = []
        for s in sides:
            if out[s].numel():
                ops.append(dist.P2POp(dist.isend, out[s], rank + s))
            if inn[s].numel():
                ops.append(dist.P2POp(dist.irecv, inn[s], rank + s))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return {s: inn[s].numpy() for s in sides if inn[s].numel()}

    def allreduce_max(self, a):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(a).copy())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return t.numpy()
