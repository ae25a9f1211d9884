"""Slab sharding of one structure across the GPUs of a node.

The box is cut into ``world`` slabs along x.  Each rank owns the atoms, rings and amides
whose (centre) x lies in its slab and receives a one-cell halo from its two neighbours.
On the GPUs the exchange is the library's own (include/arpeggio_hip.h, ``arp_comm_*`` /
``arp_shard_exchange_*``: RCCL over xGMI on the context's stream); this module only
partitions and keeps the books.  The host-buffer variants (``make_shard_distributed``,
``combine_selection``) take a ``transport`` object — ``exchange({side: uint8 array}) ->
{side: uint8 array}`` with the ranks ``rank - 1`` / ``rank + 1`` and ``allreduce_max(uint8
array)`` — which the CPU tests implement over gloo; nothing here imports torch.
Halo records are self-contained: hydrogens, bonded global ids, residue links and the
single-bond-neighbour coordinates travel with the atom.

Ownership rule: a pair is emitted by the rank that owns the atom (ring, amide) with the
lower global id, so the union of the per-rank results equals the single-GPU result and the
bgn/end orientation is unchanged (``arp_set_ownership`` / ``arp_set_group_ownership``).

The reference has no counterpart: it is a single-threaded program.
"""
from __future__ import annotations

import io
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from .core.packed import PackedComplex


def halo_width(cutoff: float = 5.0, expand: float = 6.0) -> float:
    """One cell: >= every search radius of the path (I:707 cutoff, I:1420/960/1113/1270/1351 6.0 A)."""
    return max(cutoff, expand, 6.0) * (1.0 + 1e-5) + 1e-4


def slab_edges(xmin: float, xmax: float, world: int) -> np.ndarray:
    e = np.linspace(float(xmin), float(xmax), world + 1)
    e[0], e[-1] = -np.inf, np.inf
    return e


def owner_of(x, edges) -> np.ndarray:
    return np.searchsorted(edges[1:-1], np.asarray(x, np.float64), side='right').astype(np.int64)


# ---------------------------------------------------------------------------------------------
# records
# ---------------------------------------------------------------------------------------------
def _single_bond_coords(pc: PackedComplex):
    has = pc.sb_nbr >= 0
    xyz = np.zeros((pc.n_atoms, 3), np.float32)
    xyz[has] = pc.xyz[pc.sb_nbr[has]]
    return xyz, has.astype(np.uint8)


def _runs(starts, counts) -> np.ndarray:
    """Indices of the concatenated runs [starts[k], starts[k] + counts[k]) — the rows of a CSR section that belong to a
    list of atoms, in the order of the list."""
    counts = np.asarray(counts, np.int64)
    total = int(counts.sum())
    if total == 0:
        return np.zeros(0, np.int64)
    first = np.concatenate([[0], np.cumsum(counts)[:-1]])
    return np.repeat(np.asarray(starts, np.int64) - first, counts) + np.arange(total, dtype=np.int64)


def pack_records(pc: PackedComplex, atom_ids, ring_ids, amide_ids, sel=None) -> Dict[str, np.ndarray]:
    """Self-contained records of the given atoms / rings / amides (ids = packed indices = global ids)."""
    a = np.asarray(atom_ids, np.int64)
    r = np.asarray(ring_ids, np.int64)
    m = np.asarray(amide_ids, np.int64)
    sb_xyz, sb_has = _single_bond_coords(pc)
    res = pc.res_id[a]
    hc = (pc.h_off[a + 1] - pc.h_off[a]).astype(np.int32)
    bc = (pc.bond_off[a + 1] - pc.bond_off[a]).astype(np.int32)
    h_idx, b_idx = _runs(pc.h_off[a], hc), _runs(pc.bond_off[a], bc)
    return {
        'gid': a.astype(np.int32), 'xyz': pc.xyz[a], 'vdw': pc.vdw[a], 'cov': pc.cov[a], 'tmask': pc.type_mask[a],
        'flags': pc.flags[a], 'res_gid': res.astype(np.int32), 'res_flags': pc.res_flags[res],
        'res_prev': pc.res_prev[res], 'res_next': pc.res_next[res],
        'sel': (np.ones(a.size, np.uint8) if sel is None else np.asarray(sel, np.uint8)[a]),
        'sb_xyz': sb_xyz[a], 'sb_has': sb_has[a], 'h_cnt': hc, 'h_xyz': pc.h_xyz[h_idx.astype(np.int64)],
        'bond_cnt': bc, 'bond_gid': pc.bond_idx[b_idx.astype(np.int64)].astype(np.int32),
        'ring_gid': r.astype(np.int32), 'ring_center': pc.ring_center[r], 'ring_normal': pc.ring_normal[r],
        'ring_res': pc.ring_res[r], 'amide_gid': m.astype(np.int32), 'amide_center': pc.amide_center[m],
        'amide_normal': pc.amide_normal[m], 'amide_res': pc.amide_res[m],
    }


def _to_bytes(rec: Dict[str, np.ndarray]) -> np.ndarray:
    bio = io.BytesIO()
    np.savez(bio, **rec)
    return np.frombuffer(bio.getvalue(), np.uint8).copy()


def _from_bytes(buf: np.ndarray) -> Dict[str, np.ndarray]:
    z = np.load(io.BytesIO(buf.tobytes()), allow_pickle=False)
    return {k: z[k] for k in z.files}


def _concat(recs: List[Dict[str, np.ndarray]]) -> Dict[str, np.ndarray]:
    return {k: np.concatenate([r[k] for r in recs], axis=0) for k in recs[0]}


@dataclass
class Shard:
    pc: PackedComplex                 # home + halo, atoms / rings / amides sorted by global id
    is_home: np.ndarray
    global_id: np.ndarray
    ring_home: np.ndarray
    ring_gid: np.ndarray
    amide_home: np.ndarray
    amide_gid: np.ndarray
    sel: np.ndarray                   # selection bit of every local atom
    sb_xyz: np.ndarray
    sb_has: np.ndarray
    res_gid: np.ndarray               # global residue id of every local residue
    ring_res_gid: np.ndarray
    amide_res_gid: np.ndarray
    n_res_global: int
    origin: Optional[np.ndarray] = None       # 0 home, -1 / +1 halo received from the left / right neighbour
    rank: int = 0
    world: int = 1
    send_left: Optional[np.ndarray] = None    # global ids of my home atoms that are in the left / right neighbour's halo
    send_right: Optional[np.ndarray] = None
    halo_ms: float = 0.0
    halo_bytes: int = 0
    halo: float = 0.0                         # width the faces were cut with: no search radius of a run may exceed it


def _check_radius(sh, cutoff, expand=6.0):
    """A shard holds its neighbours' atoms up to ``halo`` beyond its faces: a run whose search radius is larger would
    silently miss pairs across the faces."""
    if sh is not None and getattr(sh, 'world', 1) > 1 and getattr(sh, 'halo', 0.0) > 0.0 and max(cutoff, expand) > sh.halo:
        raise ValueError(f'search radius {max(cutoff, expand)} exceeds the halo width {sh.halo:.4f} this shard was cut with')


def _lookup(sorted_keys: np.ndarray, q: np.ndarray) -> np.ndarray:
    """Index of q in sorted_keys, -1 where absent."""
    q = np.asarray(q)
    if sorted_keys.size == 0:
        return np.full(q.shape, -1, np.int64)
    pos = np.searchsorted(sorted_keys, q)
    pos = np.minimum(pos, sorted_keys.size - 1)
    return np.where(sorted_keys[pos] == q, pos, -1).astype(np.int64)


def assemble_shard(home: Dict[str, np.ndarray], halos: Dict[int, Dict[str, np.ndarray]], n_res_global: int, rank=0, world=1) -> Shard:
    """Local PackedComplex from this rank's home records and the halo records it received
    (halos: side (-1 left, +1 right) -> records)."""
    sides = sorted(halos)
    parts = [home] + [halos[s] for s in sides]
    origin = np.concatenate([np.zeros(home['gid'].size, np.int8)] + [np.full(halos[s]['gid'].size, s, np.int8) for s in sides])
    home_flag = [np.ones(home['gid'].size, np.uint8)] + [np.zeros(h['gid'].size, np.uint8) for h in parts[1:]]
    ring_flag = [np.ones(home['ring_gid'].size, np.uint8)] + [np.zeros(h['ring_gid'].size, np.uint8) for h in parts[1:]]
    am_flag = [np.ones(home['amide_gid'].size, np.uint8)] + [np.zeros(h['amide_gid'].size, np.uint8) for h in parts[1:]]
    rec = _concat(parts)
    is_home, ring_home, am_home = np.concatenate(home_flag), np.concatenate(ring_flag), np.concatenate(am_flag)

    # atoms by global id (h_xyz / bond_gid segments follow their atoms)
    order = np.argsort(rec['gid'], kind='stable')
    gid = rec['gid'][order]
    assert np.all(np.diff(gid) > 0), 'duplicate atom in home+halo'
    h_start = np.concatenate([[0], np.cumsum(rec['h_cnt'])])[:-1]
    b_start = np.concatenate([[0], np.cumsum(rec['bond_cnt'])])[:-1]
    h_cnt, b_cnt = rec['h_cnt'][order], rec['bond_cnt'][order]
    h_sel, b_sel = _runs(h_start[order], h_cnt), _runs(b_start[order], b_cnt)
    h_xyz = rec['h_xyz'][h_sel.astype(np.int64)]
    bond_gid = rec['bond_gid'][b_sel.astype(np.int64)]
    # bonds: global partner id -> local index, partners that are not local cannot form a local pair
    owner_row = np.repeat(np.arange(gid.size), b_cnt)
    partner = _lookup(gid, bond_gid)
    keep = partner >= 0
    b_cnt_local = np.bincount(owner_row[keep], minlength=gid.size)
    bond_off = np.concatenate([[0], np.cumsum(b_cnt_local)]).astype(np.int32)
    bond_idx = partner[keep].astype(np.int32)
    h_off = np.concatenate([[0], np.cumsum(h_cnt)]).astype(np.int32)

    # rings / amides by global id
    ro = np.argsort(rec['ring_gid'], kind='stable')
    ao = np.argsort(rec['amide_gid'], kind='stable')
    ring_gid, amide_gid = rec['ring_gid'][ro], rec['amide_gid'][ao]
    ring_res_g, amide_res_g = rec['ring_res'][ro], rec['amide_res'][ao]

    # The shard keeps GLOBAL residue ids (a table of n_res_global rows of which only the rows of local residues
    # are filled): residue links, ring/amide residues and the residue sets then have the same layout on every
    # rank, so the sets can be all-reduced in place on the device.
    res_g_atoms = rec['res_gid'][order].astype(np.int64)
    nres = int(n_res_global)
    res_gid = np.arange(nres, dtype=np.int32)
    res_local = res_g_atoms
    res_flags = np.zeros(nres, np.uint8)
    res_prev = np.full(nres, -1, np.int32)
    res_next = np.full(nres, -1, np.int32)
    res_flags[res_local] = rec['res_flags'][order]
    res_prev[res_local] = rec['res_prev'][order]
    res_next[res_local] = rec['res_next'][order]
    ring_res = ring_res_g.astype(np.int32)
    amide_res = amide_res_g.astype(np.int32)

    pc = PackedComplex(
        xyz=rec['xyz'][order], vdw=rec['vdw'][order], cov=rec['cov'][order], type_mask=rec['tmask'][order],
        flags=rec['flags'][order], res_id=res_local.astype(np.int32), res_flags=res_flags, res_prev=res_prev, res_next=res_next,
        bond_off=bond_off, bond_idx=bond_idx, h_off=h_off, h_xyz=h_xyz, sb_nbr=np.full(gid.size, -1, np.int32),
        ring_center=rec['ring_center'][ro], ring_normal=rec['ring_normal'][ro], ring_res=ring_res,
        amide_center=rec['amide_center'][ao], amide_normal=rec['amide_normal'][ao], amide_res=amide_res,
        id=f'shard{rank}of{world}')
    return Shard(pc=pc, is_home=is_home[order], global_id=gid.astype(np.int32), ring_home=ring_home[ro], ring_gid=ring_gid.astype(np.int32),
                 amide_home=am_home[ao], amide_gid=amide_gid.astype(np.int32), sel=rec['sel'][order],
                 sb_xyz=rec['sb_xyz'][order], sb_has=rec['sb_has'][order], res_gid=res_gid, ring_res_gid=ring_res_g.astype(np.int32),
                 amide_res_gid=amide_res_g.astype(np.int32), n_res_global=int(n_res_global), rank=rank, world=world,
                 origin=origin[order])


def _partition(full: PackedComplex, world: int, halo: float):
    edges = slab_edges(float(full.xyz[:, 0].min()) if full.n_atoms else 0.0, float(full.xyz[:, 0].max()) if full.n_atoms else 1.0, world)
    inner = np.diff(edges[1:-1]) if world > 2 else np.array([np.inf])
    if world > 1 and np.any(inner < halo):
        raise ValueError('slabs are thinner than the halo: use fewer ranks for this structure')
    return edges, owner_of(full.xyz[:, 0], edges), owner_of(full.ring_center[:, 0], edges), owner_of(full.amide_center[:, 0], edges)


def _face_sets(full, edges, a_own, r_own, m_own, rank, side, halo):
    """Home items of `rank` inside the halo of its `side` (-1 left, +1 right) neighbour."""
    if side < 0:
        face = edges[rank]
        am = (a_own == rank) & (full.xyz[:, 0].astype(np.float64) <= face + halo)
        rm = (r_own == rank) & (full.ring_center[:, 0] <= face + halo)
        mm = (m_own == rank) & (full.amide_center[:, 0].astype(np.float64) <= face + halo)
    else:
        face = edges[rank + 1]
        am = (a_own == rank) & (full.xyz[:, 0].astype(np.float64) >= face - halo)
        rm = (r_own == rank) & (full.ring_center[:, 0] >= face - halo)
        mm = (m_own == rank) & (full.amide_center[:, 0].astype(np.float64) >= face - halo)
    return np.nonzero(am)[0], np.nonzero(rm)[0], np.nonzero(mm)[0]


def make_shard_local(full: PackedComplex, rank: int, world: int, sel=None, cutoff=5.0) -> Shard:
    """Shard built from global knowledge (no communication): the reference result of the exchange."""
    halo = halo_width(cutoff)
    edges, a_own, r_own, m_own = _partition(full, world, halo)
    home = pack_records(full, np.nonzero(a_own == rank)[0], np.nonzero(r_own == rank)[0], np.nonzero(m_own == rank)[0], sel)
    halos, sends = {}, {}
    for side, nb in ((-1, rank - 1), (+1, rank + 1)):
        if 0 <= nb < world:
            ai, ri, mi = _face_sets(full, edges, a_own, r_own, m_own, nb, -side, halo)   # what the neighbour sends towards me
            halos[side] = pack_records(full, ai, ri, mi, sel)
            sends[side] = _face_sets(full, edges, a_own, r_own, m_own, rank, side, halo)[0]
    sh = assemble_shard(home, halos, full.n_residues, rank, world)
    sh.send_left, sh.send_right = sends.get(-1), sends.get(+1)
    sh.halo = halo
    return sh


def make_shard_distributed(full: PackedComplex, rank: int, world: int, transport=None, sel=None, cutoff=5.0) -> Shard:
    """Each rank keeps only its slab of `full` and obtains its halo from its neighbours through ``transport``
    (host buffers; see the module docstring)."""
    halo = halo_width(cutoff)
    edges, a_own, r_own, m_own = _partition(full, world, halo)
    home = pack_records(full, np.nonzero(a_own == rank)[0], np.nonzero(r_own == rank)[0], np.nonzero(m_own == rank)[0], sel)
    payload, sends = {}, {}
    for side in (-1, +1):
        if 0 <= rank + side < world:
            ai, ri, mi = _face_sets(full, edges, a_own, r_own, m_own, rank, side, halo)   # my home items the neighbour needs
            payload[side] = _to_bytes(pack_records(full, ai, ri, mi, sel))
            sends[side] = ai
    t0 = time.perf_counter()
    received = transport.exchange(payload) if world > 1 else {}
    ms = (time.perf_counter() - t0) * 1e3
    halos = {s: _from_bytes(received[s]) for s in received}
    sh = assemble_shard(home, halos, full.n_residues, rank, world)
    sh.send_left, sh.send_right = sends.get(-1), sends.get(+1)
    sh.halo_ms = ms
    sh.halo_bytes = int(sum(v.size for v in payload.values()))
    sh.halo = halo
    return sh


# ---------------------------------------------------------------------------------------------
# the same on the device: the host only packs its HOME records; cutting the faces out, moving them (RCCL on device
# pointers) and merging home + halos into the resident structure happen in HBM (arp_shard_* of the C ABI)
# ---------------------------------------------------------------------------------------------
@dataclass
class DeviceShard:
    """What the host keeps of a shard assembled on the device: the id maps (to translate results and to drive the
    selection exchange); the structure itself is resident in the context."""
    n_atoms: int
    is_home: np.ndarray
    global_id: np.ndarray
    origin: np.ndarray
    sel: np.ndarray
    ring_home: np.ndarray
    ring_gid: np.ndarray
    amide_home: np.ndarray
    amide_gid: np.ndarray
    n_res_global: int
    rank: int = 0
    world: int = 1
    send_left: Optional[np.ndarray] = None
    send_right: Optional[np.ndarray] = None
    halo: float = 0.0                 # width the faces were cut with: no search radius of a run may exceed it
    halo_ms: float = 0.0
    halo_bytes: int = 0
    timings_ms: Optional[dict] = None # where the set-up time went: host packing of the home records / device work / exchange


def _home_buffer_to_device(ctx, buf, home_ids, hx, edges, halo, n_res_global, rank, world, pack_ms):
    t1 = time.perf_counter()
    ctx.shard_set_home(buf)
    faces, sends = {}, {}
    for side in (-1, +1):
        if 0 <= rank + side < world:
            lo, hi = (-np.inf, edges[rank] + halo) if side < 0 else (edges[rank + 1] - halo, np.inf)
            faces[side] = ctx.shard_pack_face(0 if side < 0 else 1, lo, hi)
            sends[side] = home_ids[(hx >= lo) & (hx <= hi)]           # the same comparison the kernel makes
    t2 = time.perf_counter()
    return faces, dict(sends=sends, halo=halo, n_res_global=n_res_global, rank=rank, world=world,
                       timings={'host_pack_home_records': pack_ms, 'upload_home_and_cut_faces': (t2 - t1) * 1e3})


def shard_home_to_device(ctx, full: PackedComplex, rank: int, world: int, sel=None, cutoff=5.0):
    """Step 1 of the device assembly: this rank's home records to its GPU, the two face buffers cut out there.
    Returns (faces: side -> (pointer, bytes), bookkeeping for ``finish_shard_on_device``)."""
    halo = halo_width(cutoff)
    edges, a_own, r_own, m_own = _partition(full, world, halo)
    from . import _capi
    t0 = time.perf_counter()
    home_ids = np.nonzero(a_own == rank)[0]
    buf = _capi.pack_records_native(full, home_ids, np.nonzero(r_own == rank)[0], np.nonzero(m_own == rank)[0], sel)
    return _home_buffer_to_device(ctx, buf, home_ids, full.xyz[home_ids, 0].astype(np.float64), edges, halo, full.n_residues, rank, world,
                                  (time.perf_counter() - t0) * 1e3)


def shard_records_to_device(ctx, rec: Dict[str, np.ndarray], book: dict, rank: int, world: int, cutoff=5.0):
    """The same from home records a rank made itself (``synth.slab_home_records``: a rank that generates — or reads — only
    its own slab never holds the whole structure).  ``book``: edges of the slabs, size of the global residue table, x of the
    home atoms."""
    halo = halo_width(cutoff)
    edges = np.asarray(book['edges'], np.float64)
    inner = np.diff(edges[1:-1]) if world > 2 else np.array([np.inf])
    if world > 1 and np.any(inner < halo):
        raise ValueError('slabs are thinner than the halo: use fewer ranks for this structure')
    from . import _capi
    t0 = time.perf_counter()
    buf = _capi.pack_records_buffer(rec)
    return _home_buffer_to_device(ctx, buf, np.asarray(rec['gid'], np.int64), np.asarray(book['home_x'], np.float64), edges, halo,
                                  int(book['n_res_global']), rank, world, (time.perf_counter() - t0) * 1e3)


def finish_shard_on_device(ctx, received, book, whole_structure=False) -> DeviceShard:
    """Step 3: merge home + received halos in HBM; ``received``: side -> (pointer, bytes[, owner])."""
    t0 = time.perf_counter()
    ctx.shard_assemble(received.get(-1), received.get(+1), book['n_res_global'])
    t1 = time.perf_counter()
    lay = ctx.shard_layout()
    sel = lay['sel']
    if whole_structure:
        if not bool(np.all(sel)):
            raise ValueError('whole_structure=True needs a shard whose selection mask is all ones')
    else:
        ctx.set_selection(sel)
    ctx.set_whole_structure(whole_structure)
    timings = dict(book.get('timings', {}), merge_on_device=(t1 - t0) * 1e3, id_maps_and_selection=(time.perf_counter() - t1) * 1e3)
    return DeviceShard(timings_ms={k: round(v, 3) for k, v in timings.items()}, n_atoms=ctx.n, is_home=(lay['origin'] == 0).astype(np.uint8), global_id=lay['global_id'], origin=lay['origin'],
                       sel=sel, ring_home=(lay['ring_origin'] == 0).astype(np.uint8), ring_gid=lay['ring_gid'],
                       amide_home=(lay['amide_origin'] == 0).astype(np.uint8), amide_gid=lay['amide_gid'],
                       n_res_global=book['n_res_global'], rank=book['rank'], world=book['world'],
                       send_left=book['sends'].get(-1), send_right=book['sends'].get(+1), halo=book['halo'])


def make_shard_device(ctx, full, rank: int, world: int, sel=None, cutoff=5.0,
                      whole_structure=False) -> DeviceShard:
    """``make_shard_distributed`` + ``upload_shard`` without the host in the data path: the halo records are cut out,
    exchanged (``arp_shard_exchange_faces``: RCCL on the context's communicator, ``Context.comm_init``) and merged on
    the GPUs.  ``full``: the whole structure (a PackedComplex), or ``(records, book)`` of this rank's home part alone
    (``synth.slab_home_records``)."""
    if isinstance(full, tuple):
        if sel is not None:
            raise ValueError('make_shard_device: home records carry their own selection column')
        faces, book = shard_records_to_device(ctx, full[0], full[1], rank, world, cutoff)
    else:
        faces, book = shard_home_to_device(ctx, full, rank, world, sel, cutoff)
    t0 = time.perf_counter()
    received = ctx.shard_exchange_faces(faces.get(-1), faces.get(+1)) if world > 1 else {}
    ms = (time.perf_counter() - t0) * 1e3
    sh = finish_shard_on_device(ctx, received, book, whole_structure)
    sh.halo_ms = ms
    sh.halo_bytes = int(sum(v[1] for v in faces.values()))
    sh.timings_ms['exchange'] = round(ms, 3)
    return sh


# ---------------------------------------------------------------------------------------------
# selection: halo exchange of selection_plus bits + all-reduce of the residue sets
# ---------------------------------------------------------------------------------------------
def combine_selection(sh: Shard, local_plus: np.ndarray, transport=None):
    """Turn the local result of the 6 A expansion into the global _make_selection state.

    local_plus is exact for home atoms (every selected atom within 6 A of a home atom is in
    home+halo); halo atoms take the bit from their owner, and the residue sets (I:1413-1437)
    are OR-ed over all ranks.
    """
    plus = np.asarray(local_plus, np.uint8).copy()
    if transport is not None and sh.world > 1:
        payload = {}
        for side, ids in ((-1, sh.send_left), (+1, sh.send_right)):
            if ids is not None:
                payload[side] = plus[_lookup(sh.global_id, ids)].astype(np.uint8)
        got = transport.exchange(payload)
        # the owner sent one bit per atom of its face set in ascending global id = the order of my halo atoms of that side
        for side, bits in got.items():
            plus[sh.origin == side] = bits
    res = np.zeros(2 * sh.n_res_global, np.uint8)
    hm = sh.is_home == 1
    res_g = sh.res_gid[sh.pc.res_id]
    res[res_g[hm & (sh.sel == 1)]] = 1
    res[sh.n_res_global + res_g[hm & (plus == 1)]] = 1
    if transport is not None and sh.world > 1:
        res = transport.allreduce_max(res)
    res_sel, res_plus = res[:sh.n_res_global], res[sh.n_res_global:]
    rr, ar = sh.ring_res_gid, sh.amide_res_gid
    ring_sel = np.where(rr >= 0, res_sel[np.maximum(rr, 0)], 0).astype(np.uint8)
    ring_plus = np.where(rr >= 0, res_plus[np.maximum(rr, 0)], 0).astype(np.uint8)
    amide_sel = np.where(ar >= 0, res_sel[np.maximum(ar, 0)], 0).astype(np.uint8)
    amide_plus = np.where(ar >= 0, res_plus[np.maximum(ar, 0)], 0).astype(np.uint8)
    return dict(sel=sh.sel, plus=plus, ring_sel=ring_sel, ring_plus=ring_plus, amide_sel=amide_sel, amide_plus=amide_plus)


# ---------------------------------------------------------------------------------------------
# GPU side
# ---------------------------------------------------------------------------------------------
def upload_shard(ctx, sh: Shard, whole_structure: bool = False):
    """``whole_structure``: the shard was cut without a selection; the context is told so (no selection exchange)."""
    if whole_structure and not bool(np.all(sh.sel)):
        raise ValueError('whole_structure=True needs a shard whose selection mask is all ones')
    ctx.set_complex(sh.pc)
    ctx.set_single_bond_neighbour_coords(sh.sb_xyz, sh.sb_has)
    ctx.set_ownership(sh.is_home, sh.global_id)
    ctx.set_group_ownership(sh.ring_home, sh.ring_gid, sh.amide_home, sh.amide_gid)
    ctx.set_selection(sh.sel)
    ctx.set_whole_structure(whole_structure)


def run_shard(ctx, sh: Shard, transport=None, cutoff=5.0, vdw_comp=0.1, include_sequence_adjacent=False):
    """run_arpeggio on one shard: local expansion, selection exchange, then the five bags (results stay in HBM)."""
    _check_radius(sh, cutoff)
    local = ctx.make_selection(sh.sel)
    st = combine_selection(sh, local['plus'], transport)
    ctx.set_selection_state(st['sel'], st['plus'], st['ring_sel'], st['ring_plus'], st['amide_sel'], st['amide_plus'])
    counts = dict(atom_atom=ctx.atom_contacts_launch(cutoff, vdw_comp, include_sequence_adjacent))
    for name in ('plane_plane', 'atom_plane', 'group_group', 'group_plane'):
        counts[name] = ctx.launch_bag(name)
    return counts


class DeviceExchange:
    """Per-pass selection exchange of one shard: halo atoms take their selection_plus bit from their owner,
    the residue sets are OR-ed over all ranks.  On the GPU (``transport=None``) both are the library's own
    (``arp_shard_exchange_plus`` / ``arp_shard_reduce_residue_sets``: RCCL on the context's stream, no host
    synchronisation between the stages); the index lists are uploaded once, here.  With a ``transport`` the context is
    expected to expose its buffers as NumPy arrays (``host_buffer``: the CPU stand-in of the tests)."""

    def __init__(self, ctx, sh, transport=None):
        self.ctx, self.sh, self.transport = ctx, sh, transport
        self.send_idx, self.recv_idx = {}, {}
        for side, ids in ((-1, sh.send_left), (+1, sh.send_right)):
            if ids is not None and 0 <= sh.rank + side < sh.world:
                self.send_idx[side] = _lookup(sh.global_id, ids).astype(np.int32)
                self.recv_idx[side] = np.nonzero(sh.origin == side)[0].astype(np.int32)
        if transport is None:
            ctx.shard_set_exchange_lists(self.send_idx.get(-1), self.send_idx.get(+1), self.recv_idx.get(-1), self.recv_idx.get(+1))

    def exchange_plus(self):
        if self.transport is None:
            self.ctx.shard_exchange_plus()
            return
        plus = self.ctx.host_buffer(self.ctx.BUF_PLUS)
        got = self.transport.exchange({side: plus[idx].astype(np.uint8) for side, idx in self.send_idx.items()})
        for side, bits in got.items():
            plus[self.recv_idx[side]] = bits

    def reduce_residue_sets(self):
        if self.transport is None:
            self.ctx.shard_reduce_residue_sets()
            return
        res = self.ctx.host_buffer(self.ctx.BUF_RES_SETS)
        res[:] = self.transport.allreduce_max(res)


def run_shard_whole_structure(ctx, cutoff=5.0, vdw_comp=0.1, include_sequence_adjacent=False, sh=None):
    """run_arpeggio on one shard when NO selection was given (whole structure, I:1395): selection_plus and the residue
    sets are known without asking the neighbours (``Context.set_whole_structure``), so the pass is the single-GPU pass on
    the shard — owned atoms, rings and amides emit — and the only traffic between the ranks is the halo of records that
    built the shard.  ``upload_shard(ctx, sh, whole_structure=True)`` prepares the context; ``sh`` (optional) lets the
    call refuse a cutoff wider than the halo."""
    _check_radius(sh, cutoff)
    return ctx.run_launch(cutoff, vdw_comp, include_sequence_adjacent, 6.0)


def run_shard_device(ctx, ex: DeviceExchange, cutoff=5.0, vdw_comp=0.1, include_sequence_adjacent=False):
    """run_arpeggio on one shard with the selection state combined on the device (three stages, two exchanges)."""
    _check_radius(ex.sh, cutoff)
    ctx.run_stage(0, cutoff, vdw_comp, include_sequence_adjacent)
    ex.exchange_plus()
    ctx.run_stage(1, cutoff, vdw_comp, include_sequence_adjacent)
    ex.reduce_residue_sets()
    return ctx.run_stage(2, cutoff, vdw_comp, include_sequence_adjacent)
