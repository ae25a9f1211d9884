"""Several structures in one pass (arp_set_batch, arpeggio_amd/batch.py): every structure's five bags must be
bit-identical to its own single-structure run, and — for the structures of the executed-reference fixture — to the
records the reference's own run_arpeggio produced (tests/golden/make_golden_core.py)."""
import numpy as np
import pytest

import oracle
from arpeggio_amd import batch, synth
from test_golden_core import core, selection_mask, canonical, check_planes   # noqa: F401  (fixture + helpers)

BAGS = ('atom_plane', 'plane_plane', 'group_group', 'group_plane')


# ------------------------------------------------------------------------------------------------------------- CPU
def test_concat_and_split_round_trip():
    pcs = [synth.make_synthetic(300, seed=1, box=(30, 30, 30), n_rings=5, n_amides=4),
           synth.make_synthetic(0, seed=2, box=(20, 20, 20), n_rings=6, n_amides=0),
           synth.make_synthetic(500, seed=3, box=(35, 35, 35), n_rings=0, n_amides=7)]
    big, off = batch.concat_complexes(pcs)
    assert big.n_atoms == sum(p.n_atoms for p in pcs) and big.n_rings == sum(p.n_rings for p in pcs)
    assert off['atom'][-1] == big.n_atoms and off['amide'][-1] == big.n_amides
    for k, pc in enumerate(pcs):
        a0, a1 = off['atom'][k], off['atom'][k + 1]
        assert np.array_equal(big.xyz[a0:a1], pc.xyz)                                    # coordinates untouched
        assert np.array_equal(big.res_id[a0:a1] - off['residue'][k], pc.res_id)
        assert np.array_equal(big.bond_off[a0:a1 + 1] - big.bond_off[a0], pc.bond_off)
        b0, b1 = big.bond_off[a0], big.bond_off[a1]
        assert np.array_equal(big.bond_idx[b0:b1] - a0, pc.bond_idx)
        assert np.array_equal(big.h_off[a0:a1 + 1] - big.h_off[a0], pc.h_off)
        sb = big.sb_nbr[a0:a1]
        assert np.array_equal(np.where(sb >= 0, sb - a0, sb), pc.sb_nbr)
        lo, hi = off['boxes'][k, :3], off['boxes'][k, 3:]
        if pc.n_atoms:
            assert (pc.xyz.astype(np.float64) >= lo).all() and (pc.xyz.astype(np.float64) <= hi).all()
    # results of the single structures, shifted and shuffled together, come apart again
    singles = []
    for pc in pcs:
        oc = oracle.OracleComplex(pc)
        oc.make_selection(None)
        singles.append(dict(atom_atom=oc.atom_contacts(5.0, 0.1, False), plane_plane=oc.plane_plane(), atom_plane=oc.atom_plane(),
                            group_group=oc.group_group(), group_plane=oc.group_plane()))
    rs = np.random.RandomState(0)

    def merged(name, cols):
        parts = []
        for k, s in enumerate(singles):
            nrec = len(s[name][next(iter(cols))])
            d = {c: np.asarray(v).copy() for c, v in s[name].items() if isinstance(v, np.ndarray) and v.ndim == 1 and len(v) == nrec}
            for c, which in cols.items():
                d[c] = d[c] + off[which][k]
            parts.append(d)
        keys = set.intersection(*[set(p) for p in parts])
        out = {c: np.concatenate([p[c] for p in parts]) for c in keys}
        perm = rs.permutation(len(out[next(iter(cols))]))
        return {c: v[perm] for c, v in out.items()}

    aa = merged('atom_atom', {'i': 'atom', 'j': 'atom'})
    o = np.lexsort((aa['j'], aa['i']))
    aa = {c: v[o] for c, v in aa.items()}
    for k, d in enumerate(batch.split_atom_contacts(aa, off)):
        for c in ('i', 'j', 'sift', 'ctype'):
            assert np.array_equal(d[c], singles[k]['atom_atom'][c]), (k, c)
    ap = merged('atom_plane', {'atom': 'atom', 'ring': 'ring'})
    o = np.lexsort((ap['atom'], ap['ring']))
    ap = {c: v[o] for c, v in ap.items()}
    for k, d in enumerate(batch.split_bag('atom_plane', ap, off)):
        assert np.array_equal(d['atom'], singles[k]['atom_plane']['atom']) and np.array_equal(d['ring'], singles[k]['atom_plane']['ring'])


# ------------------------------------------------------------------------------------------------------------- GPU
def _single(ctx, pc, sel, cutoff=5.0, comp=0.1, seq=False):
    ctx.set_complex(pc)
    if sel is not None:
        ctx.set_selection(sel)
    counts = ctx.run_launch(cutoff, comp, seq, 6.0)
    out = dict(atom_atom=ctx.atom_contacts_fetch(counts['atom_atom']))
    for b in BAGS:
        out[b] = ctx.fetch_bag(b)
    return out


def _same(a, b, what):
    assert set(a) == set(b), what
    for k in a:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        assert x.shape == y.shape, (what, k, x.shape, y.shape)
        if x.dtype.kind == 'f':
            assert np.array_equal(x.view(np.uint32 if x.dtype == np.float32 else np.uint64), y.view(np.uint32 if y.dtype == np.float32 else np.uint64)), (what, k)
        else:
            assert np.array_equal(x, y), (what, k)


@pytest.mark.gpu
def test_batch_equals_single_runs():
    from arpeggio_amd import _capi
    pcs = [synth.proteinlike(n_res=120, seed=11, n_waters=60, id='p0'),
           synth.make_synthetic(3000, seed=5, box=(45, 45, 45), n_rings=30, n_amides=40),
           synth.proteinlike(n_res=60, seed=12, n_waters=20, id='p1'),
           synth.make_synthetic(0, seed=6, box=(25, 25, 25), n_rings=40, n_amides=30),       # rings and amides only
           synth.make_synthetic(800, seed=7, box=(60, 20, 20), n_rings=0, n_amides=0),        # an elongated box
           synth.make_synthetic(1, seed=8, box=(5, 5, 5), n_rings=0, n_amides=0),             # one atom
           synth.proteinlike(n_res=200, seed=13, n_waters=100, id='p2')]
    ctx = _capi.Context(0)
    for cutoff, comp, seq in ((5.0, 0.1, False), (4.0, 0.25, True), (7.5, 0.1, False)):
        singles = [_single(ctx, pc, None, cutoff, comp, seq) for pc in pcs]
        ctx.set_batch(pcs)
        got = ctx.run_batch(cutoff, comp, seq, 6.0)
        assert len(got) == len(pcs)
        for k in range(len(pcs)):
            for name in ('atom_atom',) + BAGS:
                _same(got[k][name], singles[k][name], (cutoff, k, name))
        assert sum(len(s['atom_atom']['i']) for s in singles) > 20_000
    # per-structure selections: a ligand-like residue range in one, a few residues in another, whole structure elsewhere
    rs = np.random.RandomState(3)
    sels = []
    for pc in pcs:
        m = np.zeros(pc.n_atoms, np.uint8)
        if pc.n_atoms > 100:
            picked = rs.choice(pc.n_residues, size=max(1, pc.n_residues // 20), replace=False)
            m[np.isin(pc.res_id, picked)] = 1
        else:
            m[:] = 1
        sels.append(m)
    sels[1] = None
    singles = [_single(ctx, pc, (np.ones(pc.n_atoms, np.uint8) if s is None else s)) for pc, s in zip(pcs, sels)]
    ctx.set_batch(pcs, selections=sels)
    got = ctx.run_batch()
    for k in range(len(pcs)):
        for name in ('atom_atom',) + BAGS:
            _same(got[k][name], singles[k][name], ('selection', k, name))
    ctx.close()


@pytest.mark.gpu
def test_batch_of_fixture_structures_equals_executed_reference(core):
    """The proteinlike cases of the executed-reference fixture, all in ONE pass, against the reference's own records."""
    from arpeggio_amd import _capi
    z, meta, exports, pack = core
    cases = [c for c in meta if 'raises' not in c and c['mode'] == 'canonical' and c['pack'].startswith('proteinlike')]
    groups = {}
    for c in cases:      # one batch per parameter set (cutoff, comp, sequence neighbours are per pass)
        groups.setdefault((c['cutoff'], c['comp'], c['seq_adj']), []).append(c)
    ctx = _capi.Context(0)
    n_struct = n_rec = 0
    for (cutoff, comp, seq), cs in groups.items():
        cs = cs * 2          # every structure twice: neighbours in the grid with identical coordinates
        pcs = [pack(c['pack']) for c in cs]
        sels = [selection_mask(z, pack, c) for c in cs]
        ctx.set_batch(pcs, selections=sels)
        got = ctx.run_batch(cutoff, comp, seq, 6.0)
        for c, g in zip(cs, got):
            exp = canonical(z, c['name'])
            aa = g['atom_atom']
            assert np.array_equal(aa['i'], exp['bgn']) and np.array_equal(aa['j'], exp['end']), c['name']
            assert np.array_equal(aa['dist'].view(np.uint32), exp['dist'].view(np.uint32)), c['name']
            assert np.array_equal(aa['sift'], exp['sift']) and np.array_equal(aa['ctype'], exp['ctype']), c['name']
            check_planes(z, c['name'], g['atom_plane'], g['plane_plane'], g['group_group'], g['group_plane'])
            n_rec += len(aa['i'])
            n_struct += 1
    assert n_struct >= 8 and n_rec > 20_000
    ctx.close()
