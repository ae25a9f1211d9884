"""The C oracle (CPU) and the HIP path (GPU) against fixtures produced by executing the reference's own
run_arpeggio / _make_selection / _calculate_atom_contacts / __calculate_atom_plane_contacts / plane and group loops /
is_xbond / is_halogen_weak_hbond / get_single_bond_neighbour / get_contacts / CSV writers on data holders
(tests/golden/make_golden_core.py; inputs and outputs only are stored)."""
import gzip
import json
import os

import numpy as np
import pytest

import oracle
from arpeggio_amd.core import config
from arpeggio_amd.core.packed import PackedComplex, single_bond_neighbours


@pytest.fixture(scope='module')
def core(golden_dir):
    z = np.load(os.path.join(golden_dir, 'core_cases.npz'), allow_pickle=False)
    meta = json.load(open(os.path.join(golden_dir, 'core_cases.json')))
    exports = json.loads(gzip.open(os.path.join(golden_dir, 'core_exports.json.gz')).read().decode())
    packs = {}

    def pack(name):
        if name not in packs:
            packs[name] = PackedComplex.from_arrays(z, name + '/')
        return packs[name]

    return z, meta, exports, pack


def selection_mask(z, pack, case):
    pc = pack(case['pack'])
    if case.get('selectors'):
        from arpeggio_amd.core import utils
        idx = utils.selection_parser(case['selectors'], pc)
    elif case.get('sel'):
        idx = z[case['sel']]
    else:
        idx = np.arange(pc.n_atoms)
    m = np.zeros(pc.n_atoms, np.uint8)
    m[idx] = 1
    return m


def ids(a):
    return np.nonzero(a)[0].astype(np.int32)


def check_selection(z, name, sel, plus, ring_sel, ring_plus, amide_sel, amide_plus, pc):
    assert np.array_equal(ids(sel), z[name + '/selection'])
    assert np.array_equal(ids(plus), z[name + '/selection_plus'])
    assert np.array_equal(np.unique(pc.res_id[plus != 0]), z[name + '/selection_plus_residues'])
    assert np.array_equal(ids(ring_sel), z[name + '/selection_ring_ids'])
    assert np.array_equal(ids(ring_plus), z[name + '/selection_plus_ring_ids'])
    assert np.array_equal(ids(amide_sel), z[name + '/selection_amide_ids'])
    assert np.array_equal(ids(amide_plus), z[name + '/selection_plus_amide_ids'])


def canonical(z, name):
    """The reference's atom-atom records sorted by (min, max) atom index, with a flag 'bgn is the lower index'."""
    b, e = z[name + '/aa_bgn'], z[name + '/aa_end']
    lo, hi = np.minimum(b, e), np.maximum(b, e)
    o = np.lexsort((hi, lo))
    return {k: z[f'{name}/aa_{k}'][o] for k in ('bgn', 'end', 'sift', 'ctype', 'dist')}


def check_planes(z, name, got_ap, got_pp, got_gg, got_gp):
    o = np.lexsort((z[name + '/ap_atom'], z[name + '/ap_ring']))
    assert np.array_equal(got_ap['ring'], z[name + '/ap_ring'][o]) and np.array_equal(got_ap['atom'], z[name + '/ap_atom'][o])
    assert np.array_equal(got_ap['dist'], z[name + '/ap_dist'][o])          # float64, bit-exact
    assert np.array_equal(got_ap['mask'], z[name + '/ap_mask'][o])
    assert np.array_equal(got_ap['ctype'], z[name + '/ap_ctype'][o])
    # plane-plane: creation order of the reference == (bgn, end) order
    assert np.array_equal(got_pp['bgn'], z[name + '/pp_bgn']) and np.array_equal(got_pp['end'], z[name + '/pp_end'])
    assert np.array_equal(got_pp['dist'], z[name + '/pp_dist'])
    assert np.array_equal(got_pp['type1'], z[name + '/pp_type1'])
    t2 = np.where(got_pp['type2'] >= config.PP_SAME, 255, got_pp['type2'])
    assert np.array_equal(t2, z[name + '/pp_type2'])
    assert np.array_equal(got_pp['ctype'], z[name + '/pp_ctype'])
    for got, pre, kb, ke in ((got_gg, 'gg', 'bgn', 'end'), (got_gp, 'gp', 'amide', 'ring')):
        assert np.array_equal(got[kb], z[f'{name}/{pre}_bgn']) and np.array_equal(got[ke], z[f'{name}/{pre}_end'])
        assert np.array_equal(got['dist'].astype(np.float64), z[f'{name}/{pre}_dist'])
        assert np.array_equal(got['ctype'], z[f'{name}/{pre}_ctype'])


RES_SIFTS = ('ring_ring_inter_integer_sift', 'ring_atom_inter_integer_sift', 'atom_ring_inter_integer_sift',
             'mc_atom_ring_inter_integer_sift', 'sc_atom_ring_inter_integer_sift', 'amide_amide_inter_integer_sift',
             'amide_ring_inter_integer_sift', 'ring_amide_inter_integer_sift')


# ------------------------------------------------------------------------------------------------------------- CPU
def test_fixture_covers_every_branch(core):
    z, meta, exports, pack = core
    assert len(meta) > 130
    sift_seen, ctype_seen, ap_seen, pp_seen = 0, set(), 0, set()
    for c in meta:
        if 'raises' in c:
            continue
        n = c['name']
        sift_seen |= int(np.bitwise_or.reduce(z[n + '/aa_sift'])) if len(z[n + '/aa_sift']) else 0
        ctype_seen |= set(z[n + '/aa_ctype'].tolist())
        ap_seen |= int(np.bitwise_or.reduce(z[n + '/ap_mask'])) if len(z[n + '/ap_mask']) else 0
        pp_seen |= set(z[n + '/pp_type1'].tolist())
    assert sift_seen == (1 << 15) - 1                      # all 15 flags occur
    assert ctype_seen == set(range(6))                     # all six atom-atom contact types
    assert ap_seen == 0b11111                              # CARBONPI .. METSULPHURPI
    assert len(pp_seen) >= 8
    assert {c['mode'] for c in meta if 'mode' in c} == {'canonical', 'reversed', 'random'}


def test_oracle_equals_executed_reference(core):
    """oracle/ref_c.c == the reference's own run_arpeggio on every case: selection sets, every atom-atom record in the
    orientation the reference's loop saw it (SIFt, contact type, float32 distance bit-exact), all four plane bags."""
    z, meta, exports, pack = core
    checked = 0
    for c in meta:
        if 'raises' in c:
            continue
        name, pc = c['name'], pack(c['pack'])
        oc = oracle.OracleComplex(pc)
        sel = selection_mask(z, pack, c)
        plus = oc.make_selection(sel)
        check_selection(z, name, sel, plus, oc.ring_sel, oc.ring_plus, oc.amide_sel, oc.amide_plus, pc)
        exp = canonical(z, name)
        lo, hi = np.minimum(exp['bgn'], exp['end']), np.maximum(exp['bgn'], exp['end'])
        if c['mode'] == 'canonical':
            got = oc.atom_contacts(c['cutoff'], c['comp'], c['seq_adj'])
            assert np.array_equal(got['i'], exp['bgn']) and np.array_equal(got['j'], exp['end']), name
            assert np.array_equal(got['dist'].view(np.uint32), exp['dist'].view(np.uint32)), name
            assert np.array_equal(got['sift'], exp['sift']) and np.array_equal(got['ctype'], exp['ctype']), name
        else:
            # the sequence filter reads only res_end (I:734), so even WHICH pairs are emitted depends on the orientation
            # the KD-tree delivered: evaluate every pair within the cutoff in the orientation the reference's loop saw
            pi, pj, _ = oracle.search_all(pc.xyz, c['cutoff'], active=plus, grid=False)
            emitted = {(int(b), int(e)): r for r, (b, e) in enumerate(zip(exp['bgn'], exp['end']))}
            seen = 0
            for i, j in zip(pi.tolist(), pj.tolist()):
                if (i, j) in emitted or (j, i) in emitted:
                    b, e = (i, j) if (i, j) in emitted else (j, i)
                    r = emitted[(b, e)]
                    ok, d, sft, ct, err = oc.pair_contact(b, e, c['comp'], c['seq_adj'])
                    assert ok and err == 0 and sft == exp['sift'][r] and ct == exp['ctype'][r], (name, b, e)
                    assert d == exp['dist'][r], (name, b, e)
                    seen += 1
                elif c['mode'] == 'reversed':        # the loop saw (j, i) and dropped it
                    assert not oc.pair_contact(j, i, c['comp'], c['seq_adj'])[0], (name, i, j)
                else:                                # random orientation: it was dropped in at least one of the two
                    assert not (oc.pair_contact(i, j, c['comp'], c['seq_adj'])[0]
                                and oc.pair_contact(j, i, c['comp'], c['seq_adj'])[0]), (name, i, j)
            assert seen == len(emitted), name
        check_planes(z, name, oc.atom_plane(), oc.plane_plane(), oc.group_group(), oc.group_plane())
        checked += len(lo)
    assert checked > 200_000


def test_oracle_accumulators_equal_executed_reference(core):
    """Per-atom OR masks, feature masks, hbond / polar counters and — in the canonical delivery order — the integer
    sifts of U:224-242 after the reference's own loop."""
    z, meta, exports, pack = core
    n_int = 0
    for c in meta:
        if 'raises' in c:
            continue
        name, pc = c['name'], pack(c['pack'])
        contacts = dict(i=z[name + '/aa_bgn'], j=z[name + '/aa_end'], sift=z[name + '/aa_sift'], ctype=z[name + '/aa_ctype'])
        acc = oracle.atom_accumulators(pc.n_atoms, contacts)
        assert np.array_equal(acc['sift'], z[name + '/atom_sift']), name
        assert np.array_equal((acc['sift'] >> 5) & 0x3FF, z[name + '/atom_fsift']), name
        assert np.array_equal(acc['counts'], z[name + '/atom_counts']), name
        if c['mode'] == 'canonical':
            o = np.lexsort((contacts['j'], contacts['i']))
            assert np.array_equal(o, np.arange(len(o)))      # the canonical runs delivered the pairs sorted by (i, j)
            isift = oracle.atom_integer_sifts(pc.n_atoms, contacts)
            assert np.array_equal(isift, z[name + '/atom_integer_sift']), name
            n_int += int((isift == 2).sum())
    assert n_int > 1000        # the '2' entries (bit set before the last pair AND by the last pair) are exercised


def test_single_bond_neighbours_equal_executed_reference(core):
    z, meta, exports, pack = core
    pc = pack('proteinlike')
    is_h = (pc.flags & config.F_HYDROGEN) != 0
    got = single_bond_neighbours(pc.bond_off, pc.bond_idx, z['proteinlike/bond_order'], z['proteinlike/bond_aromatic'], is_h)
    assert np.array_equal(got, z['proteinlike/sb_nbr_reference'])
    assert (got < 0).sum() > 100 and (got >= 0).sum() > 1000


def test_xbond_without_neighbour_raises_like_the_reference(core):
    z, meta, exports, pack = core
    c = next(m for m in meta if m.get('raises'))
    assert c['raises'] == 'AttributeError' and "'NoneType' object has no attribute 'GetId'" in c['message']
    oc = oracle.OracleComplex(pack(c['pack']))
    oc.make_selection(None)
    assert oc.atom_contacts()['err'] != 0


# ------------------------------------------------------------------------------------------------------------- GPU
def _run_hip(ctx, pc, sel, c):
    ctx.set_complex(pc)
    ctx.set_selection(sel)
    counts = ctx.run_launch(c['cutoff'], c['comp'], c['seq_adj'], 6.0)
    return counts


@pytest.mark.gpu
def test_hip_equals_executed_reference(core):
    """The HIP kernels, through the C ABI, against the records the reference's own code produced — no oracle in
    between.  Canonical-orientation cases: everything bit for bit; the other cases: pair set, distances, selection sets
    and the plane bags (orientation-free quantities)."""
    from arpeggio_amd import _capi
    z, meta, exports, pack = core
    ctx = _capi.Context(0)
    n_rec = 0
    for c in meta:
        if 'raises' in c:
            continue
        name, pc = c['name'], pack(c['pack'])
        sel = selection_mask(z, pack, c)
        counts = _run_hip(ctx, pc, sel, c)
        m = ctx.make_selection_masks()
        check_selection(z, name, sel, m['plus'], m['ring_sel'], m['ring_plus'], m['amide_sel'], m['amide_plus'], pc)
        check_planes(z, name, ctx.fetch_bag('atom_plane'), ctx.fetch_bag('plane_plane'), ctx.fetch_bag('group_group'),
                     ctx.fetch_bag('group_plane'))
        if c['mode'] != 'canonical':      # the HIP path is defined on the canonical orientation (bgn = lower index)
            continue
        got = ctx.atom_contacts_fetch(counts['atom_atom'])
        exp = canonical(z, name)
        lo = exp['bgn']
        assert np.array_equal(got['i'], exp['bgn']) and np.array_equal(got['j'], exp['end']), name
        assert np.array_equal(got['dist'].view(np.uint32), exp['dist'].view(np.uint32)), name
        assert np.array_equal(got['sift'], exp['sift']) and np.array_equal(got['ctype'], exp['ctype']), name
        acc = ctx.atom_accumulators()
        assert np.array_equal(acc['sift'], z[name + '/atom_sift']), name
        assert np.array_equal(acc['counts'], z[name + '/atom_counts']), name
        assert np.array_equal(ctx.atom_integer_sifts(), z[name + '/atom_integer_sift']), name
        n_rec += len(lo)
    assert n_rec > 150_000
    ctx.close()


@pytest.mark.gpu
def test_hip_xbond_without_neighbour_raises_like_the_reference(core):
    from arpeggio_amd import _capi
    z, meta, exports, pack = core
    c = next(m for m in meta if m.get('raises'))
    ctx = _capi.Context(0)
    ctx.set_complex(pack(c['pack']))
    ctx.set_selection(np.ones(2, np.uint8))
    with pytest.raises(AttributeError, match="'NoneType' object has no attribute 'GetId'"):
        ctx.run_launch(5.0, 0.1, False, 6.0)
    ctx.close()


@pytest.mark.gpu
def test_drop_in_exports_equal_executed_reference(core, tmp_path):
    """InteractionComplex.get_contacts() and the CSV writers, byte for byte against the text the reference's own
    get_contacts / write_contacts / write_atom_types / write_atom_sifts / write_residue_sifts / write_polar_matching
    produced on the same structure (rows of the set-ordered files compared as sorted lists)."""
    from arpeggio_amd.core import InteractionComplex, export
    z, meta, exports, pack = core
    assert len(exports) >= 5
    for name, exp in exports.items():
        c = next(m for m in meta if m['name'] == name)
        pc = pack(c['pack'])
        ic = InteractionComplex(pc)
        ic.structure_checks()
        ic.initialize()
        sel = z[c['sel']] if c.get('sel') else (c.get('selectors') or [])   # atom list (extension) or selector strings
        ic.run_arpeggio(sel, c['cutoff'], c['comp'], c['seq_adj'])
        assert json.dumps(ic.get_contacts(), sort_keys=True) == exp['get_contacts'], name
        wd = tmp_path / name.replace(':', '_')
        wd.mkdir()
        ic.write_json(str(wd / 'out.json'))        # the CLI's file (CLI:184-188), written natively
        assert open(wd / 'out.json').read() == json.dumps(json.loads(exp['get_contacts']), indent=4, sort_keys=True), name
        os.remove(wd / 'out.json')
        ic.write_contacts(ic.selection if (c.get('selectors') or c.get('sel')) else [], str(wd))
        assert np.array_equal(export.potential_fsift(pc), z[name + '/atom_potential_fsift']), name
        ic.write_atom_types(str(wd))
        ic.write_atom_sifts(str(wd))
        ic.write_residue_sifts(str(wd))
        ic.write_polar_matching(str(wd))
        files = sorted(os.listdir(wd))
        assert files == sorted(k for k in exp if k != 'get_contacts'), name
        for fn in files:
            got = open(wd / fn, newline='').read()
            if fn.endswith(('_contacts.csv', '_atomtypes.csv')):          # list-ordered in the reference
                assert got == exp[fn], (name, fn)
            else:                                                            # the reference iterates a set
                g, e = got.split('\r\n'), exp[fn].split('\r\n')
                has_header = not fn.endswith('polarmatch.csv')
                if has_header:
                    assert g[0] == e[0], (name, fn)
                assert sorted(g[1 if has_header else 0:]) == sorted(e[1 if has_header else 0:]), (name, fn)
