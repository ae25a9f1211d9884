"""The canonical (i, j) order of the atom-atom bag made on the device (csrc/arp_sort.h) and the one-copy fetch of all
five bags (arp_fetch_packed), through the C ABI.  Needs a real MI355X: `pytest -m gpu`.

The order is the boundary's definition of the reference's record order (interactions.py:707 delivers the pairs in KD-tree
order, interactions.py:183-190 exports them in that order; DESIGN.md 1 defines ascending (bgn, end) packed index instead):
the checks are bit-exact — the sorted columns must be the unsorted ones permuted by np.lexsort."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def capi():
    from arpeggio_amd import _capi
    return _capi


@pytest.fixture(scope='module')
def ctx(capi):
    c = capi.Context(0)
    yield c
    c.close()


def _same_bag(a, b):
    assert len(a['i']) == len(b['i'])
    for k in ('i', 'j', 'sift', 'ctype'):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a['dist'].view(np.uint32), b['dist'].view(np.uint32))


def _host_sorted(raw):
    o = np.lexsort((raw['j'], raw['i']))
    return {k: v[o] for k, v in raw.items()}


# 1 contact .. several tiles of 16 384 per block (more than 256 tiles: 4.3 M records at 350 k atoms)
@pytest.mark.parametrize('n,seed', [(2, 1), (40, 2), (700, 3), (1300, 4), (2600, 5), (20000, 6), (100000, 3), (350000, 7)])
def test_device_order_is_the_lexsort_of_the_unsorted_bag(ctx, n, seed):
    from arpeggio_amd import synth
    pc = synth.config3(n, seed=seed)
    ctx.set_complex(pc)
    cnt = ctx.atom_contacts_launch(5.0, 0.1, False)
    raw = {k: v.copy() for k, v in ctx.atom_contacts_fetch(cnt, sort=False).items()}
    acc0 = ctx.atom_accumulators()
    got = ctx.atom_contacts_fetch(cnt, sort=True)
    assert len(got['i']) == cnt
    _same_bag(got, _host_sorted(raw))
    if cnt:
        key = got['i'].astype(np.int64) << 32 | got['j'].astype(np.int64)
        assert np.all(np.diff(key) > 0), 'strictly ascending (every pair once)'
    # the sort is idempotent, fetches after it keep the order, and the per-atom accumulators do not depend on it
    ctx.sort_contacts()
    _same_bag(ctx.atom_contacts_fetch(cnt, sort=False), got)
    acc1 = ctx.atom_accumulators()
    assert np.array_equal(acc0['sift'], acc1['sift']) and np.array_equal(acc0['counts'], acc1['counts'])
    # a new launch brings back the device's own order until sorted again
    cnt2 = ctx.atom_contacts_launch(5.0, 0.1, False)
    assert cnt2 == cnt
    _same_bag(_host_sorted(ctx.atom_contacts_fetch(cnt2, sort=False)), got)


def test_no_contacts(ctx):
    from helpers import tiny_complex
    ctx.set_complex(tiny_complex([[0, 0, 0], [50, 0, 0]]))
    cnt = ctx.atom_contacts_launch(5.0, 0.1, False)
    assert cnt == 0
    got = ctx.atom_contacts_fetch(cnt, sort=True)
    assert all(len(v) == 0 for v in got.values())
    bags, _ = ctx.fetch_packed()
    assert len(bags['atom_atom']['i']) == 0


def test_global_ids_of_a_shard_sort_as_31_bit_keys(ctx):
    """Contacts of a shard carry global atom ids (arp_set_ownership): the key is built from them."""
    from arpeggio_amd import synth
    pc = synth.config3(5000, seed=11)
    ctx.set_complex(pc)
    rng = np.random.default_rng(5)
    gid = np.sort(rng.choice(2**31 - 2, pc.n_atoms, replace=False)).astype(np.int32)
    ctx.set_ownership(np.ones(pc.n_atoms, np.uint8), gid)
    cnt = ctx.atom_contacts_launch(5.0, 0.1, False)
    raw = {k: v.copy() for k, v in ctx.atom_contacts_fetch(cnt, sort=False).items()}
    got = ctx.atom_contacts_fetch(cnt, sort=True)
    assert cnt > 0 and got['i'].max() > 2**24
    _same_bag(got, _host_sorted(raw))
    ctx.set_complex(pc)      # (drops the ownership)


def test_packed_fetch_equals_the_five_fetches(ctx, capi):
    from arpeggio_amd import synth
    for pc in (synth.proteinlike(), synth.config5(600, 500)):
        ctx.set_complex(pc)
        counts = ctx.run_launch(5.0, 0.1, False, 6.0)
        one_by_one = {'atom_atom': {k: v.copy() for k, v in ctx.atom_contacts_fetch(counts['atom_atom'], sort=True).items()}}
        for name in ('plane_plane', 'atom_plane', 'group_group', 'group_plane'):
            one_by_one[name] = {k: v.copy() for k, v in ctx.fetch_bag(name).items()}
        buf = capi.pinned_empty(64, np.uint8)          # too small on purpose: grown by the call
        bags, buf = ctx.fetch_packed(buf)
        assert set(bags) == set(one_by_one)
        for name, exp in one_by_one.items():
            assert len(exp[next(iter(exp))]) == counts[name]
            for k, v in exp.items():
                g = bags[name][k]
                assert g.dtype == v.dtype and g.shape == v.shape, (name, k)
                assert np.array_equal(g.view(np.uint8), v.view(np.uint8)), (name, k)     # bit-identical, NaN angles included
        # a second call into the same buffer (already sorted on the device) gives the same bytes
        again, buf2 = ctx.fetch_packed(buf)
        assert buf2 is buf
        _same_bag(again['atom_atom'], one_by_one['atom_atom'])


def test_packed_fetch_into_pinned_and_pageable_memory(ctx, capi):
    """The same bytes whatever the host buffer is (page-locked from arp_host_alloc, or plain pageable memory), sorted inside the
    call or before it, and call after call into the same buffer."""
    from arpeggio_amd import synth
    ctx.set_complex(synth.config3(30000, seed=13))
    counts = ctx.run_launch(5.0, 0.1, False, 6.0)
    assert counts['atom_atom'] > 100000
    pinned = capi.pinned_empty(64, np.uint8)
    a, pinned = ctx.fetch_packed(pinned)
    a = {n: {k: v.copy() for k, v in b.items()} for n, b in a.items()}
    key = a['atom_atom']['i'].astype(np.int64) << 32 | a['atom_atom']['j'].astype(np.int64)
    assert np.all(np.diff(key) > 0)
    b, pinned2 = ctx.fetch_packed(pinned)                        # (sorted already)
    assert pinned2 is pinned
    pageable = np.empty(pinned.nbytes, np.uint8)
    ctx.run_launch(5.0, 0.1, False, 6.0)                         # (the device's own order again)
    c_, pageable2 = ctx.fetch_packed(pageable)
    assert pageable2 is pageable
    for other in (b, c_):
        for name, bag in a.items():
            for k, v in bag.items():
                assert np.array_equal(other[name][k].view(np.uint8), v.view(np.uint8)), (name, k)
    for _ in range(10):
        ctx.run_launch(5.0, 0.1, False, 6.0)
        d, _ = ctx.fetch_packed(pinned)
        assert np.array_equal(d['atom_atom']['j'], a['atom_atom']['j']) and np.array_equal(d['atom_atom']['ctype'], a['atom_atom']['ctype'])


@pytest.mark.parametrize('n', [300, 2500, 2900])
def test_small_bags_take_the_one_launch_path_and_agree_with_the_radix_passes(capi, n, monkeypatch):
    """Bags of up to 32 768 records of structures with up to 12 288 atoms are grouped by bgn atom by ONE block (k_sort_small);
    the same bag through the radix passes (ARP_SORT_SMALL=0 is read once per process: a sharded id range forces them instead)."""
    from arpeggio_amd import synth
    pc = synth.config3(n, seed=31)
    c = capi.Context(0)
    try:
        c.set_complex(pc)
        cnt = c.atom_contacts_launch(5.0, 0.1, False)
        assert 0 < cnt <= 32768
        raw = {k: v.copy() for k, v in c.atom_contacts_fetch(cnt, sort=False).items()}
        got = {k: v.copy() for k, v in c.atom_contacts_fetch(cnt, sort=True).items()}
        _same_bag(got, _host_sorted(raw))
        # global ids beyond the small path's counters: the radix passes order the same records
        gid = (np.arange(pc.n_atoms, dtype=np.int64) * 40000 + 7).astype(np.int32)
        c.set_ownership(np.ones(pc.n_atoms, np.uint8), gid)
        cnt2 = c.atom_contacts_launch(5.0, 0.1, False)
        assert cnt2 == cnt
        big = c.atom_contacts_fetch(cnt2, sort=True)
        assert np.array_equal(big['i'], gid[got['i']]) and np.array_equal(big['j'], gid[got['j']])
        assert np.array_equal(big['sift'], got['sift']) and np.array_equal(big['dist'].view(np.uint32), got['dist'].view(np.uint32))
    finally:
        c.close()


def test_ring_and_amide_bags_of_several_thousand_records_are_ordered_on_the_device(ctx, capi):
    """Bags of up to ARP_BAG_SORT_MAX (8192) records leave arp_fetch_packed in their canonical order (k_bag_order: bitonic network
    in LDS, one block per bag); checked against the host order of the same records for bags on both sides of 4096."""
    from arpeggio_amd import synth
    seen = []
    for nr, na, L in ((1500, 1200, 100.0), (1800, 700, 60.0), (900, 700, 60.0)):
        ctx.set_complex(synth.config5(nr, na, L=L))
        counts = ctx.run_launch(5.0, 0.1, False, 6.0)
        bags, _ = ctx.fetch_packed()
        for name, (_, _, order) in ctx._BAGS.items():
            m = counts[name]
            if m == 0:
                continue
            seen.append(m)
            raw = {k: v.copy() for k, v in ctx.fetch_bag(name, sort=False).items()}
            o = np.lexsort((raw[order[1]], raw[order[0]]))
            for k, v in raw.items():
                assert np.array_equal(bags[name][k].view(np.uint8), v[o].view(np.uint8)), (name, k, m)
    assert any(4096 < m <= capi.BAG_SORT_MAX for m in seen) and any(64 < m <= 4096 for m in seen), seen


def test_ring_and_amide_bags_beyond_one_block_are_ordered_on_the_device_too(ctx, capi):
    """Bags of MORE than ARP_BAG_SORT_MAX records (BASELINE configs[4]: 42 k plane-plane records) go through the radix passes of the
    atom-atom bag with the record index as the payload (bag_order_large): the packed fetch delivers them in the reference's creation
    order with no host sort (round 4 left these to np.lexsort)."""
    from arpeggio_amd import synth
    seen = []
    for nr, na, L in ((10_000, 10_000, 100.0), (6000, 9000, 70.0)):
        ctx.set_complex(synth.config5(nr, na, L=L))
        counts = ctx.run_launch(5.0, 0.1, False, 6.0)
        bags, _ = ctx.fetch_packed()
        for name, (_, _, order) in ctx._BAGS.items():
            m = counts[name]
            if m == 0:
                continue
            seen.append(m)
            raw = {k: v.copy() for k, v in ctx.fetch_bag(name, sort=False).items()}
            o = np.lexsort((raw[order[1]], raw[order[0]]))
            for k, v in raw.items():
                assert np.array_equal(bags[name][k].view(np.uint8), v[o].view(np.uint8)), (name, k, m)
    assert sum(m > capi.BAG_SORT_MAX for m in seen) >= 3 and max(seen) > 40_000, seen


def test_sort_enqueued_by_the_pass_itself_gives_the_same_bags(capi):
    """arp_set_sort_after_pass: run_launch enqueues the canonical sort before it returns.  Every way of fetching afterwards —
    one piece, the sorted columns alone, the unsorted bag — gives what a context without the switch gives; a pass that is not
    fetched at all, a change of structure and a ring-heavy structure (bags that outgrow the room reserved behind the sorted
    columns are sorted once more) included."""
    from arpeggio_amd import synth
    a, b = capi.Context(0), capi.Context(0)
    b.set_sort_after_pass(True)
    for pc in (synth.config3(30_000, seed=41), synth.proteinlike(n_res=90, n_waters=30, seed=42), synth.config5(3000, 3000, seed=43, L=60.0),
               synth.config3(2_000, seed=44)):
        a.set_complex(pc); b.set_blob(capi.pack_blob(pc))
        for cutoff in (5.0, 4.0):
            ca, cb = a.run_launch(cutoff, 0.1, False, 6.0), b.run_launch(cutoff, 0.1, False, 6.0)
            assert ca == cb
        b.run_launch(4.5, 0.1, False, 6.0); ca = a.run_launch(4.5, 0.1, False, 6.0)      # (the pass before was never fetched)
        xa, _ = a.fetch_packed()
        xb, _ = b.fetch_packed()
        for bag in xa:
            for k in xa[bag]:
                assert np.array_equal(xa[bag][k], xb[bag][k], equal_nan=(xa[bag][k].dtype.kind == 'f')), (pc.id, bag, k)
        _same_bag(xa['atom_atom'], b.atom_contacts_fetch(ca['atom_atom'], sort=True))
        _same_bag(_host_sorted(a.atom_contacts_fetch(ca['atom_atom'], sort=False)), _host_sorted(b.atom_contacts_fetch(ca['atom_atom'], sort=False)))
    a.close(); b.close()


def test_rows_layout_of_the_packed_fetch_is_the_same_bag(capi):
    """arp_set_packed_layout(ARP_LAYOUT_ROWS): N + 1 row offsets instead of the bgn column — with and without the sort enqueued by
    the pass, for a structure with record-less atoms (explicit hydrogens), a structure without contacts at all, a large and a
    small bag (one-launch sort); switching back and forth on one context; arp_atom_contacts_fetch still hands out records."""
    from arpeggio_amd import synth
    a, b = capi.Context(0), capi.Context(0)
    b.set_packed_layout(rows=True)
    lone = synth.config3(64, seed=45)
    lone.xyz[:] = lone.xyz * 40.0            # nothing within 5 A of anything: an empty bag
    for after in (False, True):
        b.set_sort_after_pass(after)
        for pc in (synth.config3(30_000, seed=41), synth.proteinlike(n_res=90, n_waters=30, seed=42), synth.config3(2_000, seed=44), lone):
            a.set_complex(pc); b.set_complex(pc)
            ca, cb = a.run_launch(5.0, 0.1, False, 6.0), b.run_launch(5.0, 0.1, False, 6.0)
            assert ca == cb
            xa, _ = a.fetch_packed()
            xb, _ = b.fetch_packed()
            row = xb['atom_atom']['row']
            assert 'i' not in xb['atom_atom'] and len(row) == pc.n_atoms + 1 and row[0] == 0 and row[-1] == ca['atom_atom'] and np.all(np.diff(row) >= 0)
            assert np.array_equal(row, np.searchsorted(xa['atom_atom']['i'], np.arange(pc.n_atoms + 1)))
            for k in ('i', 'j', 'dist', 'sift', 'ctype'):
                assert np.array_equal(xa['atom_atom'][k], xb['atom_atom'][k]), (pc.id, k)
            for bag in ('plane_plane', 'atom_plane', 'group_group', 'group_plane'):
                for k in xa[bag]:
                    assert np.array_equal(xa[bag][k], xb[bag][k], equal_nan=(xa[bag][k].dtype.kind == 'f')), (pc.id, bag, k)
            _same_bag(xa['atom_atom'], b.atom_contacts_fetch(ca['atom_atom'], sort=True))      # (records again: the slab is re-made)
            xb2, _ = b.fetch_packed()                                                          # ... and rows again
            assert np.array_equal(xb2['atom_atom']['row'], row) and np.array_equal(xb2['atom_atom']['j'], xa['atom_atom']['j'])
    b.set_packed_layout(rows=False)
    xb, _ = b.fetch_packed()
    assert 'row' not in xb['atom_atom'] and np.array_equal(xb['atom_atom']['i'], xa['atom_atom']['i'])
    a.close(); b.close()
