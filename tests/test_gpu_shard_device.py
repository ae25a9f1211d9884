"""Shards assembled on the device (arp_shard_set_home / _pack_face / _assemble) against the host-side assembly.

One GPU, one context per rank: the face buffers a rank cuts out are handed to its neighbours' contexts as device
pointers — exactly what RCCL delivers in ``sharding.make_shard_device`` — so the whole data path of the multi-GPU set-up
runs here except the transport."""
import numpy as np
import pytest

from arpeggio_amd import sharding, synth

pytestmark = pytest.mark.gpu

@pytest.fixture(scope='module')
def capi():
    from arpeggio_amd import _capi
    return _capi


NAMES = ('plane_plane', 'atom_plane', 'group_group', 'group_plane')


def _structures():
    rng = np.random.default_rng(11)
    slab = synth.slab_config(5000, 3, seed=8)                       # rings, amides, bonds, single-bond neighbours
    sel = np.zeros(slab.n_atoms, np.uint8)
    sel[np.isin(slab.res_id, rng.choice(slab.n_residues, slab.n_residues // 12, replace=False))] = 1
    prot = synth.proteinlike(n_res=400, n_waters=150, seed=5)       # explicit hydrogens (h_xyz runs), real bond graph
    return [('slabs', slab, None), ('slabs_sel', slab, sel), ('proteinlike', prot, None)]


def _assemble_all(capi, full, world, sel, whole):
    ctxs = [capi.Context(0) for _ in range(world)]
    step1 = [sharding.shard_home_to_device(c, full, r, world, sel) for r, c in enumerate(ctxs)]
    shards = []
    for r, c in enumerate(ctxs):
        received = {}
        if r > 0:
            received[-1] = step1[r - 1][0][+1]          # what the left neighbour cut out for its right side
        if r + 1 < world:
            received[+1] = step1[r + 1][0][-1]
        shards.append(sharding.finish_shard_on_device(c, received, step1[r][1], whole_structure=whole))
    return ctxs, shards


@pytest.mark.parametrize('world', [2, 3])
def test_device_assembled_shard_equals_host_assembled(capi, world):
    for name, full, sel in _structures():
        if name == 'proteinlike' and world == 3:
            continue                                     # slabs thinner than the halo
        try:
            sharding._partition(full, world, sharding.halo_width())
        except ValueError:
            continue
        ctxs, shards = _assemble_all(capi, full, world, sel, whole=sel is None)
        ref = capi.Context(0)
        for r, (c, ds) in enumerate(zip(ctxs, shards)):
            hs = sharding.make_shard_local(full, r, world, sel)
            # id maps
            assert np.array_equal(ds.global_id, hs.global_id) and np.array_equal(ds.origin, hs.origin), (name, r)
            assert np.array_equal(ds.is_home, hs.is_home) and np.array_equal(ds.sel, hs.sel)
            assert np.array_equal(ds.ring_gid, hs.ring_gid) and np.array_equal(ds.ring_home, hs.ring_home)
            assert np.array_equal(ds.amide_gid, hs.amide_gid) and np.array_equal(ds.amide_home, hs.amide_home)
            assert np.array_equal(ds.send_left if ds.send_left is not None else [], hs.send_left if hs.send_left is not None else [])
            assert np.array_equal(ds.send_right if ds.send_right is not None else [], hs.send_right if hs.send_right is not None else [])
            assert (ds.origin != 0).sum() > 0
            # the resident arrays, read back
            got = capi.unpack_blob(c.get_blob())
            pc = hs.pc
            assert np.array_equal(got['xyz'], pc.xyz) and np.array_equal(got['rad'][:, 0], pc.vdw) and np.array_equal(got['rad'][:, 1], pc.cov)
            for k, want in (('type_mask', pc.type_mask), ('flags', pc.flags), ('res_id', pc.res_id), ('res_flags', pc.res_flags),
                            ('res_prev', pc.res_prev), ('res_next', pc.res_next), ('bond_off', pc.bond_off), ('bond_idx', pc.bond_idx),
                            ('h_off', pc.h_off), ('h_xyz', pc.h_xyz), ('ring_center', pc.ring_center), ('ring_normal', pc.ring_normal),
                            ('ring_res', pc.ring_res), ('amide_center', pc.amide_center), ('amide_normal', pc.amide_normal),
                            ('amide_res', pc.amide_res)):
                assert np.array_equal(got[k], want), (name, r, k)
            assert np.all(got['sb_nbr'] == -1)
            idx = got['rad_idx']
            known = idx != 0xFFFF
            assert known.mean() > 0.99 and np.array_equal(got['rad_tab'][idx[known]], got['rad'][known])
            hdr = got['header']
            assert all(hdr.lo[k] <= pc.xyz[:, k].min() and hdr.hi[k] >= pc.xyz[:, k].max() for k in range(3))
            # and what the pass makes of it
            sharding.upload_shard(ref, hs, whole_structure=sel is None)
            if sel is None:
                n_dev, n_host = sharding.run_shard_whole_structure(c, sh=ds), sharding.run_shard_whole_structure(ref, sh=hs)
            else:
                n_dev, n_host = c.run_launch(), ref.run_launch()          # local expansion of the shard's selection bits
            assert n_dev == n_host and n_dev['atom_atom'] > 100, (name, r, n_dev, n_host)
            a, b = c.atom_contacts_fetch(n_dev['atom_atom']), ref.atom_contacts_fetch(n_host['atom_atom'])
            for k in a:
                assert np.array_equal(a[k], b[k]), (name, r, k)
            for bag in NAMES:
                x, y = c.fetch_bag(bag), ref.fetch_bag(bag)
                for k in x:
                    assert np.array_equal(x[k], y[k], equal_nan=x[k].dtype.kind == 'f'), (name, r, bag, k)
        for c in ctxs + [ref]:
            c.close()


def test_device_assembly_rejects_bad_buffers(capi):
    full = synth.slab_config(2000, 2, seed=3)
    c0, c1 = capi.Context(0), capi.Context(0)
    f0, b0 = sharding.shard_home_to_device(c0, full, 0, 2)
    f1, b1 = sharding.shard_home_to_device(c1, full, 1, 2)
    with pytest.raises(Exception, match='twice|ascending'):
        c0.shard_assemble(f1[-1], f1[-1], full.n_residues)            # the same halo on both sides: every atom twice
    with pytest.raises(Exception):
        c0.shard_assemble((f1[-1][0], 100), None, full.n_residues)    # truncated buffer
    with pytest.raises(Exception):
        c0.shard_assemble(f1[-1], None, 3)                            # residue ids beyond the table
    with pytest.raises(Exception):
        capi.Context(0).shard_pack_face(0, 0.0, 1.0)                  # no home records
    ds = sharding.finish_shard_on_device(c0, {+1: f1[-1]}, b0, whole_structure=True)     # and the good call still works
    assert ds.n_atoms > 2000 and sharding.run_shard_whole_structure(c0, sh=ds)['atom_atom'] > 0
    with pytest.raises(ValueError, match='halo'):
        sharding.run_shard_whole_structure(c0, cutoff=7.5, sh=ds)
    c0.close(); c1.close()


def test_one_rank_communicator_through_the_c_abi(capi):
    """arp_comm_* with a one-rank RCCL communicator (the boxes of this pool have one GPU): librccl.so is loaded, the
    communicator is created from a unique id, the whole driver of a device-assembled shard runs (no neighbours: no faces
    travel), and the three-stage pass with the per-pass exchange calls — gather / grouped send-recv / scatter with no
    peers, ncclAllReduce over one rank on the context's stream — equals the single-context pass."""
    full = synth.slab_config(4000, 2, seed=6)
    one = capi.Context(0)
    one.set_complex(full)
    n1 = one.run_launch()
    c1 = capi.Context(0)
    c1.comm_init(0, 1, capi.Context.comm_unique_id())
    ds = sharding.make_shard_device(c1, full, 0, 1, whole_structure=True)
    assert ds.n_atoms == full.n_atoms and sharding.run_shard_whole_structure(c1, sh=ds) == n1
    sel = (full.res_id % 7 == 3).astype(np.uint8)
    c2 = capi.Context(0)
    c2.comm_init(0, 1, capi.Context.comm_unique_id())
    ds2 = sharding.make_shard_device(c2, full, 0, 1, sel=sel)
    ex = sharding.DeviceExchange(c2, ds2)
    one.set_selection(sel)
    exp = one.run_launch()
    assert sharding.run_shard_device(c2, ex) == exp
    got, ref = c2.atom_contacts_fetch(exp['atom_atom']), one.atom_contacts_fetch(exp['atom_atom'])
    for k in ('i', 'j', 'sift', 'ctype'):
        assert np.array_equal(got[k], ref[k]), k
    with pytest.raises(Exception):
        c2.comm_init(0, 1, capi.Context.comm_unique_id())      # a second communicator on the same context is refused
    c1.comm_destroy(); c2.comm_destroy()
    for c in (c1, c2, one):
        c.close()


@pytest.mark.parametrize('seed', range(6))
def test_random_shards_device_vs_host(capi, seed):
    """Random structure sizes, rank counts and selections: every device-assembled shard gives the results of the host-assembled
    one, and the union over the ranks is the single-context result."""
    rng = np.random.default_rng(500 + seed)
    world = int(rng.integers(2, 5))
    full = synth.slab_config(int(rng.integers(1500, 6000)), world, seed=int(rng.integers(1, 100)))
    whole = bool(rng.integers(0, 2))
    sel = None
    if not whole:
        sel = (rng.random(full.n_residues) < rng.choice([0.05, 0.4, 0.9]))[full.res_id].astype(np.uint8)
        sel[0] = 1
    ctxs, shards = _assemble_all(capi, full, world, sel, whole=whole)
    ref = capi.Context(0)
    one = capi.Context(0)
    one.set_complex(full)
    if sel is not None:
        one.set_selection(sel)
    n_one = one.run_launch()
    keys = []
    for r, (c, ds) in enumerate(zip(ctxs, shards)):
        hs = sharding.make_shard_local(full, r, world, sel)
        assert np.array_equal(ds.global_id, hs.global_id) and np.array_equal(ds.origin, hs.origin)
        sharding.upload_shard(ref, hs, whole_structure=whole)
        n_dev, n_host = c.run_launch(), ref.run_launch()
        assert n_dev == n_host, (seed, r, n_dev, n_host)
        a, b = c.atom_contacts_fetch(n_dev['atom_atom']), ref.atom_contacts_fetch(n_host['atom_atom'])
        assert all(np.array_equal(a[k], b[k]) for k in a)
        keys.append(a['i'].astype(np.int64) * full.n_atoms + a['j'])
    if whole:      # (with a partial selection the shards would first have to exchange selection_plus: covered elsewhere)
        want = one.atom_contacts_fetch(n_one['atom_atom'])
        assert np.array_equal(np.sort(np.concatenate(keys)), want['i'].astype(np.int64) * full.n_atoms + want['j'])
    for c in ctxs + [ref, one]:
        c.close()


def test_rank_with_an_empty_slab(capi):
    """Two bodies 200 A apart cut into three slabs: the middle rank owns nothing and receives nothing.  Its assembled shard
    is an empty structure (zero atoms, zero contacts) on both the device and the host path, and the other two ranks give the
    single-context result between them."""
    from helpers import concat_packs
    full = concat_packs(synth.config3(3000, seed=1), synth.config3(3000, seed=2), shift=(200.0, 0.0, 0.0))
    full.validate()
    ctxs, shards = _assemble_all(capi, full, 3, None, whole=True)
    assert shards[1].n_atoms == 0 and sharding.make_shard_local(full, 1, 3, None).pc.n_atoms == 0
    one = capi.Context(0)
    one.set_complex(full)
    n_one = one.run_launch()
    want = one.atom_contacts_fetch(n_one['atom_atom'])
    keys, planes = [], 0
    for c in ctxs:
        n = c.run_launch()
        g = c.atom_contacts_fetch(n['atom_atom'])
        keys.append(g['i'].astype(np.int64) * full.n_atoms + g['j'])
        planes += sum(n[k] for k in NAMES)
    assert len(keys[1]) == 0
    assert np.array_equal(np.sort(np.concatenate(keys)), want['i'].astype(np.int64) * full.n_atoms + want['j'])
    assert planes == sum(n_one[k] for k in NAMES) > 0
    for c in ctxs + [one]:
        c.close()


def test_shards_from_per_rank_records_equal_shards_from_the_whole_structure(capi):
    """bench.py --gpus N: every rank makes the records of its own slab (synth.slab_home_records) and never holds the whole
    structure; the shard assembled from them on the device — id maps, resident arrays, the contacts of a whole-structure pass —
    equals the one cut out of slab_config(...)."""
    n_per_slab, world = 6000, 3
    full = synth.slab_config(n_per_slab, world, seed=4)
    ctxs, shards = _assemble_all(capi, full, world, None, True)
    ctxs2 = [capi.Context(0) for _ in range(world)]
    step1 = [sharding.shard_records_to_device(c, *synth.slab_home_records(n_per_slab, world, r, seed=4), r, world) for r, c in enumerate(ctxs2)]
    for r, c in enumerate(ctxs2):
        received = {}
        if r > 0:
            received[-1] = step1[r - 1][0][+1]
        if r + 1 < world:
            received[+1] = step1[r + 1][0][-1]
        ds = sharding.finish_shard_on_device(c, received, step1[r][1], whole_structure=True)
        ref = shards[r]
        for k in ('global_id', 'origin', 'is_home', 'sel', 'ring_gid', 'ring_home', 'amide_gid', 'amide_home'):
            assert np.array_equal(getattr(ds, k), getattr(ref, k)), (r, k)
        for k in ('send_left', 'send_right'):
            x, y = getattr(ds, k), getattr(ref, k)
            assert (x is None) == (y is None) and (x is None or np.array_equal(x, y)), (r, k)
        got, want = capi.unpack_blob(c.get_blob()), capi.unpack_blob(ctxs[r].get_blob())
        for k in ('xyz', 'rad', 'type_mask', 'flags', 'res_id', 'res_flags', 'res_prev', 'res_next', 'bond_off', 'bond_idx', 'h_off', 'h_xyz',
                  'ring_center', 'ring_normal', 'ring_res', 'amide_center', 'amide_normal', 'amide_res'):
            assert np.array_equal(got[k], want[k]), (r, k)
        n_a, n_b = sharding.run_shard_whole_structure(c, sh=ds), sharding.run_shard_whole_structure(ctxs[r], sh=ref)
        assert n_a == n_b and n_a['atom_atom'] > 100
        a, b = c.atom_contacts_fetch(n_a['atom_atom']), ctxs[r].atom_contacts_fetch(n_b['atom_atom'])
        for k in a:
            assert np.array_equal(a[k], b[k]), (r, k)
    for c in ctxs + ctxs2:
        c.close()
    # the one-call form bench.py uses, on a world of one (no neighbour, no exchange)
    one, whole = capi.Context(0), capi.Context(0)
    ds = sharding.make_shard_device(one, synth.slab_home_records(n_per_slab, 1, 0, seed=4), 0, 1, whole_structure=True)
    whole.set_complex(synth.slab_config(n_per_slab, 1, seed=4))
    n_a, n_b = sharding.run_shard_whole_structure(one, sh=ds), whole.run_launch()
    assert dict(n_a) == dict(n_b) and n_a['atom_atom'] > 100
    a, b = one.atom_contacts_fetch(n_a['atom_atom']), whole.atom_contacts_fetch(n_b['atom_atom'])
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    one.close(); whole.close()
