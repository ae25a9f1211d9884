"""Edge cases through the C ABI on the GPU: empty / tiny / degenerate inputs, dense cells (> 64 atoms per cell),
far-apart atoms, buffer regrowth, capacity and call-order errors."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def capi():
    from arpeggio_amd import _capi
    return _capi


@pytest.fixture()
def ctx(capi):
    c = capi.Context(0)
    yield c
    c.close()


def _check(ctx, pc, sel=None, brute=True):
    import oracle
    ctx.set_complex(pc)
    masks = ctx.make_selection(sel)
    oc = oracle.OracleComplex(pc)
    plus = oc.make_selection(sel, use_grid=not brute)
    assert np.array_equal(masks['plus'], plus)
    got = ctx.atom_contacts()
    exp = oc.atom_contacts(use_grid=not brute)
    assert len(got['i']) == len(exp['i'])
    for k in ('i', 'j', 'sift', 'ctype'):
        assert np.array_equal(got[k], exp[k]), k
    assert np.array_equal(got['dist'].view(np.uint32), exp['dist'].view(np.uint32))
    return got


def test_empty_and_tiny_structures(ctx):
    from helpers import tiny_complex
    got = _check(ctx, tiny_complex(np.zeros((0, 3), np.float32)))
    assert len(got['i']) == 0
    assert ctx.run_launch() == dict(atom_atom=0, plane_plane=0, atom_plane=0, group_group=0, group_plane=0)
    assert len(_check(ctx, tiny_complex([[1, 2, 3]]))['i']) == 0
    assert len(_check(ctx, tiny_complex([[1, 2, 3], [1, 2, 3]]))['i']) == 1          # coincident atoms: distance 0, clash
    gi, gj = ctx.search_all(5.0)
    assert (gi.tolist(), gj.tolist()) == ([0], [1])


def test_all_atoms_in_one_cell_more_than_a_wavefront(ctx):
    """700 atoms inside a 4 A box: one home cell with hundreds of atoms (several 64-atom home chunks, j > h logic)."""
    from helpers import random_dense_pack
    pc = random_dense_pack(21, n=700, box=4.0)
    got = _check(ctx, pc)
    assert len(got['i']) > 50_000


def test_two_thousand_atoms_in_one_cell(ctx):
    """2000 atoms inside a 4 A box: one home cell whose home blocks (32 atoms x <= 1024 candidates) are units several waves and
    blocks claim — the clump that used to be a serial chain on one wave; ~1.9 M pairs, runs of ~1000 records per bgn atom in
    the device sort (far beyond its 64-record LDS window)."""
    from helpers import random_dense_pack
    pc = random_dense_pack(22, n=2000, box=4.0)
    got = _check(ctx, pc)
    assert len(got['i']) > 1_500_000
    # the same pass once more (the by-atoms split chosen from the first pass's tests per atom) gives the same records
    again = ctx.atom_contacts()
    assert np.array_equal(again['i'], got['i']) and np.array_equal(again['j'], got['j']) and np.array_equal(again['sift'], got['sift'])


def test_identical_coordinates_many_atoms(ctx):
    from arpeggio_amd.core import config
    from helpers import tiny_complex
    n = 200
    pc = tiny_complex(np.full((n, 3), 7.5, np.float32), type_mask=config.ATOM_TYPE_BIT['hydrophobe'])
    got = _check(ctx, pc)
    assert len(got['i']) == n * (n - 1) // 2 and np.all(got['dist'] == 0)


def test_far_apart_and_large_coordinates(ctx):
    from helpers import tiny_complex
    rng = np.random.default_rng(4)
    # clusters 5,000 A apart: the grid hits its cell cap and has to coarsen
    centres = np.array([[0, 0, 0], [5000, 0, 0], [0, 5000, 0], [0, 0, 5000], [-5000, -5000, -5000]], np.float32)
    xyz = (centres[:, None, :] + rng.normal(0, 3.0, (5, 60, 3))).reshape(-1, 3).astype(np.float32)
    got = _check(ctx, tiny_complex(xyz))
    assert len(got['i']) > 100
    # large offsets: float32 coordinates around 1e4 (spacing ~1e-3)
    xyz2 = (rng.random((400, 3)) * 20 + 12345.0).astype(np.float32)
    _check(ctx, tiny_complex(xyz2))
    xyz3 = (rng.random((400, 3)) * 20 - 9876.0).astype(np.float32)
    _check(ctx, tiny_complex(xyz3))


def test_pair_and_bag_buffers_regrow(ctx):
    """A small structure sizes the buffers; a dense one must overflow them and trigger the re-run."""
    import oracle
    from arpeggio_amd import synth
    from helpers import random_dense_pack
    small = synth.config3(300, seed=2)
    ctx.set_complex(small)
    ctx.run_launch()
    dense = random_dense_pack(5, n=900, box=9.0)                   # ~400k pairs within 5 A  >>  900 * 16 + 8192
    got = _check(ctx, dense)
    assert len(got['i']) > 100_000
    rings = synth.make_synthetic(0, seed=3, box=(9.0, 9.0, 9.0), n_rings=400, n_amides=400)   # ~80k ring pairs >> 400 * 16 + 256
    ctx.set_complex(rings)
    counts = ctx.run_launch()
    oc = oracle.OracleComplex(rings)
    oc.make_selection(None, use_grid=False)
    assert counts['plane_plane'] == len(oc.plane_plane()['bgn']) > 400 * 16 + 256
    assert counts['group_group'] == len(oc.group_group()['bgn'])
    assert counts['group_plane'] == len(oc.group_plane()['amide'])


def test_capacity_and_call_order_errors(capi):
    from arpeggio_amd import synth
    L = capi.load()
    c = capi.Context(0)
    pc = synth.config3(2000, seed=1)
    i4, f4 = np.zeros(4, np.int32), np.zeros(4, np.float32)
    u2, u1 = np.zeros(4, np.uint16), np.zeros(4, np.uint8)
    cnt = C.c_int64(0)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    # fetch before any launch
    rc = L.arp_atom_contacts_fetch(c._h, 4, p(i4), p(i4), p(f4), p(u2), p(u1), C.byref(cnt))
    assert rc == capi.ARP_E_ARG and b'no launch' in L.arp_last_error(c._h)
    c.set_complex(pc)
    n = c.atom_contacts_launch()
    assert n > 4
    rc = L.arp_atom_contacts_fetch(c._h, 4, p(i4), p(i4), p(f4), p(u2), p(u1), C.byref(cnt))
    assert rc == capi.ARP_E_CAPACITY and cnt.value == n            # required count reported
    d4 = np.zeros(4, np.float64)
    rc = L.arp_plane_plane(c._h, 1, p(i4), p(i4), p(d4), p(d4), p(d4), p(d4), p(u1), p(u1), p(u1), C.byref(cnt))
    assert rc in (capi.ARP_OK, capi.ARP_E_CAPACITY)
    if rc == capi.ARP_E_CAPACITY:
        assert cnt.value > 1
    # bad arguments
    assert L.arp_search_all(c._h, -1.0, None, 4, p(i4), p(i4), C.byref(cnt)) == capi.ARP_E_ARG
    assert L.arp_run_stage(c._h, 7, 5.0, 0.1, 0, 6.0, None) == capi.ARP_E_ARG
    bad = np.array([5, 3, 1], np.int32)
    with pytest.raises(ValueError):
        c.set_ownership(np.ones(pc.n_atoms, np.uint8), np.arange(pc.n_atoms, dtype=np.int32)[::-1].copy())
    with pytest.raises(capi.NativeLibraryError):
        capi.Context(99)
    c.close()


def test_selection_of_a_single_atom_and_of_nothing_nearby(ctx):
    from helpers import tiny_complex
    xyz = np.array([[0, 0, 0], [3, 0, 0], [5.9, 0, 0], [12, 0, 0], [40, 40, 40]], np.float32)
    pc = tiny_complex(xyz)
    sel = np.array([1, 0, 0, 0, 0], np.uint8)
    got = _check(ctx, pc, sel)
    assert got['ctype'].tolist().count(2) >= 1                     # INTER contacts with the selected atom
    sel = np.array([0, 0, 0, 0, 1], np.uint8)                      # isolated atom: selection_plus == selection, no contacts
    got = _check(ctx, pc, sel)
    assert len(got['i']) == 0


def test_more_distinct_radii_than_the_table_holds(ctx):
    """The sift record carries an index into a 256-entry {vdw, cov} table; atoms beyond it read the uploaded radii."""
    from helpers import random_dense_pack
    pc = random_dense_pack(33, n=900, box=16.0)
    rng = np.random.default_rng(5)
    pc.vdw = (np.asarray(pc.vdw) + rng.integers(0, 700, pc.n_atoms) * 1e-4).astype(np.float64)   # ~700 distinct pairs
    pc.cov = (np.asarray(pc.cov) + rng.integers(0, 3, pc.n_atoms) * 1e-3).astype(np.float64)
    assert len({(a, b) for a, b in zip(pc.vdw.tolist(), pc.cov.tolist())}) > 256
    _check(ctx, pc)


def test_atoms_with_hundreds_of_hydrogens_and_bonds(ctx):
    """Counts are stored saturated at 255 in the record; beyond that the CSR offsets are read."""
    from helpers import tiny_complex
    from arpeggio_amd.core import config
    rng = np.random.default_rng(9)
    n = 320
    xyz = (rng.random((n, 3)) * 9.0).astype(np.float32)
    tm = np.zeros(n, np.uint16)
    tm[0] = config.ATOM_TYPE_BIT['hbond donor'] | config.ATOM_TYPE_BIT['weak hbond donor']
    tm[1:] = config.ATOM_TYPE_BIT['hbond acceptor']
    v = rng.standard_normal((300, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    h = {0: (xyz[0].astype(np.float64) + v).tolist()}                       # 300 hydrogens on atom 0
    bonds = [(0, j) for j in range(1, 301)]                                 # and 300 bonds
    pc = tiny_complex(xyz, type_mask=tm, res_id=np.arange(n, dtype=np.int32), bonds=bonds, h=h)
    got = _check(ctx, pc)
    first = (got['i'] == 0)
    cov_bit, hb_bit = 1 << config.SIFT_NAMES.index('covalent'), 1 << config.SIFT_NAMES.index('hbond')
    assert first.any() and (got['sift'][first] & cov_bit).any() and (got['sift'][first] & hb_bit).any()


def test_malformed_inputs_are_refused_before_any_kernel_runs(capi):
    """Non-finite coordinates (no grid cell), broken CSR offsets and out-of-range residue indices come back as
    ARP_E_ARG from the setters instead of reaching a kernel that indexes with them."""
    L = capi.load()
    h = C.c_void_p()
    assert L.arp_create(0, C.byref(h)) == 0
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    n = 4
    xyz = np.zeros((n, 3), np.float32)
    vdw, cov = np.full(n, 1.7), np.full(n, 0.76)
    tm, fl, res = np.zeros(n, np.uint16), np.zeros(n, np.uint16), np.arange(n, dtype=np.int32)
    bad = xyz.copy()
    bad[2, 1] = np.nan
    assert L.arp_set_atoms(h, n, p(bad), p(vdw), p(cov), p(tm), p(fl), p(res)) == -1
    bad[2, 1] = np.inf
    assert L.arp_set_atoms(h, n, p(bad), p(vdw), p(cov), p(tm), p(fl), p(res)) == -1
    neg = res.copy()
    neg[1] = -3
    assert L.arp_set_atoms(h, n, p(xyz), p(vdw), p(cov), p(tm), p(fl), p(neg)) == -1
    assert L.arp_set_atoms(h, n, p(xyz), p(vdw), p(cov), p(tm), p(fl), p(res)) == 0
    rf, prv, nxt = np.zeros(2, np.uint8), np.full(2, -1, np.int32), np.full(2, -1, np.int32)
    assert L.arp_set_residues(h, 2, p(rf), p(prv), p(nxt)) == -1            # atoms refer to residues 0..3
    off = np.array([0, 2, 1, 3, 3], np.int32)                                # decreasing
    assert L.arp_set_bonds(h, p(off), p(np.zeros(3, np.int32))) == -1
    assert L.arp_set_hydrogens(h, p(off), p(np.zeros(9, np.float64))) == -1
    ctr = np.zeros((1, 3), np.float64)
    ctr[0, 0] = np.nan
    assert L.arp_set_rings(h, 1, p(ctr), p(np.zeros((1, 3))), p(np.zeros(1, np.int32))) == -1
    assert b'non-finite' in L.arp_last_error(h)
    # without a residue table the selection sets cannot be built: clean error, not an out-of-bounds write
    counts = np.zeros(5, np.int64)
    assert L.arp_run_launch(h, C.c_double(5.0), C.c_double(0.1), 0, C.c_double(6.0), p(counts)) == -1
    L.arp_destroy(h)


@pytest.mark.parametrize('box', [165.0, 260.0, 402.0])
def test_sparse_structures_whose_grids_need_the_tiled_scan(ctx, box):
    """A few thousand atoms in a large box: tens of thousands of (mostly empty) cells, histogram scan in several
    16384-cell tiles (3, 10 and ~33 tiles) with clusters so that there are contacts to compare."""
    from helpers import random_dense_pack
    pc = random_dense_pack(41, n=1500, box=14.0)
    rng = np.random.default_rng(int(box))
    shift = (rng.random((pc.n_atoms // 50 + 1, 3)) * (box - 14.0)).astype(np.float32)     # 50-atom clusters
    pc.xyz = (np.asarray(pc.xyz) * 0.35 + shift[np.arange(pc.n_atoms) // 50]).astype(np.float32)
    pc.h_xyz = np.asarray(pc.h_xyz).reshape(-1, 3)
    par = np.repeat(np.arange(pc.n_atoms), np.diff(pc.h_off))
    pc.h_xyz = ((pc.h_xyz * 0.35) + shift[par // 50]).reshape(pc.h_xyz.shape)
    got = _check(ctx, pc, brute=False)
    assert len(got['i']) > 1000


def test_many_random_small_structures(ctx):
    """60 random soups of 1..500 atoms in boxes of 3..80 A (from one crowded cell to mostly empty grids), random
    selections, every atom type / flag, bonds, hydrogens: selection_plus, contacts, SIFts and distances against the
    brute-force oracle.  Exercises partial home blocks, grid borders, the start-table batches and the task queues."""
    import oracle
    from helpers import random_dense_pack
    rng = np.random.default_rng(2024)
    total = 0
    for case in range(60):
        n = int(rng.integers(1, 500))
        box = float(rng.choice([3.0, 6.0, 9.5, 14.0, 22.0, 37.0, 80.0]))
        pc = random_dense_pack(1000 + case, n=max(n, 4), box=box)
        if case % 5 == 2:      # duplicated coordinates
            k = max(1, pc.n_atoms // 10)
            pc.xyz[rng.integers(0, pc.n_atoms, k)] = pc.xyz[rng.integers(0, pc.n_atoms, k)]
        elif case % 5 == 3:    # atoms on a 1.25 A lattice: many equal distances, pairs exactly at the cut-offs
            pc.xyz = (np.round(pc.xyz / 1.25) * 1.25).astype(np.float32)
        sel = None if case % 3 == 0 else (rng.random(pc.n_atoms) < rng.choice([0.02, 0.3, 0.9])).astype(np.uint8)
        if sel is not None and sel.sum() == 0:
            sel[int(rng.integers(0, pc.n_atoms))] = 1
        ctx.set_complex(pc)
        masks = ctx.make_selection(sel)
        oc = oracle.OracleComplex(pc)
        plus = oc.make_selection(sel, use_grid=False)
        assert np.array_equal(masks['plus'], plus), case
        got = ctx.atom_contacts()
        exp = oc.atom_contacts(use_grid=False)
        assert len(got['i']) == len(exp['i']), case
        for k in ('i', 'j', 'sift', 'ctype'):
            assert np.array_equal(got[k], exp[k]), (case, k)
        assert np.array_equal(got['dist'].view(np.uint32), exp['dist'].view(np.uint32)), case
        # the one-wait pass gives the same list
        if sel is not None:
            ctx.set_selection(sel)
        counts = ctx.run_launch()
        again = ctx.atom_contacts_fetch(counts['atom_atom'])
        assert np.array_equal(again['i'], exp['i']) and np.array_equal(again['sift'], exp['sift']), case
        total += len(exp['i'])
    assert total > 100_000


def test_two_contexts_on_two_host_threads(capi):
    """One arp_ctx per host thread: two threads run different structures concurrently on one GPU and
    each gets exactly what a lone run gives."""
    import threading
    from helpers import random_dense_pack
    packs = [random_dense_pack(71, n=1800, box=28.0), random_dense_pack(72, n=2300, box=31.0)]
    lone = []
    for pc in packs:
        c = capi.Context(0)
        c.set_complex(pc)
        k = c.run_launch()
        lone.append((k, c.atom_contacts_fetch(k['atom_atom'])))
        c.close()
    results, errors = [None, None], []

    def work(t):
        try:
            c = capi.Context(0)
            c.set_complex(packs[t])
            for _ in range(40):
                k = c.run_launch()
            results[t] = (k, c.atom_contacts_fetch(k['atom_atom']))
            c.close()
        except Exception as exc:      # surfaces in the main thread below
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for t in range(2):
        assert results[t][0] == lone[t][0]
        for f in ('i', 'j', 'sift', 'ctype'):
            assert np.array_equal(results[t][1][f], lone[t][1][f]), (t, f)
        assert np.array_equal(results[t][1]['dist'].view(np.uint32), lone[t][1]['dist'].view(np.uint32))


def test_random_ring_and_amide_sets(ctx):
    """40 random ring / amide sets on random soups — coincident centres, zero normals (NaN angles -> class ''), rings
    without a residue, whole and partial selections — through arp_run_launch: ids, classes, masks and contact types
    exactly, distances and angles within 1e-4 degrees / Angstrom, the north-star bound (acos differs by an ulp between the two libms)."""
    import oracle
    from helpers import random_dense_pack
    rng = np.random.default_rng(99)

    def close(a, b, tol=1e-4):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b)) and (np.nan_to_num(np.abs(a - b)) <= tol).all()

    bags = (('plane_plane', ('bgn', 'end'), ('bgn', 'end', 'type1', 'type2', 'ctype'), ('dist', 'dihedral', 'theta_bgn', 'theta_end')),
            ('atom_plane', ('ring', 'atom'), ('atom', 'ring', 'mask', 'ctype'), ('dist', 'theta')),
            ('group_group', ('bgn', 'end'), ('bgn', 'end', 'ctype'), ('dist', 'dihedral', 'theta')),
            ('group_plane', ('amide', 'ring'), ('amide', 'ring', 'ctype'), ('dist', 'dihedral', 'theta')))
    records = 0
    for case in range(40):
        nr, na = int(rng.integers(1, 300)), int(rng.integers(1, 300))
        box = float(rng.choice([4.0, 9.0, 20.0, 45.0]))
        pc = random_dense_pack(9000 + case, n=int(rng.integers(50, 600)), box=box)
        rc, rn = rng.random((nr, 3)) * box, rng.standard_normal((nr, 3))
        ac, an = (rng.random((na, 3)) * box).astype(np.float32), rng.standard_normal((na, 3)).astype(np.float32)
        if case % 4 == 1:
            rc[rng.integers(0, nr, max(1, nr // 5))] = rc[rng.integers(0, nr, max(1, nr // 5))]
            rn[rng.integers(0, nr, max(1, nr // 8))] = 0.0
        if case % 4 == 2:
            an[rng.integers(0, na, max(1, na // 8))] = 0.0
        pc.ring_center, pc.ring_normal = rc, rn
        pc.ring_res = rng.integers(-1, pc.n_residues, nr).astype(np.int32)
        pc.ring_atoms = []
        pc.amide_center, pc.amide_normal = ac, an
        pc.amide_res = rng.integers(-1, pc.n_residues, na).astype(np.int32)
        pc.amide_atoms = np.full((na, 4), -1, np.int32)
        sel = None if case % 3 == 0 else (rng.random(pc.n_atoms) < 0.3).astype(np.uint8)
        if sel is not None and sel.sum() == 0:
            sel[0] = 1
        ctx.set_complex(pc)
        if sel is not None:
            ctx.set_selection(sel)
        ctx.run_launch()
        oc = oracle.OracleComplex(pc)
        oc.make_selection(sel, use_grid=False)
        for name, order, exact, tol in bags:
            g, e = ctx.fetch_bag(name), getattr(oc, name)()
            assert len(g[exact[0]]) == len(e[exact[0]]), (case, name)
            o = np.lexsort((e[order[1]], e[order[0]]))
            for k in exact:
                assert np.array_equal(g[k], e[k][o]), (case, name, k)
            for k in tol:
                assert close(g[k], e[k][o]), (case, name, k)
            records += len(o)
    assert records > 20_000


@pytest.mark.gpu
def test_blob_upload_equals_classic_upload_and_rejects_bad_input():
    """arp_set_blob (one packed copy + device-side validation) leaves the context exactly as the classic setters do; the
    checks those setters make on the host are made on the device and turn into the same ValueError."""
    from arpeggio_amd import _capi, synth
    pc = synth.config3(20_000, seed=9)
    a, b = _capi.Context(0), _capi.Context(0)
    a.set_complex(pc)
    blob = _capi.pack_blob(pc)
    b.set_blob(blob)
    sel = np.zeros(pc.n_atoms, np.uint8)
    sel[pc.res_id % 7 == 0] = 1
    for s in (None, sel):
        if s is not None:
            a.set_selection(s); b.set_selection(s)
        ca, cb = a.run_launch(5.0, 0.1, False, 6.0), b.run_launch(5.0, 0.1, False, 6.0)
        assert ca == cb
        ra, rb = a.atom_contacts_fetch(ca['atom_atom']), b.atom_contacts_fetch(cb['atom_atom'])
        for k in ra:
            assert np.array_equal(ra[k], rb[k]), k
        for bag in ('plane_plane', 'atom_plane', 'group_group', 'group_plane'):
            xa, xb = a.fetch_bag(bag), b.fetch_bag(bag)
            for k in xa:
                assert np.array_equal(xa[k], xb[k], equal_nan=(xa[k].dtype.kind == 'f')), (bag, k)
    # a second structure into the same context, then back: views are re-pointed, nothing stale survives
    pc2 = synth.proteinlike(n_res=60, n_waters=30)
    b.set_blob(_capi.pack_blob(pc2))
    a.set_complex(pc2)
    assert a.run_launch(5.0, 0.1, False, 6.0) == b.run_launch(5.0, 0.1, False, 6.0)
    # corrupted blobs
    hdr_size = 512
    def broken(mutate):
        bad = _capi.pack_blob(pc2)
        h = _capi.BlobHeader.from_buffer(bad)
        mutate(bad, h)
        del h
        return bad
    def nan_coord(bad, h): np.frombuffer(bad, np.float32, 4, int(h.off[0]))[1] = np.nan
    def res_range(bad, h): np.frombuffer(bad, np.int32, 1, int(h.off[4]))[0] = int(h.nres)
    def bond_range(bad, h): np.frombuffer(bad, np.int32, 1, int(h.off[9]))[0] = int(h.n)
    def csr(bad, h): np.frombuffer(bad, np.int32, 3, int(h.off[10]))[1] = 10**6
    def outside(bad, h): h.hi[0] = h.lo[0]
    def magic(bad, h): h.magic = 1
    def counts(bad, h): h.n = h.n + 1
    for m in (nan_coord, res_range, bond_range, csr, outside, magic, counts):
        with pytest.raises(ValueError):
            b.set_blob(broken(m))
    b.set_blob(_capi.pack_blob(pc2))                      # and the context is still usable
    assert a.run_launch(5.0, 0.1, False, 6.0) == b.run_launch(5.0, 0.1, False, 6.0)
    a.close(); b.close()


def _five_bags(ctx, counts):
    out = {'atom_atom': ctx.atom_contacts_fetch(counts['atom_atom'])}
    for bag in ('plane_plane', 'atom_plane', 'group_group', 'group_plane'):
        out[bag] = ctx.fetch_bag(bag)
    return out


def _same_bags(x, y, what):
    for bag in x:
        for k in x[bag]:
            assert np.array_equal(x[bag][k], y[bag][k], equal_nan=(x[bag][k].dtype.kind == 'f')), (what, bag, k)


@pytest.mark.gpu
def test_lists_made_with_the_upload_survive_whatever_follows_the_upload():
    """arp_set_blob builds the centre grids and the ring / amide candidate lists of the new structure on the second stream while the
    host waits for the validation (the atom-plane list atom by atom against the ring grid).  Whatever the caller does next — another
    upload before any pass, classic setters on top of the blob, a batch declared over it, a partial selection — the five bags are
    those of a context set up by the classic setters (whose lists are built inside its first pass)."""
    import dataclasses
    from arpeggio_amd import _capi, batch, synth
    rich = synth.make_synthetic(6000, seed=21, box=(60.0, 60.0, 60.0), n_rings=1500, n_amides=1500, id='rich')     # rings every few angstroms
    plain = synth.config3(20_000, seed=22)
    prot = synth.proteinlike(n_res=80, n_waters=40, seed=23)
    a, b = _capi.Context(0), _capi.Context(0)
    # 1. one upload after the other, the first one never evaluated: its lists are abandoned half-way
    for first, second in ((plain, rich), (rich, prot), (prot, plain)):
        b.set_blob(_capi.pack_blob(first))
        b.set_blob(_capi.pack_blob(second))
        a.set_complex(second)
        ca, cb = a.run_launch(5.0, 0.1, False, 6.0), b.run_launch(5.0, 0.1, False, 6.0)
        assert ca == cb, (first.id, second.id)
        _same_bags(_five_bags(a, ca), _five_bags(b, cb), second.id)
    # ... and more centres than the one-launch centre grids take (16 384): the general grid build runs on the upload's stream
    many = synth.make_synthetic(3000, seed=24, box=(90.0, 90.0, 90.0), n_rings=17_000, n_amides=300, id='many')
    b.set_blob(_capi.pack_blob(many)); a.set_complex(many)
    ca, cb = a.run_launch(5.0, 0.1, False, 6.0), b.run_launch(5.0, 0.1, False, 6.0)
    assert ca == cb and ca['plane_plane'] > 1000
    _same_bags(_five_bags(a, ca), _five_bags(b, cb), 'many rings')
    # 2. classic setters on top of a blob: new rings and amides (the same atoms) stale the lists of the upload
    b.set_blob(_capi.pack_blob(rich))
    moved = dataclasses.replace(rich, ring_center=rich.ring_center[::-1].copy(), ring_normal=rich.ring_normal[::-1].copy(), ring_res=rich.ring_res[::-1].copy(),
                                amide_center=(rich.amide_center + np.float32(0.75)).astype(np.float32))
    b._check(b._L.arp_set_rings(b._h, moved.n_rings, _capi._p(moved.ring_center), _capi._p(moved.ring_normal), _capi._p(moved.ring_res)), 'arp_set_rings')
    b._check(b._L.arp_set_amides(b._h, moved.n_amides, _capi._p(moved.amide_center), _capi._p(moved.amide_normal), _capi._p(moved.amide_res)), 'arp_set_amides')
    b._keep = (b._keep, moved)
    a.set_complex(moved)
    ca, cb = a.run_launch(5.0, 0.1, False, 6.0), b.run_launch(5.0, 0.1, False, 6.0)
    assert ca == cb and ca['plane_plane'] > 0 and ca['atom_plane'] > 0 and ca['group_group'] > 0 and ca['group_plane'] > 0
    _same_bags(_five_bags(a, ca), _five_bags(b, cb), 'classic rings on a blob')
    # 3. a partial selection and other parameters over lists that came with the upload (they hold every candidate, the pass filters)
    b.set_blob(_capi.pack_blob(rich)); a.set_complex(rich)
    sel = (rich.res_id % 3 == 0).astype(np.uint8)
    a.set_selection(sel); b.set_selection(sel)
    for cutoff, comp in ((5.0, 0.1), (4.0, 0.3), (6.5, 0.0)):
        ca, cb = a.run_launch(cutoff, comp, False, 6.0), b.run_launch(cutoff, comp, False, 6.0)
        assert ca == cb, (cutoff, comp)
        _same_bags(_five_bags(a, ca), _five_bags(b, cb), ('selection', cutoff))
    # 4. a batch declared over an uploaded concatenation: the grids of the upload are thrown away, the structures keep apart
    pcs = [synth.make_synthetic(1500, seed=30 + k, box=(40.0, 40.0, 40.0), n_rings=200, n_amides=200, id='b%d' % k) for k in range(3)]
    big, off = batch.concat_complexes(pcs)
    b.set_blob(_capi.pack_blob(big)); b.declare_batch(off)
    a.set_batch(pcs)
    ca, cb = a.run_launch(5.0, 0.1, False, 6.0), b.run_launch(5.0, 0.1, False, 6.0)
    assert ca == cb and ca['plane_plane'] > 0
    _same_bags(_five_bags(a, ca), _five_bags(b, cb), 'batch')
    a.close(); b.close()


@pytest.mark.gpu
def test_weighted_runs_of_atoms_change_no_result():
    """Sparse and clumped grids give the blocks of k_search runs of ATOMS; from the third pass over a grid on, runs of equal WEIGHT
    (k_cell_weights / k_balance_atoms, worked out by the second pass).  Any partition of the atoms is a correct one: every pass of
    a sequence over one structure gives the same five bags, for a chain folded onto itself (clumps of dozens of atoms per cell), a
    batch of proteins in one grid and a protein in its bounding box; a change of cutoff or structure in between starts over."""
    from arpeggio_amd import _capi, synth
    ctx = _capi.Context(0)
    ctx.set_grid_reuse(False)
    chain = synth.proteinlike(n_res=1200, n_waters=600, seed=31)
    prot = synth.proteinlike(n_res=150, n_waters=60, seed=32)
    for what, setup in (('chain', lambda: ctx.set_complex(chain)), ('protein', lambda: ctx.set_complex(prot)),
                        ('batch', lambda: ctx.set_batch([synth.proteinlike(n_res=60 + 7 * k, n_waters=20, seed=40 + k) for k in range(6)]))):
        setup()
        ref = None
        for k, cutoff in enumerate((5.0, 5.0, 5.0, 5.0, 4.0, 4.0, 4.0, 5.0)):
            cnt = ctx.run_launch(cutoff, 0.1, False, 6.0)
            bags = _five_bags(ctx, cnt)
            if cutoff == 5.0:
                if ref is None:
                    ref = bags
                else:
                    _same_bags(ref, bags, (what, k))
        assert len(ref['atom_atom']['i']) > 1000
    ctx.close()


def test_enqueue_and_wait_keep_several_contexts_busy_from_one_thread():
    """arp_run_enqueue / arp_run_wait: the pass of run_launch in two calls; three contexts driven round-robin by one thread give
    what three run_launch calls give, an enqueue without its wait (or a wait without an enqueue) is refused."""
    from arpeggio_amd import _capi, synth
    pcs = [synth.config3(6000 + 3000 * k, seed=5 + k) for k in range(3)]
    ctxs = [_capi.Context(0) for _ in pcs]
    want = []
    for c, pc in zip(ctxs, pcs):
        c.set_complex(pc)
        n = c.run_launch(5.0, 0.1, False, 6.0)
        want.append((n, c.atom_contacts_fetch(n['atom_atom'])))
    for rep in range(4):
        for c in ctxs:
            c.run_enqueue(5.0, 0.1, False, 6.0)
        for c, (n, contacts) in zip(ctxs, want):
            got = c.run_wait()
            assert got == n
            if rep == 3:
                g = c.atom_contacts_fetch(got['atom_atom'])
                assert all(np.array_equal(g[k], contacts[k]) for k in contacts)
    ctxs[0].run_enqueue()
    with pytest.raises(Exception, match='waited'):
        ctxs[0].run_enqueue()
    assert ctxs[0].run_wait() == want[0][0]
    with pytest.raises(Exception, match='enqueued'):
        ctxs[0].run_wait()
    # a pass whose buffers are too small is repeated inside the wait: a fresh context, first pass through enqueue / wait
    fresh = _capi.Context(0)
    fresh.set_complex(pcs[2])
    fresh.run_enqueue()
    assert fresh.run_wait() == want[2][0]
    for c in ctxs + [fresh]:
        c.close()


def test_covalent_flag_for_every_bond_count(ctx):
    """The covalent test reads the first four bonded neighbours of bgn from a quad beside the record and walks the CSR list only
    for the fifth and later ones: centres with 0 .. 9 partners, as the lower and as the higher index of the pair, partners
    listed before and after unbonded atoms at the same distance."""
    from helpers import tiny_complex
    from arpeggio_amd.core import config
    rng = np.random.default_rng(21)
    xyz, bonds = [], []
    for k in range(10):                       # centre with k bonded and 3 unbonded atoms 1.5 A away, clusters 30 A apart
        for centre_first in (True, False):
            base = len(xyz)
            c = np.array([30.0 * k, 40.0 * centre_first, 0.0])
            v = rng.standard_normal((k + 3, 3))
            v = 1.5 * v / np.linalg.norm(v, axis=1, keepdims=True)
            pts = [c] + list(c + v) if centre_first else list(c + v) + [c]
            xyz += pts
            centre = base if centre_first else base + k + 3
            others = [base + 1 + t for t in range(k + 3)] if centre_first else [base + t for t in range(k + 3)]
            order = rng.permutation(k + 3)
            bonds += [(centre, others[t]) for t in order[:k]]
    xyz = np.array(xyz, np.float32)
    pc = tiny_complex(xyz, res_id=np.arange(len(xyz), dtype=np.int32), bonds=bonds)
    got = _check(ctx, pc)
    cov = 1 << config.SIFT_NAMES.index('covalent')
    pairs = {(int(a), int(b)) for a, b, s in zip(got['i'], got['j'], got['sift']) if s & cov}
    assert pairs == {(min(a, b), max(a, b)) for a, b in bonds}


def test_blob_round_trip_with_and_without_the_radius_column(capi):
    """arp_set_blob leaves the per-atom radii on the host when every atom's pair is in the blob's table (structures of 32 768 atoms
    and more: below that a second copy costs more than the bytes) and writes them on the device from the table: what arp_get_blob returns is the uploaded blob, byte for byte, either
    way — also for a structure with more than 256 distinct radius pairs, whose radii do travel."""
    from arpeggio_amd import synth
    for n, many in ((3000, False), (40000, False), (40000, True)):
        pc = synth.config3(n, seed=17)
        if many:      # 400 distinct {vdw, cov} pairs: 144 of them cannot be in the table of 256
            pc.vdw = pc.vdw + (np.arange(pc.n_atoms) % 400) * 1e-3
        blob = capi.pack_blob(pc)
        hdr = capi.unpack_blob(blob)['header']
        assert (int(hdr.n_rad) == 256) == many
        c = capi.Context(0)
        try:
            c.set_blob(blob)
            back = c.get_blob()
            assert np.array_equal(np.asarray(back), np.asarray(blob)), (n, many)
            got = c.atom_contacts()
            c2 = capi.Context(0)
            try:
                c2.set_complex(pc)
                exp = c2.atom_contacts()
            finally:
                c2.close()
            for k in ('i', 'j', 'sift', 'ctype'):
                assert np.array_equal(got[k], exp[k]), (n, many, k)
        finally:
            c.close()


def test_list_counts_of_one_structure_never_reach_the_next(capi):
    """The entry counts of the ring / amide candidate lists travel with the counters of a pass; they are written with atomics like
    every other counter (a plain store used to stay in one XCD's L2 past the publication and surface in the NEXT structure's
    first pass as a list that was 'too small', now and then three times in a row: arp_run_launch -3)."""
    from arpeggio_amd import synth
    from helpers import random_dense_pack
    big = synth.config5(900, 700, L=60.0)
    rng = np.random.default_rng(3)
    c = capi.Context(0)
    try:
        for k in range(25):
            c.set_complex(big)
            n_big = c.run_launch()
            assert n_big['plane_plane'] > 1000
            small = random_dense_pack(500 + k, n=200, box=9.0)
            small.ring_center, small.ring_normal = rng.random((3, 3)) * 9.0, rng.standard_normal((3, 3))
            small.ring_res = np.zeros(3, np.int32)
            small.ring_atoms = []
            c.set_complex(small)
            n_small = c.run_launch()                  # (must not need three attempts)
            assert n_small['plane_plane'] <= 3
    finally:
        c.close()


def test_bad_radii_are_rejected_whether_or_not_the_radius_column_travels(capi):
    """From 32 768 atoms on the per-atom radii are written on the device from the blob's table; a non-finite table entry in
    use, an index beyond the table, or (radii travelling: an atom outside the table) a non-finite per-atom radius fail the
    validation either way, and the context takes a good blob afterwards."""
    from arpeggio_amd import synth
    pc = synth.config3(40_000, seed=23)
    c = capi.Context(0)
    try:
        def broken(mutate):
            bad = capi.pack_blob(pc)
            h = capi.BlobHeader.from_buffer(bad)
            mutate(bad, h)
            del h
            return bad
        def tab_nan(bad, h): np.frombuffer(bad, np.float64, 2, int(h.off[20]))[0] = np.nan          # entry 0 is in use
        def idx_range(bad, h): np.frombuffer(bad, np.uint16, 1, int(h.off[19]))[0] = int(h.n_rad)
        def outside_table_nan(bad, h):                                                                # one atom outside the table: the column travels
            np.frombuffer(bad, np.uint16, 8, int(h.off[19]))[5] = 0xFFFF
            np.frombuffer(bad, np.float64, 16, int(h.off[1]))[10] = np.inf
        for m in (tab_nan, idx_range, outside_table_nan):
            with pytest.raises(ValueError):
                c.set_blob(broken(m))
        def outside_table_ok(bad, h): np.frombuffer(bad, np.uint16, 8, int(h.off[19]))[5] = 0xFFFF    # (its radii are in the column)
        c.set_blob(broken(outside_table_ok))
        n1 = c.run_launch()
        c.set_blob(capi.pack_blob(pc))
        assert c.run_launch() == n1
    finally:
        c.close()


@pytest.mark.gpu
def test_each_ring_amide_loop_alone_from_its_list_and_by_its_grid_walk():
    """arp_*_launch alone: ARP_BAG_LISTS chooses, loop by loop, between the evaluation from the static candidate list (as a whole
    pass does it) and the grid walk; both must deliver the same bags (two processes: the switch is read once)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('import sys, json, hashlib, numpy as np; sys.path.insert(0, %r)\n'
            'from arpeggio_amd import synth, _capi\n'
            'out = {}\n'
            'for pc in (synth.config5(1500, 1500, seed=7, L=45.0), synth.proteinlike(n_res=120, n_waters=40, seed=9)):\n'
            '    c = _capi.Context(0); c.set_complex(pc)\n'
            '    for name in ("plane_plane", "atom_plane", "group_group", "group_plane"):\n'
            '        n = c.launch_bag(name); b = c.fetch_bag(name, sort=True)\n'
            '        h = hashlib.sha256()\n'
            '        for k in sorted(b): h.update(np.ascontiguousarray(b[k]).tobytes())\n'
            '        out[pc.id + ":" + name] = [int(n), h.hexdigest()]\n'
            '    c.close()\n'
            'print(json.dumps(out))\n') % root
    res = {}
    for mode in ('0', '15'):
        r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, ARP_BAG_LISTS=mode), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    assert res['0'] == res['15'] and sum(v[0] for v in res['0'].values()) > 1000, res


@pytest.mark.gpu
def test_sift_blocks_dealt_by_index_give_the_same_contacts():
    """ARP_SIFT_SEG_BY_BLOCK=1: the per-pair kernel takes a block's pair-list segment from its index instead of from the XCD it runs
    on — what arp_create switches to on a device whose dispatcher is not round-robin over eight XCDs.  Same records, same counters."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('import sys, json, hashlib, numpy as np; sys.path.insert(0, %r)\n'
            'from arpeggio_amd import synth, _capi\n'
            'out = {}\n'
            'for pc in (synth.config3(40_000, seed=11), synth.proteinlike(n_res=150, n_waters=60, seed=12)):\n'
            '    c = _capi.Context(0); c.set_complex(pc)\n'
            '    for k in range(3):\n'
            '        cnt = c.run_launch(5.0, 0.1, False, 6.0)\n'
            '    bags, _ = c.fetch_packed()\n'
            '    h = hashlib.sha256()\n'
            '    for name in sorted(bags):\n'
            '        for k in sorted(bags[name]): h.update(np.ascontiguousarray(bags[name][k]).tobytes())\n'
            '    out[pc.id] = [dict(cnt), h.hexdigest(), c.stats()["candidates"]]\n'
            '    c.close()\n'
            'print(json.dumps(out))\n') % root
    res = {}
    for mode in ('0', '1'):
        r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, ARP_SIFT_SEG_BY_BLOCK=mode), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    assert res['0'] == res['1'] and all(v[0]['atom_atom'] > 1000 for v in res['0'].values()), res
