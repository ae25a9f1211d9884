"""Known-answer tests of the oracle for the quirks of the reference's per-pair loop
(SURVEY.md §8a, Q1-Q16), derived by hand from the cited lines.  CPU only."""
import numpy as np
import pytest

import oracle
from arpeggio_amd.core import config
from helpers import tiny_complex

T = config.ATOM_TYPE_BIT
S = {n: 1 << i for i, n in enumerate(config.SIFT_NAMES)}


def pair(pc, b=0, e=1, comp=0.1, seq_adj=False, sel=None, plus=None):
    oc = oracle.OracleComplex(pc, in_sel=sel, in_plus=plus)
    return oc.pair_contact(b, e, comp, seq_adj)


def two_atoms(d, **kw):
    return tiny_complex([[0, 0, 0], [d, 0, 0]], **kw)


@pytest.mark.parametrize('d,flag', [(1.0, 'clash'), (1.52, 'vdw_clash'), (3.39, 'vdw_clash'), (3.4, 'vdw'),
                                    (3.5, 'vdw'), (3.51, 'proximal'), (4.9, 'proximal')])
def test_exclusive_ladder_and_boundaries(d, flag):
    """I:759-773: strict < for clash / vdw_clash, <= for vdw (+comp), float32 compares (Q5).
    vdw 1.7+1.7 = 3.4, cov 0.76+0.76 = 1.52, comp 0.1 -> 3.5."""
    ok, dist, sift, ct, err = pair(two_atoms(d))
    assert ok and dist == np.float32(d)
    assert sift & 0x1F == S[flag]


def test_covalent_beats_clash_and_still_gets_feature_flags():
    """I:756 covalent first; I:786 gate ignores covalent (Q7)."""
    pc = two_atoms(1.0, bonds=[(0, 1)], type_mask=T['hydrophobe'])
    ok, _, sift, _, _ = pair(pc)
    assert sift & 0x1F == S['covalent'] and sift & S['hydrophobic']
    # the same pair unbonded is a clash and gets no feature flags
    ok, _, sift, _, _ = pair(two_atoms(1.0, type_mask=T['hydrophobe']))
    assert sift == S['clash']


def test_metal_complex_evaluated_outside_the_gate():
    """I:777-783 (Q6): metal flag even for a clash pair."""
    pc = tiny_complex([[0, 0, 0], [1.0, 0, 0]], type_mask=[T['hbond acceptor'], 0], flags=[0, config.F_METAL])
    ok, _, sift, _, _ = pair(pc)
    assert sift == S['clash'] | S['metal_complex']
    pc = tiny_complex([[0, 0, 0], [2.81, 0, 0]], type_mask=[T['hbond acceptor'], 0], flags=[0, config.F_METAL])
    assert not pair(pc)[2] & S['metal_complex']


def test_hydrogen_and_same_residue_and_sequence_filters():
    assert not pair(two_atoms(3.0, flags=[config.F_HYDROGEN, 0]))[0]          # I:712
    assert not pair(two_atoms(3.0, res_id=[0, 0]))[0]                          # I:729
    P = config.R_POLYPEPTIDE | config.R_HAS_SEQ
    adj = dict(res_id=[0, 1], res_flags=[P, P], res_prev=[-1, 0], res_next=[1, -1])
    assert not pair(two_atoms(3.0, **adj))[0]                                  # I:733-741
    assert pair(two_atoms(3.0, **adj), seq_adj=True)[0]
    # Q4: only res_end.is_polypeptide is tested: bgn polypeptide / end not -> kept
    adj2 = dict(res_id=[0, 1], res_flags=[P, config.R_HAS_SEQ], res_prev=[-1, 0], res_next=[1, -1])
    assert pair(two_atoms(3.0, **adj2))[0]
    assert not pair(two_atoms(3.0, **adj2), b=1, e=0)[0]                       # orientation matters


def test_water_shortcut_needs_vdw_distance():
    """I:791-799 (Q8)."""
    W = config.F_WATER
    da = T['hbond acceptor'] | T['hbond donor']
    pc = tiny_complex([[0, 0, 0], [3.4, 0, 0]], type_mask=[da, T['hbond acceptor']], flags=[W, 0])
    assert pair(pc)[2] & (S['hbond'] | S['polar']) == S['hbond'] | S['polar']
    # beyond vdw_sum + comp the water shortcut does not fire and the donor branch needs hydrogens
    pc = tiny_complex([[0, 0, 0], [3.6, 0, 0]], type_mask=[da, T['hbond acceptor']], flags=[W, 0])
    s = pair(pc)[2]
    assert not s & S['hbond'] and not s & S['polar']          # 3.6 > polar distance 3.5


def test_hbond_angle_and_orientation_dependence():
    """U:73-93, I:804-819 (Q9): if/elif — when both atoms are donor+acceptor only bgn-as-donor is tested."""
    da = T['hbond acceptor'] | T['hbond donor']
    # bgn has a hydrogen pointing at end (angle 180 deg); end has none
    pc = tiny_complex([[0, 0, 0], [2.9, 0, 0]], type_mask=[da, da], h={0: [[1.0, 0, 0]]})
    assert pair(pc, 0, 1)[2] & S['hbond']
    assert not pair(pc, 1, 0)[2] & S['hbond']      # reversed orientation: end-as-donor never tried
    assert pair(pc, 1, 0)[2] & S['polar']          # polar is distance only
    # hydrogen at 80 degrees: fails angle >= 1.57
    pc = tiny_complex([[0, 0, 0], [2.9, 0, 0]], type_mask=[T['hbond donor'], T['hbond acceptor']],
                      h={0: [[np.cos(np.deg2rad(80)), np.sin(np.deg2rad(80)), 0]]})
    assert not pair(pc)[2] & S['hbond']
    pc = tiny_complex([[0, 0, 0], [2.9, 0, 0]], type_mask=[T['hbond donor'], T['hbond acceptor']],
                      h={0: [[np.cos(np.deg2rad(60)), np.sin(np.deg2rad(60)), 0]]})
    # angle(donor, H, acceptor) is measured AT the hydrogen: here ~ 100 deg >= 1.57 rad, |H-A| = 2.55 <= 3.0
    assert pair(pc)[2] & S['hbond']
    # same angle but the hydrogen too far from the acceptor (|H-A| = 3.15 > 1.2 + 1.7 + 0.1)
    pc = tiny_complex([[0, 0, 0], [2.9, 0, 0]], type_mask=[T['hbond donor'], T['hbond acceptor']],
                      h={0: [[np.cos(np.deg2rad(95)), np.sin(np.deg2rad(95)), 0]]})
    assert not pair(pc)[2] & S['hbond']


def test_weak_hbond_overwrite_order():
    """I:857-886 (Q10): the later applicable branch overwrites SIFt[6]."""
    acc_wd = T['hbond acceptor'] | T['weak hbond donor']
    # branch (1) end donates to bgn: succeeds; branch (2) bgn donates to end: no hydrogens -> overwrites with 0
    pc = tiny_complex([[0, 0, 0], [3.0, 0, 0]], type_mask=[acc_wd, acc_wd], h={1: [[2.0, 0, 0]]})
    s = pair(pc)[2]
    assert not s & S['weak_hbond'] and s & S['weak_polar']
    # only branch (1) applicable -> flag stays
    pc = tiny_complex([[0, 0, 0], [3.0, 0, 0]], type_mask=[T['hbond acceptor'], T['weak hbond donor']], h={1: [[2.0, 0, 0]]})
    assert pair(pc)[2] & S['weak_hbond']


def test_xbond_gate_angle_and_none_neighbour():
    """I:889-895, U:158-179 (Q11)."""
    xd, xa = T['xbond donor'], T['xbond acceptor']
    # C-X...A linear: theta = 180 deg >= 2.09 rad
    pc = tiny_complex([[-1.7, 0, 0], [0, 0, 0], [3.3, 0, 0]], type_mask=[0, xd, xa], bonds=[(0, 1)], res_id=[0, 0, 1])
    assert pair(pc, 1, 2)[2] & S['xbond']
    assert pair(pc, 2, 1)[2] & S['xbond']       # elif branch, mirrored
    # bent to 100 deg: fails
    pc = tiny_complex([[1.7 * np.cos(np.deg2rad(100)), 1.7 * np.sin(np.deg2rad(100)), 0], [0, 0, 0], [3.3, 0, 0]],
                      type_mask=[0, xd, xa], bonds=[(0, 1)], res_id=[0, 0, 1])
    assert not pair(pc, 1, 2)[2] & S['xbond']
    # too far for the vdw gate (3.4 + 0.1)
    pc = tiny_complex([[-1.7, 0, 0], [0, 0, 0], [3.6, 0, 0]], type_mask=[0, xd, xa], bonds=[(0, 1)], res_id=[0, 0, 1])
    assert not pair(pc, 1, 2)[2] & S['xbond']
    # donor without a single-bond neighbour: the reference dereferences None
    pc = tiny_complex([[0, 0, 0], [3.3, 0, 0]], type_mask=[xd, xa])
    assert pair(pc)[4] == -4


def test_ionic_carbonyl_aromatic_hydrophobic_thresholds():
    for d, on in ((4.0, True), (4.01, False)):
        pc = two_atoms(d, type_mask=[T['pos ionisable'], T['neg ionisable']])
        assert bool(pair(pc)[2] & S['ionic']) is on
    for d, on in ((3.6, True), (3.61, False)):
        pc = two_atoms(d, type_mask=[T['carbonyl oxygen'], T['carbonyl carbon']])
        assert bool(pair(pc)[2] & S['carbonyl']) is on
        assert bool(pair(pc, 1, 0)[2] & S['carbonyl']) is on
    for d, on in ((4.0, True), (4.01, False)):
        assert bool(pair(two_atoms(d, type_mask=T['aromatic']))[2] & S['aromatic']) is on
    for d, on in ((4.5, True), (4.51, False)):
        assert bool(pair(two_atoms(d, type_mask=T['hydrophobe']))[2] & S['hydrophobic']) is on


def test_deuterium_is_not_filtered():
    """I:712 filters element 'H' only (Q2): a 'D' atom is packed without F_HYDROGEN and stays."""
    assert pair(two_atoms(3.0, flags=[0, 0]))[0]


def test_search_all_grid_equals_brute_force_random_and_adversarial():
    rng = np.random.default_rng(5)
    for n, L in ((500, 20.0), (4000, 40.0), (3000, 200.0)):
        xyz = (rng.random((n, 3)) * L).astype(np.float32)
        for r in (5.0, 6.0, 1.0):
            gi, gj, cand = oracle.search_all(xyz, r, grid=True)
            bi, bj, _ = oracle.search_all(xyz, r, grid=False)
            assert np.array_equal(gi, bi) and np.array_equal(gj, bj)
            assert cand >= len(gi)
    g = np.arange(0, 26, 5, dtype=np.float32)
    lattice = np.array([[x, y, z] for x in g for y in g for z in g[:2]], np.float32)   # spacing == radius
    pts = np.concatenate([lattice, lattice[:7], [[3, 4, 0]]]).astype(np.float32)      # duplicates, 3-4-5 triangle
    gi, gj, _ = oracle.search_all(pts, 5.0, grid=True)
    bi, bj, _ = oracle.search_all(pts, 5.0, grid=False)
    assert np.array_equal(gi, bi) and np.array_equal(gj, bj)
    # inclusive boundary: the pair at exactly 5.0 is present
    k = len(lattice) + 7
    assert ((bi == 0) & (bj == k)).any()
    act = (np.arange(len(pts)) % 2).astype(np.uint8)
    gi, gj, _ = oracle.search_all(pts, 5.0, active=act, grid=True)
    bi, bj, _ = oracle.search_all(pts, 5.0, active=act, grid=False)
    assert np.array_equal(gi, bi) and np.array_equal(gj, bj) and np.all(act[gi] == 1)


def test_make_selection_expands_by_six_angstrom_including_hydrogens():
    """I:1420-1424: both atoms of every pair <= 6.0 with >= 1 selected atom join selection_plus."""
    pc = tiny_complex([[0, 0, 0], [6.0, 0, 0], [12.0, 0, 0], [12.1, 0, 0], [30, 0, 0]],
                      flags=[0, config.F_HYDROGEN, 0, 0, 0])
    oc = oracle.OracleComplex(pc)
    plus = oc.make_selection(np.array([1, 0, 0, 0, 0], np.uint8), use_grid=False)
    assert plus.tolist() == [1, 1, 0, 0, 0]          # the hydrogen at exactly 6.0 joins; no transitive growth
    plus2 = oc.make_selection(np.array([1, 0, 0, 0, 0], np.uint8), use_grid=True)
    assert plus2.tolist() == plus.tolist()


def test_plane_plane_classes_at_bin_edges():
    L = oracle.lib()
    names = config.PLANE_PLANE_NAMES
    assert names[L.orc_pp_class(30.0, 30.0)] == 'FF'
    assert names[L.orc_pp_class(30.0, 30.0001)] == 'OF'
    assert names[L.orc_pp_class(30.0, 60.0001)] == 'EE'
    assert names[L.orc_pp_class(30.0001, 30.0)] == 'FT'
    assert names[L.orc_pp_class(60.0, 90.0)] == 'ET'
    assert names[L.orc_pp_class(60.0001, 30.0)] == 'FE'
    assert names[L.orc_pp_class(90.0, 60.0)] == 'OE'
    assert names[L.orc_pp_class(90.0, 90.0)] == 'EF'
    assert names[L.orc_pp_class(float('nan'), 10.0)] == ''
    assert names[L.orc_pp_class(10.0, float('nan'))] == ''


def test_initialize_geometry_restatement_known_answers():
    """oracle/ref_py.py ring / amide geometry and ring -> residue assignment on hand-made inputs (I:1697-1733,
    1531-1589, 1453-1492)."""
    import numpy as np
    from oracle import ref_py
    ang = np.arange(6) * np.pi / 3
    hexagon = np.stack([1.39 * np.cos(ang) + 4.0, 1.39 * np.sin(ang) - 2.0, np.full(6, 7.5)], axis=1).astype(np.float32)
    extra = np.array([[4.0, -2.0, 8.6], [4.0, -2.0, 10.4], [9.0, 9.0, 9.0]], np.float32)   # 1.1 A and 2.9 A above the centre, one far away
    xyz = np.concatenate([hexagon, extra])
    ctr, nrm = ref_py.ring_geometry(xyz, [list(range(6))])
    assert np.allclose(ctr[0], [4.0, -2.0, 7.5], atol=1e-6)
    assert np.allclose(nrm[0], [0, 0, 1], atol=1e-6)                         # counter-clockwise ring: +z
    ctr2, nrm2 = ref_py.ring_geometry(xyz, [list(range(5, -1, -1))])
    assert np.allclose(nrm2[0], [0, 0, -1], atol=1e-6)                       # same atoms the other way round: -z
    res_id = np.array([3, 3, 3, 3, 3, 3, 8, 9, 1], np.int32)
    res, dist = ref_py.ring_residues(xyz, res_id, ctr)
    assert res.tolist() == [8] and abs(dist[0] - 1.1) < 1e-6                 # the atom 1.1 A above beats the ring atoms (1.39 A)
    res, dist = ref_py.ring_residues(xyz[:6], res_id[:6], ctr)
    assert res.tolist() == [3] and abs(dist[0] - 1.39) < 1e-6
    res, dist = ref_py.ring_residues(xyz[8:], res_id[8:], ctr)
    assert res.tolist() == [-1] and dist.tolist() == [-1.0]                  # nothing within 3 A (I:1476-1479)
    # amide in the plane z = 2: centre = midpoint of C and N, normal = +-z
    am = np.array([[1.3, 0.2, 2.0], [0.0, 0.0, 2.0], [-0.6, 1.05, 2.0], [-0.8, -1.2, 2.4]], np.float32)   # N, C, O, CA
    c, n = ref_py.amide_geometry(am, [[0, 1, 2, 3]])
    assert c.dtype == np.float32 and np.allclose(c[0], [0.65, 0.1, 2.0], atol=1e-6)
    assert np.allclose(np.abs(n[0]), [0, 0, 1], atol=1e-5)


def test_openmp_timing_variant_counts_what_the_oracle_counts():
    """bench.py's multi-core CPU figure runs the same work: candidates, contacts and the SIFt checksum of
    oracle.pass_openmp equal those of the single-thread restatement (whole-structure selection)."""
    import numpy as np
    import oracle
    from arpeggio_amd import synth
    pc = synth.config3(6000, seed=11)
    oc = oracle.OracleComplex(pc)
    oc.make_selection(None)
    r = oc.atom_contacts()
    for threads in (1, 3):
        got = oracle.pass_openmp(oc, threads=threads)
        assert got['candidates'] == int(r['stats'][0]) and got['contacts'] == len(r['i'])
        assert got['sift_checksum'] == int(r['sift'].astype(np.int64).sum())
