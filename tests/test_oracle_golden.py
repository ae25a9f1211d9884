"""The oracle (oracle/ref_c.c) against golden vectors produced by executing the
reference's own function bodies (tests/golden/make_golden.py).  CPU only."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle
from arpeggio_amd.core import config
from helpers import planes_only_complex

RAD_TOL = 1e-4 * np.pi / 180.0   # north-star tolerance: 1e-4 degrees


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_norm_f32_bit_exact(golden_dir):
    g = np.load(os.path.join(golden_dir, 'norm_f32.npz'))
    L = oracle.lib()
    got = np.array([L.orc_dist_f32(_p(g['a'][k]), _p(g['c'][k])) for k in range(len(g['a']))], np.float32)
    assert np.array_equal(got.view(np.uint32), g['dist'].view(np.uint32))


def test_get_angle_three_dtype_paths(golden_dir):
    g = np.load(os.path.join(golden_dir, 'angles.npz'))
    L = oracle.lib()
    a, b, c, bh, ch = g['a'], g['b'], g['c'], g['bh'], g['ch']
    a64, c64 = a.astype(np.float64), c.astype(np.float64)
    K = len(a)
    got32 = np.array([L.orc_get_angle_f32(_p(a[k]), _p(b[k]), _p(c[k])) for k in range(K)])
    got64 = np.array([L.orc_get_angle_f64(_p(a64[k]), _p(bh[k]), _p(c64[k])) for k in range(K)])
    gotmx = np.array([L.orc_get_angle_mixed(_p(a[k]), _p(b[k]), _p(ch[k])) for k in range(K)])
    # float32 path: allow 1 float32 ulp of acos (NumPy's SIMD arccos vs libm)
    assert np.max(np.abs(got32 - g['ang_f32'])) <= 5e-7
    assert np.max(np.abs(got64 - g['ang_f64'])) <= RAD_TOL
    assert np.max(np.abs(gotmx - g['ang_mix'])) <= RAD_TOL
    # NaN -> pi substitution happened identically
    assert np.array_equal(got32 == np.pi, g['ang_f32'] == np.pi)
    assert np.array_equal(got64 == np.pi, g['ang_f64'] == np.pi)
    # the float64 paths differ from NumPy only by arccos' last bit (SIMD arccos vs libm acos)
    assert np.max(np.abs(got64 - g['ang_f64'])) <= 1e-15
    assert np.max(np.abs(gotmx - g['ang_mix'])) <= 1e-15
    assert np.mean(got64 == g['ang_f64']) > 0.8


def test_group_angles(golden_dir):
    g = np.load(os.path.join(golden_dir, 'group_angles.npz'))
    L = oracle.lib()
    K = len(g['n64'])
    n64, m64, p64, n32, m32, p32 = (g[k] for k in ('n64', 'm64', 'p64', 'n32', 'm32', 'p32'))
    ga64 = np.array([L.orc_group_angle_f64(_p(n64[k]), _p(p64[k])) for k in range(K)])
    ga32 = np.array([L.orc_group_angle_f32(_p(n32[k]), _p(p32[k])) for k in range(K)], np.float32)
    ga3264 = np.array([L.orc_group_angle_f32n_f64p(_p(n32[k]), _p(p64[k])) for k in range(K)])
    for got, exp, tol in ((ga64, g['ga_64'], 1e-4), (ga32, g["ga_32"], 1e-4), (ga3264, g['ga_3264'], 1e-4)):
        assert np.array_equal(np.isnan(got), np.isnan(exp))
        ok = ~np.isnan(exp)
        assert np.max(np.abs(got[ok].astype(np.float64) - exp[ok])) <= tol


def test_is_hbond_and_weak(golden_dir):
    g = np.load(os.path.join(golden_dir, 'hbond.npz'))
    L = oracle.lib()
    don, acc, hoff, hxyz = g['don'], g['acc'], g['hoff'], g['hxyz']
    comp = float(g['comp'])
    for name, thr in (('is_hbond', 1.57), ('is_weak_hbond', 2.27)):
        got = np.array([L.orc_is_hbond_like(_p(don[k]), _p(hxyz[hoff[k]:]) if hoff[k] < len(hxyz) else None,
                                            int(hoff[k + 1] - hoff[k]), _p(acc[k]), float(g['acc_vdw'][k]), comp, thr)
                        for k in range(len(don))], np.int8)
        assert np.array_equal(got, g[name]), name
        assert 0 < got.sum() < len(got)


def test_contact_type_lut(golden_dir):
    rows = json.load(open(os.path.join(golden_dir, 'contact_type.json')))
    assert len(rows) == 16
    L = oracle.lib()
    for r in rows:
        code = L.orc_contact_type(r['bgn_sel'], r['end_sel'], r['bgn_water'], r['end_water'])
        assert config.CONTACT_TYPE_NAMES[code] == r['contact_type'], r


def test_sift_accumulators(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, 'sift_updates.json')))
    L = oracle.lib()
    code = {n: i for i, n in enumerate(config.CONTACT_TYPE_NAMES)}
    for case in cases:
        state = (C.c_uint16 * 4)()
        isift = ((C.c_uint8 * 15) * 4)()
        for st in case['steps']:
            add = sum(b << k for k, b in enumerate(st['addition']))
            L.orc_update_atom_sift(state, isift, C.c_uint16(add), code[st['contact_type']])
        fin = case['final']
        for slot, nm in enumerate(('sift', 'sift_inter_only', 'sift_intra_only', 'sift_water_only')):
            assert [(state[slot] >> k) & 1 for k in range(15)] == fin[nm], nm
            assert [(state[slot] >> (k + 5)) & 1 for k in range(10)] == fin['actual_f' + nm], nm
        for slot, nm in enumerate(('integer_sift', 'integer_sift_inter_only', 'integer_sift_intra_only', 'integer_sift_water_only')):
            assert list(isift[slot]) == fin[nm], nm


@pytest.fixture(scope='module')
def planes(golden_dir):
    g = np.load(os.path.join(golden_dir, 'planes_input.npz'))
    exp = json.load(open(os.path.join(golden_dir, 'planes_expected.json')))
    pc = planes_only_complex(g['ring_center'], g['ring_normal'], g['ring_res'], g['amide_center'], g['amide_normal'],
                             g['amide_res'], g['nres'])
    oc = oracle.OracleComplex(pc)
    oc.ring_sel[:] = g['ring_sel']; oc.ring_plus[:] = g['ring_plus']
    oc.amide_sel[:] = g['amide_sel']; oc.amide_plus[:] = g['amide_plus']
    return oc, exp


def test_plane_plane_loop_matches_reference(planes):
    oc, exp = planes
    got = oc.plane_plane()
    e = exp['plane_plane']
    assert len(got['bgn']) == len(e) > 500
    for k, rec in enumerate(e):   # same creation order as the reference
        assert (got['bgn'][k], got['end'][k]) == (rec['bgn_id'], rec['end_id'])
        assert got['dist'][k] == rec['distance']            # float64, bit-exact
        types = [config.PLANE_PLANE_NAMES[got['type1'][k]]]
        if got['type2'][k] < config.PP_SAME:
            types.append(config.PLANE_PLANE_NAMES[got['type2'][k]])
        assert types == rec['contact_type'], (k, rec)
        assert config.CONTACT_TYPE_NAMES[got['ctype'][k]] == rec['text']


def test_group_group_loop_matches_reference(planes):
    oc, exp = planes
    got = oc.group_group()
    e = exp['group_group']
    assert len(got['bgn']) == len(e) > 10
    for k, rec in enumerate(e):
        assert (got['bgn'][k], got['end'][k]) == (rec['bgn_id'], rec['end_id'])
        assert rec['distance_dtype'] == 'float32'
        assert got['dist'][k] == np.float32(rec['distance'])   # float32, bit-exact
        assert config.CONTACT_TYPE_NAMES[got['ctype'][k]] == rec['text']


def test_group_plane_loop_matches_reference(planes):
    oc, exp = planes
    got = oc.group_plane()
    e = exp['group_plane']
    assert len(got['amide']) == len(e) > 10
    for k, rec in enumerate(e):
        assert (got['amide'][k], got['ring'][k]) == (rec['bgn_id'], rec['end_id'])
        assert got['dist'][k] == rec['distance']
        assert config.CONTACT_TYPE_NAMES[got['ctype'][k]] == rec['text']


def test_accumulator_restatement_matches_reference_updates(golden_dir):
    """orc_atom_accumulators (contact-list form) == the reference's update_atom_sift sequence (sift_updates.json)."""
    cases = json.load(open(os.path.join(golden_dir, 'sift_updates.json')))
    code = {n: i for i, n in enumerate(config.CONTACT_TYPE_NAMES)}
    for case in cases:
        steps = case['steps']
        contacts = dict(i=np.zeros(len(steps), np.int32), j=np.ones(len(steps), np.int32),
                        sift=np.array([sum(b << k for k, b in enumerate(st['addition'])) for st in steps], np.uint16),
                        ctype=np.array([code[st['contact_type']] for st in steps], np.uint8))
        acc = oracle.atom_accumulators(2, contacts)
        fin = case['final']
        for slot, nm in enumerate(('sift', 'sift_inter_only', 'sift_intra_only', 'sift_water_only')):
            assert [(int(acc['sift'][0, slot]) >> k) & 1 for k in range(15)] == fin[nm]
            assert [(int(acc['sift'][1, slot]) >> (k + 5)) & 1 for k in range(10)] == fin['actual_f' + nm]


def test_residue_ring_sifts_match_reference(planes, golden_dir):
    """Host computation of the per-residue ring-ring integer SIFt (I:1171-1176) from the plane-plane bag
    == the counters the reference's own loop left on the residues (planes_expected.json)."""
    from arpeggio_amd.core.interactions import residue_plane_sifts
    oc, exp = planes
    pp = oc.plane_plane()
    got = residue_plane_sifts(oc.pc, {'plane_plane': pp, 'group_group': oc.group_group(), 'group_plane': oc.group_plane()})
    want = np.zeros_like(got['ring_ring_inter_integer_sift'])
    for r, v in exp['ring_ring_inter_integer_sift'].items():
        want[int(r)] = v
    assert want.sum() > 100
    assert np.array_equal(got['ring_ring_inter_integer_sift'], want)
    for name in ('amide_amide_inter_integer_sift', 'amide_ring_inter_integer_sift', 'ring_amide_inter_integer_sift'):
        w = np.zeros_like(got[name])
        for r, v in exp[name].items():
            w[int(r)] = v
        assert w.sum() > 0 and np.array_equal(got[name], w), name
