"""Row f2 (the geometric part of initialize()) against fixtures produced by EXECUTING the reference's own
_perceive_amide_groups (I:1531-1589) and _assign_aromatic_rings_to_residues (I:1453-1492) on data holders
(tests/golden/make_golden_prepare.py; inputs and outputs only are stored).

Bit-exact: amide centre (float32), amide residue, ring residue, ring - atom distance (float64).  Documented deviation: the
amide NORMAL.  The reference takes it from LAPACK's float32 SVD of three centred points; its direction carries that
routine's rounding noise (about 1e-7 per component).  The HIP path computes the exact plane normal (cross product in
float64, rounded to float32), which is within 1e-4 degrees — the north star's bound on angles; 5.2e-6 measured — of the
reference's wherever the three atoms are not collinear; the test reports the largest angle and asserts that no AMIDEAMIDE /
AMIDERING record of the fixture structures appears or disappears because of it (the 30-degree tests of I:1281, 1363)."""
import os

import numpy as np
import pytest

from arpeggio_amd.core.interactions import amide_majority_residue
from oracle import ref_py


@pytest.fixture(scope='module')
def prep(golden_dir):
    z = np.load(os.path.join(golden_dir, 'prepare_cases.npz'), allow_pickle=False)
    return z, [str(n) for n in z['names']]


def angle_deg(a, b):
    """Unsigned angle between directions a and b (sign of a plane normal is arbitrary), float64."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    c = np.abs((a * b).sum(axis=1)) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    s = np.linalg.norm(np.cross(a, b), axis=1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    return np.degrees(np.arctan2(s, c))


def conditioning(xyz, quads):
    """sin of the angle at C between C->O and C->N: near 0 = collinear atoms, plane undefined."""
    x = xyz.astype(np.float64)
    u, v = x[quads[:, 2]] - x[quads[:, 1]], x[quads[:, 0]] - x[quads[:, 1]]
    return np.linalg.norm(np.cross(u, v), axis=1) / (np.linalg.norm(u, axis=1) * np.linalg.norm(v, axis=1) + 1e-300)


def test_restatement_equals_executed_reference(prep):
    z, names = prep
    n_am = n_ring = 0
    for n in names:
        xyz, res_id, quads = z[n + '/xyz'], z[n + '/res_id'], z[n + '/amide_atoms']
        ctr, nrm = ref_py.amide_geometry(xyz, quads)
        assert np.array_equal(ctr.view(np.uint32), z[n + '/amide_center'].view(np.uint32)), n           # float32, bit for bit
        assert (angle_deg(nrm, z[n + '/amide_normal']) < 1e-3).all(), n                                 # same NumPy call; LAPACK build may differ
        assert np.array_equal(amide_majority_residue(res_id, quads), z[n + '/amide_res']), n
        res, dist = ref_py.ring_residues(xyz, res_id, z[n + '/ring_center'])
        assert np.array_equal(res, z[n + '/ring_res']), n
        assert np.array_equal(dist.view(np.uint64), z[n + '/ring_dist'].view(np.uint64)), n              # float64, bit for bit
        n_am += len(quads); n_ring += len(res)
    assert n_am > 600 and n_ring > 100
    assert (z['proteinlike_displaced/ring_res'] < 0).sum() >= 1          # the 'residue None' branch (I:1476-1479) occurs


def test_majority_residue_ties_take_the_first():
    res_id = np.array([5, 5, 6, 6, 7, 8, 9], np.int32)
    assert amide_majority_residue(res_id, [[0, 2, 3, 1]]).tolist() == [5]      # 5, 6, 6, 5 -> tie, first = 5
    assert amide_majority_residue(res_id, [[2, 0, 1, 3]]).tolist() == [6]      # 6, 5, 5, 6 -> tie, first = 6
    assert amide_majority_residue(res_id, [[4, 5, 6, 0]]).tolist() == [7]      # all different -> first
    assert amide_majority_residue(res_id, [[4, 2, 3, 0]]).tolist() == [6]


@pytest.mark.gpu
def test_hip_geometry_equals_executed_reference(prep):
    from arpeggio_amd import _capi, synth
    z, names = prep
    ctx = _capi.Context(0)
    worst, n_well = 0.0, 0
    for n in names:
        xyz, res_id, quads = z[n + '/xyz'], z[n + '/res_id'], z[n + '/amide_atoms']
        pc = synth.make_synthetic(len(xyz), seed=1, box=(10, 10, 10), n_rings=0, n_amides=0)
        pc.xyz = xyz.copy()
        pc.res_id = (res_id - res_id.min()).astype(np.int32)
        nres = int(pc.res_id.max()) + 1
        pc.res_flags, pc.res_prev, pc.res_next = np.zeros(nres, np.uint8), np.full(nres, -1, np.int32), np.full(nres, -1, np.int32)
        ctx.set_complex(pc)
        ctr, nrm = ctx.amide_geometry(quads)
        assert np.array_equal(ctr.view(np.uint32), z[n + '/amide_center'].view(np.uint32)), n
        ang = angle_deg(nrm, z[n + '/amide_normal'])
        well = conditioning(xyz, quads) > 0.05
        assert (ang[well] < 1e-4).all(), (n, float(ang[well].max()))      # the north-star bound on angles (measured: 5.2e-6 degrees)
        worst = max(worst, float(ang[well].max()) if well.any() else 0.0)
        n_well += int(well.sum())
        res, dist = ctx.ring_residues(z[n + '/ring_center'])
        exp = z[n + '/ring_res']
        assert np.array_equal(np.where(res >= 0, res + res_id.min(), res), exp), n
        assert np.array_equal(dist.view(np.uint64), z[n + '/ring_dist'].view(np.uint64)), n
    print(f'amide normals: {n_well} well-conditioned groups, largest angle to the reference SVD normal {worst:.2e} degrees')
    assert n_well > 500
    ctx.close()


@pytest.mark.gpu
def test_amide_normal_deviation_flips_no_record(prep):
    """run_arpeggio on the stand-in with the reference's SVD normals and with the HIP normals: the same AMIDEAMIDE /
    AMIDERING records (ids, distances, contact types); the two angle columns agree to 1e-3 degrees."""
    from arpeggio_amd import _capi, synth
    z, names = prep
    pc = synth.proteinlike()
    assert np.array_equal(pc.xyz, z['proteinlike/xyz']) and np.array_equal(pc.amide_atoms, z['proteinlike/amide_atoms'])
    ctx = _capi.Context(0)
    bags = []
    ctx.set_complex(pc)
    hip_ctr, hip_nrm = ctx.amide_geometry(pc.amide_atoms)
    for nrm in (z['proteinlike/amide_normal'], hip_nrm):
        pc.amide_center, pc.amide_normal = z['proteinlike/amide_center'].copy(), np.ascontiguousarray(nrm, np.float32)
        ctx.set_complex(pc)
        ctx.run_launch(5.0, 0.1, False, 6.0)
        bags.append((ctx.fetch_bag('group_group'), ctx.fetch_bag('group_plane')))
    flips = 0
    for ref_bag, hip_bag, k0, k1 in ((bags[0][0], bags[1][0], 'bgn', 'end'), (bags[0][1], bags[1][1], 'amide', 'ring')):
        a = set(zip(ref_bag[k0].tolist(), ref_bag[k1].tolist()))
        b = set(zip(hip_bag[k0].tolist(), hip_bag[k1].tolist()))
        flips += len(a ^ b)
        if a == b:
            assert np.array_equal(ref_bag['dist'], hip_bag['dist']) and np.array_equal(ref_bag['ctype'], hip_bag['ctype'])
            assert np.allclose(ref_bag['dihedral'], hip_bag['dihedral'], atol=1e-3) and np.allclose(ref_bag['theta'], hip_bag['theta'], atol=1e-3)
    n_rec = len(bags[0][0]['bgn']) + len(bags[0][1]['amide'])
    print(f'AMIDEAMIDE + AMIDERING records with the reference normals: {n_rec}; records that differ with the HIP normals: {flips}')
    assert n_rec > 20 and flips == 0
    ctx.close()
