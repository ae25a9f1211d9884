#!/usr/bin/env python3
"""Golden vectors for the geometric part of initialize(), produced by EXECUTING the reference's own
``InteractionComplex._perceive_amide_groups`` (interactions.py:1531-1589: float32 centroid, np.linalg.svd plane) and
``_assign_aromatic_rings_to_residues`` (interactions.py:1453-1492) on data holders (build container only; needs
/root/reference).

What is replaced: OpenBabel's SMARTS matcher by a holder whose ``GetMapList`` returns the stored (N, C, O, C-alpha)
quadruples, Bio.PDB.NeighborSearch by the brute-force holder of make_golden_core.py (inclusive float64 test, atoms
delivered in index order).  What is executed for real: everything arpeggio itself computes — the majority residue of a
match, both centroids, the SVD normal (LAPACK through this container's NumPy), the closest atom to a ring centre, its
distance and residue.

    python tests/golden/make_golden_prepare.py     ->  tests/golden/prepare_cases.npz
"""
import collections
import logging
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get('ARP_GOLDEN_OUT', HERE)      # (tests/test_fixture_freshness.py regenerates into a scratch directory)
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import make_golden_core as core   # noqa: E402

from arpeggio_amd import synth   # noqa: E402  (inputs only: deterministic structures)

REF = core.REF


def run(pc, amide_atoms, ring_center, name, out):
    logging.disable(logging.CRITICAL)
    config = core.load_by_path('ref_config_prepare', os.path.join(REF, 'config.py'))
    pc.ensure_labels()
    chain = core.Chain('A')
    residues = [core.Residue(r, chain, pc.res_name[r], ' ', int(pc.res_seq[r]), ' ') for r in range(pc.n_residues)]
    atoms = [core.Atom(i, residues[int(pc.res_id[i])], pc.atom_name[i], pc.element[i], pc.xyz[i].copy(), i + 1, False) for i in range(pc.n_atoms)]
    ob_atoms = {i + 1: types.SimpleNamespace(GetId=(lambda i=i: 7000 + i)) for i in range(pc.n_atoms)}     # GetAtom is 1-based

    class SmartsPattern:
        def Init(self, smarts):
            assert smarts == config.AMIDE_SMARTS

        def Match(self, mol):
            return True

        def GetMapList(self):
            return [tuple(int(a) + 1 for a in q) for q in amide_atoms]

    ns = {'np': np, 'logging': logging, 'config': config, 'collections': collections,
          'ob': types.SimpleNamespace(OBSmartsPattern=SmartsPattern), 'NeighborSearch': core.NeighborSearch}
    IC = core.compile_class(os.path.join(REF, 'interactions.py'), 'InteractionComplex',
                            ['_perceive_amide_groups', '_assign_aromatic_rings_to_residues'], ns)
    self_ = IC.__new__(IC)
    self_.s_atoms = atoms
    self_.biopython_str = core.Structure(residues)
    self_.ob_mol = types.SimpleNamespace(GetAtom=lambda idx: ob_atoms[idx])
    self_.ob_to_bio = {7000 + i: a for i, a in enumerate(atoms)}
    self_._perceive_amide_groups()                                   # interactions.py:1531-1589, executed
    am = self_.biopython_str.amides
    assert list(am.keys()) == list(range(len(amide_atoms)))
    out[name + '/xyz'] = pc.xyz
    out[name + '/res_id'] = pc.res_id
    out[name + '/amide_atoms'] = np.asarray(amide_atoms, np.int32).reshape(-1, 4)
    ctr = np.array([am[k]['center'] for k in am]).reshape(-1, 3)
    nrm = np.array([am[k]['normal'] for k in am]).reshape(-1, 3)
    assert ctr.dtype == np.float32 and nrm.dtype == np.float32, (ctr.dtype, nrm.dtype)
    out[name + '/amide_center'] = ctr
    out[name + '/amide_normal'] = nrm
    out[name + '/amide_res'] = np.array([am[k]['residue'].idx for k in am], np.int32)
    for k in am:
        assert np.array_equal(am[k]['normal_opp'], -am[k]['normal']) and [a.idx for a in am[k]['atoms']] == list(amide_atoms[k])
    # rings: the centre is OpenBabel's (an input here); the residue assignment is arpeggio's
    for r, c in enumerate(ring_center):
        self_.biopython_str.rings[r] = {'ring_id': r, 'center': np.asarray(c, np.float64)}
    self_._assign_aromatic_rings_to_residues()                       # interactions.py:1453-1492, executed
    rg = self_.biopython_str.rings
    out[name + '/ring_center'] = np.asarray(ring_center, np.float64).reshape(-1, 3)
    out[name + '/ring_res'] = np.array([-1 if rg[r]['residue'] is None else rg[r]['residue'].idx for r in rg], np.int32)
    dist = np.array([rg[r].get('residue_shortest_distance', -1.0) for r in rg], np.float64)
    out[name + '/ring_dist'] = dist
    # residue.rings lists (I:1488-1491)
    per_res = [getattr(res, 'rings', []) for res in residues]
    out[name + '/res_ring_count'] = np.array([len(x) for x in per_res], np.int32)
    return len(am), len(rg)


def main():
    out = {}
    rs = np.random.RandomState(17)
    names = []
    # 1. the 1tqn_h stand-in: real amide quadruples (N, C, O, CA of every peptide bond) and its own ring centres
    pc = synth.proteinlike()
    run(pc, pc.amide_atoms, pc.ring_center, 'proteinlike', out); names.append('proteinlike')
    # ring centres displaced so that some have no atom within 3 A (residue None), some sit exactly between two atoms
    far = pc.ring_center + rs.normal(scale=2.5, size=pc.ring_center.shape)
    mid = 0.5 * (pc.xyz[pc.ring_atoms[0][0]].astype(np.float64) + pc.xyz[pc.ring_atoms[0][1]].astype(np.float64))
    run(pc, pc.amide_atoms[:5], np.vstack([far, mid[None], pc.xyz[10].astype(np.float64)[None], [[500.0, 500.0, 500.0]]]),
        'proteinlike_displaced', out); names.append('proteinlike_displaced')
    # 2. random soups: arbitrary quadruples (nearly collinear and nearly coincident ones included: ill-conditioned planes)
    for k in range(4):
        pcs = synth.make_synthetic(600, seed=40 + k, box=(22, 22, 22), n_rings=12, n_amides=0)
        quads = []
        for _ in range(60):
            a = rs.randint(0, pcs.n_atoms)
            d = np.linalg.norm(pcs.xyz - pcs.xyz[a], axis=1)
            near = np.argsort(d)[1:12]
            quads.append([a] + rs.choice(near, 3, replace=False).tolist())
        # the asserts of I:1548-1552 read the element of the four atoms: matches must not share atoms, roles N, C, O, C
        el = np.array(['C'] * pcs.n_atoms, dtype=object)
        used = set()
        good = []
        for q in quads:
            if any(x in used for x in q) or len(set(q)) < 4:
                continue
            used.update(q)
            el[q[0]], el[q[1]], el[q[2]], el[q[3]] = 'N', 'C', 'O', 'C'
            good.append(q)
        pcs.element = el.tolist()
        name = f'soup{k}'
        run(pcs, np.array(good, np.int32), pcs.ring_center + rs.normal(scale=1.0, size=pcs.ring_center.shape), name, out); names.append(name)
    out['names'] = np.array(names)
    np.savez_compressed(os.path.join(OUT, 'prepare_cases.npz'), **out)
    print({n: (len(out[n + '/amide_atoms']), len(out[n + '/ring_res']), int((out[n + '/ring_res'] < 0).sum())) for n in names})


if __name__ == '__main__':
    main()
