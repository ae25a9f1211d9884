"""Two independent restatements must agree: oracle/ref_py.py (real NumPy calls, the reference's idiom) and
oracle/ref_c.c (explicit arithmetic model).  Agreement to the last bit re-validates the model of NumPy's
float32 / float64 dot products and NEP-50 comparisons on this machine.  CPU only."""
import numpy as np
import pytest

import oracle
from oracle.ref_py import RefPy
from helpers import known_answer_packs, random_dense_pack


def _same(a, b, name=''):
    assert len(a['i']) == len(b['i']), name
    for k in ('i', 'j', 'sift', 'ctype'):
        assert np.array_equal(a[k], b[k]), (name, k)
    assert np.array_equal(a['dist'].view(np.uint32), b['dist'].view(np.uint32)), name


def test_known_answer_packs_both_orientations():
    for name, pc in known_answer_packs():
        oc = oracle.OracleComplex(pc)
        rp = RefPy(pc)
        for b in range(pc.n_atoms):
            for e in range(pc.n_atoms):
                if b == e:
                    continue
                ok, d, s, ct, err = oc.pair_contact(b, e)
                try:
                    r = rp.pair(b, e)
                except AttributeError:
                    assert err == -4, name
                    continue
                assert err == 0, name
                assert (r is not None) == ok, (name, b, e)
                if ok:
                    assert (np.float32(r[0]).view(np.uint32), r[1], r[2]) == (np.float32(d).view(np.uint32), s, ct), (name, b, e)


@pytest.mark.parametrize('seed', [1, 2, 3])
def test_dense_soup_full_loop(seed):
    pc = random_dense_pack(seed, n=260, box=12.0)
    rng = np.random.default_rng(seed)
    sel = (rng.random(pc.n_atoms) < 0.4).astype(np.uint8)
    oc = oracle.OracleComplex(pc)
    plus = oc.make_selection(sel, use_grid=False)
    rp = RefPy(pc, sel, plus)
    for seq_adj, comp in ((False, 0.1), (True, 0.33)):
        _same(rp.atom_contacts(5.0, comp, seq_adj), oc.atom_contacts(5.0, comp, seq_adj, use_grid=False), f'seed{seed}')
