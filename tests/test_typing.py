"""Table-driven atom typing and element flags against the executed reference code
(tests/golden/make_golden_typing.py: _ob_atom_typing, address_ambiguities, _extend_atom_properties).  CPU only."""
import json
import os

import numpy as np
import pytest

from arpeggio_amd.core import config, typing
from helpers import tiny_complex


def _pack(rows, smarts_types):
    res_id, names = [], []
    last = None
    for r in rows:
        key = (r['resname'], r['water'])
        if key != last or r['water']:
            names.append(r['resname'])
            last = key
        res_id.append(len(names) - 1)
    mask = [sum(config.ATOM_TYPE_BIT[t] for t in ts) for ts in smarts_types]
    flags = [config.F_WATER if r['water'] else 0 for r in rows]
    pc = tiny_complex(np.zeros((len(rows), 3), np.float32), type_mask=mask, flags=flags, res_id=res_id)
    pc.res_name = names
    pc.atom_name = [r['name'] for r in rows]
    return pc


@pytest.mark.parametrize('variant', ['default', 'ambiguities'])
def test_protein_typing_equals_executed_reference(golden_dir, variant):
    g = json.load(open(os.path.join(golden_dir, 'typing.json')))[variant]
    pc = _pack(g['atoms'], g['smarts_types'])
    got = typing.apply_protein_typing(pc, use_ambiguities=(variant == 'ambiguities'))
    want = np.array([sum(config.ATOM_TYPE_BIT[t] for t in ts) for ts in g['final_types']], np.uint16)
    assert np.array_equal(got, want), [(g['atoms'][k], k) for k in np.nonzero(got != want)[0][:5]]
    assert (got != pc.type_mask).sum() > 100                    # the dictionary really overrides the matcher
    by = {(a['resname'], a['name']): k for k, a in enumerate(g['atoms'])}
    T = config.ATOM_TYPE_BIT
    # the accidents of the dictionary travel as data: the keys fused by a missing comma are in the table (and match no atom)
    assert {'GLNOE1GLNNE2', 'TRPCD1TRPCE3', 'TRYCB'} <= set(typing.table()['keys'])
    # ... a residue name with a space is not a standard residue (membership is tested unstripped, I:1968) ...
    assert got[by[('GLN ', 'OE1')]] == pc.type_mask[by[('GLN ', 'OE1')]]
    # ... waters are donors and acceptors whatever the matcher said
    w = got[by[('HOH', 'O')]]
    assert w & T['hbond donor'] and w & T['hbond acceptor']
    # 'xbond donor' is not a dictionary type: it survives the override on standard residues
    k = next(k for k, a in enumerate(g['atoms']) if a['resname'] in typing.table()['std_res'] and pc.type_mask[k] & T['xbond donor'])
    assert got[k] & T['xbond donor']


def test_ambiguity_variant_differs_only_on_the_struck_keys(golden_dir):
    g = json.load(open(os.path.join(golden_dir, 'typing.json')))
    a, b = typing.key_masks(False), typing.key_masks(True)
    changed = {k for k in a if a[k] != b[k]}
    assert changed and changed <= {'ASNND2', 'GLNNE2', 'HISCE1', 'HISCD2', 'ASNOD1', 'GLNOE1'}   # (GLNNE2 / GLNOE1 partly hide in the fused key)
    assert g['default']['final_types'] != g['ambiguities']['final_types']


def test_element_flags_equal_executed_reference(golden_dir):
    g = json.load(open(os.path.join(golden_dir, 'typing.json')))['element_flags']
    f = typing.element_flags([r['element'] for r in g])
    for r, v in zip(g, f):
        assert bool(v & config.F_METAL) == r['is_metal'] and bool(v & config.F_HALOGEN) == r['is_halogen'], r


def test_element_radii_table():
    vdw, cov = typing.element_radii(['C', 'n', ' O ', 'FE', 'H'])
    assert vdw.tolist() == [1.70, 1.55, 1.52, 2.05, 1.10] and cov.tolist() == [0.76, 0.71, 0.66, 1.32, 0.31]
    with pytest.raises(KeyError):
        typing.element_radii(['C', 'XX'])


def test_interaction_complex_address_ambiguities_retypes_standard_residues():
    """The drop-in method (I:120-133): afterwards the atoms of standard residues carry the types of the dictionary with the
    ASN / GLN / HIS keys struck, ligand atoms keep theirs."""
    from arpeggio_amd import synth
    from arpeggio_amd.core import InteractionComplex, typing as ty
    pc = synth.proteinlike(n_res=150, n_waters=5)
    lig = np.array([pc.res_name[r] not in ty.table()['std_res'] and pc.res_name[r] != 'HOH' for r in pc.res_id])
    before = pc.type_mask.copy()
    ic = InteractionComplex(pc)
    ic.address_ambiguities()
    assert np.array_equal(pc.type_mask, ty.apply_protein_typing(pc, use_ambiguities=True))
    assert lig.any() and np.array_equal(pc.type_mask[lig], before[lig])
    acc = np.uint16(1) << np.uint16(0)
    nd2 = [i for i in range(pc.n_atoms) if pc.res_name[pc.res_id[i]] == 'ASN' and pc.atom_name[i] == 'ND2']
    if nd2:       # struck from 'hbond acceptor' by the ambiguity rule, still a donor
        from arpeggio_amd.core import config
        assert all(not (pc.type_mask[i] & config.ATOM_TYPE_BIT['hbond acceptor']) for i in nd2)
