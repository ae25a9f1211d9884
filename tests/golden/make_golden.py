#!/usr/bin/env python3
"""Generate golden input/output vectors by EXECUTING the reference's own code.

Run in the build container only (needs /root/reference; the GPU box never sees it):

    python tests/golden/make_golden.py

How: the reference package cannot be imported (its modules import BioPython,
OpenBabel and gemmi at load time, none of which is installed).  This script therefore
parses arpeggio/core/utils.py and arpeggio/core/interactions.py with ``ast`` *where
they lie* under /root/reference, compiles the individual function definitions that
contain no BioPython/OpenBabel call, and runs them on plain data holders.  No module,
class or function of the missing libraries is stood in for: the executed bodies touch
only NumPy, ``config`` (the reference's config.py, loaded by path; it has no
third-party imports) and the attributes of the data holders.

Only inputs and outputs are written (``*.npz`` / ``*.json`` next to this script);
no reference source text is stored anywhere in this repository.

Pinned by this: get_angle, group_angle, group_group_angle, is_hbond, is_weak_hbond,
update_atom_sift / _fsift / _integer_sift, is_digit + selection_parser (utils.py);
__get_contact_type, __calculate_plane_plane_contacts, __calculate_group_group_contacts,
__calculate_group_plane_contacts (interactions.py); np.linalg.norm on float32 (NumPy).
NOT pinned (need BioPython/OpenBabel objects): _calculate_atom_contacts,
__calculate_atom_plane_contacts, _make_selection, is_xbond, is_halogen_weak_hbond.
"""
import ast
import collections
import importlib.util
import json
import logging
import os
import sys
import types

import numpy as np

REF = '/root/reference/arpeggio/core'
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get('ARP_GOLDEN_OUT', HERE)      # (tests/test_fixture_freshness.py regenerates into a scratch directory)
sys.path.insert(0, os.path.join(HERE, '..', '..'))


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def extract_functions(path, names, namespace, in_class=None):
    """Compile the FunctionDef nodes called `names` from `path` into `namespace`."""
    tree = ast.parse(open(path).read(), filename=path)
    body = tree.body
    if in_class:
        body = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == in_class).body
    found = {}
    for node in body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, path, 'exec'), namespace)
            found[node.name] = namespace[node.name]
    missing = set(names) - set(found)
    if missing:
        raise RuntimeError(f'not found in {path}: {missing}')
    return found


def extract_assignments(path, names, namespace):
    tree = ast.parse(open(path).read(), filename=path)
    for node in tree.body:
        if isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id in names for t in node.targets):
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, 'exec'), namespace)


def main():
    logging.disable(logging.CRITICAL)
    config = load_by_path('ref_config', os.path.join(REF, 'config.py'))
    exceptions = load_by_path('ref_exceptions', os.path.join(REF, 'exceptions.py'))

    # ---- utils.py functions ----
    uns = {'np': np, 'logging': logging, 'config': config, 'collections': collections,
           'SelectionError': exceptions.SelectionError}
    U = extract_functions(os.path.join(REF, 'utils.py'),
                          ['get_angle', 'group_angle', 'group_group_angle', 'is_hbond', 'is_weak_hbond',
                           'update_atom_sift', 'update_atom_fsift', 'update_atom_integer_sift', 'is_digit',
                           'selection_parser'], uns)
    utils = types.SimpleNamespace(**U)

    # ---- interactions.py methods (as plain functions taking `self`) ----
    ins = {'np': np, 'logging': logging, 'config': config, 'collections': collections, 'utils': utils}
    extract_assignments(os.path.join(REF, 'interactions.py'), ['PlanePlaneContact', 'AtomPlaneContact', 'AtomAtomContact'], ins)
    I = extract_functions(os.path.join(REF, 'interactions.py'),
                          ['__get_contact_type', '__calculate_plane_plane_contacts',
                           '__calculate_group_group_contacts', '__calculate_group_plane_contacts'],
                          ins, in_class='InteractionComplex')

    rng = np.random.default_rng(20260927)
    f32, f64 = np.float32, np.float64

    # ------------------------------------------------------------------ angles
    K = 4000
    a = (rng.random((K, 3)) * 20).astype(f32)
    b = (a + rng.standard_normal((K, 3)) * 1.5).astype(f32)
    c = (b + rng.standard_normal((K, 3)) * 2.0).astype(f32)
    # degenerate rows: collinear (NaN-prone), coincident points
    b[0] = a[0] + f32(1.0); c[0] = a[0] + f32(2.0)          # collinear, 180 deg
    c[1] = a[1]                                              # angle 0
    b[2] = a[2]                                              # zero-length v1 -> NaN -> pi
    for k in range(3, 40):                                   # exactly collinear integer points
        a[k] = np.array([k, 2 * k, 3 * k], f32); b[k] = a[k] * f32(2); c[k] = a[k] * f32(3 + (k % 2) * -4)
    bh = b.astype(f64) + rng.standard_normal((K, 3)) * 1e-3  # float64 "hydrogen" points
    bh[2] = a[2].astype(f64)
    ch = c.astype(f64) + rng.standard_normal((K, 3)) * 1e-3
    with np.errstate(all='ignore'):
        ang_f32 = np.array([float(utils.get_angle(a[k], b[k], c[k])) for k in range(K)])            # U:174 usage
        ang_f64 = np.array([float(utils.get_angle(a[k], bh[k], c[k])) for k in range(K)])           # U:90 usage
        ang_mix = np.array([float(utils.get_angle(a[k], b[k], ch[k])) for k in range(K)])           # U:151 usage
    np.savez_compressed(os.path.join(OUT, 'angles.npz'), a=a, b=b, c=c, bh=bh, ch=ch,
                        ang_f32=ang_f32, ang_f64=ang_f64, ang_mix=ang_mix)

    # ------------------------------------------------------------- group angles
    n64 = rng.standard_normal((K, 3)); n64 /= np.linalg.norm(n64, axis=1, keepdims=True)
    m64 = rng.standard_normal((K, 3)); m64 /= np.linalg.norm(m64, axis=1, keepdims=True)
    p64 = rng.standard_normal((K, 3)) * 3
    m64[:50] = n64[:50]            # parallel normals (cos may round above 1 -> NaN)
    m64[50:100] = -n64[50:100]     # antiparallel
    p64[:50] = n64[:50] * 4.0
    n32, m32, p32 = n64.astype(f32), m64.astype(f32), p64.astype(f32)
    with np.errstate(all='ignore'):
        ga_64 = np.array([float(abs(utils.group_angle({'normal': n64[k]}, p64[k], True, True))) for k in range(K)])
        ga_32 = np.array([abs(utils.group_angle({'normal': n32[k]}, p32[k], True, True)) for k in range(K)])
        ga_3264 = np.array([float(abs(utils.group_angle({'normal': n32[k]}, p64[k], True, True))) for k in range(K)])
        gg_64 = np.array([float(abs(utils.group_group_angle({'normal': n64[k]}, {'normal': m64[k]}, True, True))) for k in range(K)])
        gg_32 = np.array([abs(utils.group_group_angle({'normal': n32[k]}, {'normal': m32[k]}, True, True)) for k in range(K)])
        gg_3264 = np.array([float(abs(utils.group_group_angle({'normal': n32[k]}, {'normal': m64[k]}, True, True))) for k in range(K)])
    assert ga_32.dtype == np.float32 and gg_32.dtype == np.float32
    np.savez_compressed(os.path.join(OUT, 'group_angles.npz'), n64=n64, m64=m64, p64=p64, n32=n32, m32=m32, p32=p32,
                        ga_64=ga_64, ga_32=ga_32, ga_3264=ga_3264, gg_64=gg_64, gg_32=gg_32, gg_3264=gg_3264)

    # --------------------------------------------------------- float32 distance
    norm32 = np.array([np.linalg.norm(a[k] - c[k]) for k in range(K)])
    assert norm32.dtype == np.float32
    np.savez_compressed(os.path.join(OUT, 'norm_f32.npz'), a=a, c=c, dist=norm32)

    # ------------------------------------------------------- is_hbond / is_weak
    KH = 3000
    don = (rng.random((KH, 3)) * 10).astype(f32)
    nh = rng.integers(0, 4, KH)
    hoff = np.concatenate([[0], np.cumsum(nh)]).astype(np.int32)
    hdir = rng.standard_normal((int(nh.sum()), 3)); hdir /= np.linalg.norm(hdir, axis=1, keepdims=True)
    hxyz = np.repeat(don.astype(f64), nh, axis=0) + hdir * 1.0
    acc = (don + (rng.standard_normal((KH, 3)) * 1.8)).astype(f32)
    acc_vdw = rng.choice([1.52, 1.55, 1.7, 1.8], KH)
    comp = 0.1
    res_h = np.zeros(KH, np.int8); res_w = np.zeros(KH, np.int8)
    for k in range(KH):
        d = types.SimpleNamespace(coord=don[k], h_coords=[hxyz[j] for j in range(hoff[k], hoff[k + 1])])
        ac = types.SimpleNamespace(coord=acc[k], vdw_radius=float(acc_vdw[k]))
        res_h[k] = utils.is_hbond(d, ac, comp)
        res_w[k] = utils.is_weak_hbond(d, ac, comp)
    np.savez_compressed(os.path.join(OUT, 'hbond.npz'), don=don, hoff=hoff, hxyz=hxyz, acc=acc, acc_vdw=acc_vdw,
                        comp=np.float64(comp), is_hbond=res_h, is_weak_hbond=res_w)

    # ------------------------------------------------------- __get_contact_type
    class FAtom:
        def __init__(self, idx, water):
            self.idx, self.water = idx, water

        def get_full_id(self):
            return ('s', 0, 'A', ('W' if self.water else ' ', self.idx, ' '), ('X', ' '))

        def __hash__(self):
            return hash(self.idx)

        def __eq__(self, o):
            return self.idx == o.idx

        def __repr__(self):
            return f'<Atom {self.idx}>'

    ct_rows = []
    for bs in (0, 1):
        for es in (0, 1):
            for bw in (0, 1):
                for ew in (0, 1):
                    A, B = FAtom(0, bw), FAtom(1, ew)
                    sel = set()
                    if bs: sel.add(A)
                    if es: sel.add(B)
                    ct_rows.append({'bgn_sel': bs, 'end_sel': es, 'bgn_water': bw, 'end_water': ew,
                                    'contact_type': I['__get_contact_type'](None, A, B, sel)})
    json.dump(ct_rows, open(os.path.join(OUT, 'contact_type.json'), 'w'), indent=1)

    # ------------------------------------------------------------- sift updates
    ctypes_all = ['INTRA_NON_SELECTION', 'INTRA_SELECTION', 'INTER', 'SELECTION_WATER', 'NON_SELECTION_WATER', 'WATER_WATER']
    cases = []
    for case in range(40):
        at = types.SimpleNamespace()
        for nm in ('sift', 'sift_inter_only', 'sift_intra_only', 'sift_water_only', 'integer_sift',
                   'integer_sift_inter_only', 'integer_sift_intra_only', 'integer_sift_water_only'):
            setattr(at, nm, [0] * 15)
        for nm in ('actual_fsift', 'actual_fsift_inter_only', 'actual_fsift_intra_only', 'actual_fsift_water_only'):
            setattr(at, nm, [0] * 10)
        steps = []
        for _ in range(int(rng.integers(1, 8))):
            add = [int(x) for x in (rng.random(15) < 0.25)]
            ct = ctypes_all[int(rng.integers(0, 6))]
            # call order of interactions.py:924-934
            utils.update_atom_integer_sift(at, add, ct)
            utils.update_atom_sift(at, add, ct)
            utils.update_atom_fsift(at, add[5:], ct)
            steps.append({'addition': add, 'contact_type': ct})
        cases.append({'steps': steps, 'final': {k: [int(x) for x in v] for k, v in vars(at).items()}})
    json.dump(cases, open(os.path.join(OUT, 'sift_updates.json'), 'w'))

    # ------------------------------------------- plane / group loops (real code)
    from arpeggio_amd import synth
    pc = synth.make_synthetic(0, seed=11, box=(22.0, 22.0, 22.0), n_rings=260, n_amides=300)
    R, A = pc.n_rings, pc.n_amides
    # make some rings share a residue (intra-residue EE rule, I:1154) and some residue None
    ring_res = pc.ring_res.copy()
    ring_res[1:60:2] = ring_res[0:59:2]
    amide_res = pc.amide_res.copy()
    amide_res[1:40:2] = amide_res[0:39:2]
    amide_res[200:230] = ring_res[200:230]       # amide and ring in one residue
    nres = pc.n_residues

    class Res:
        def __init__(self, idx):
            self.idx = idx
            self.ring_ring_inter_integer_sift = [0] * 9
            self.amide_amide_inter_integer_sift = [0]
            self.amide_ring_inter_integer_sift = [0]
            self.ring_amide_inter_integer_sift = [0]

        def __eq__(self, o):
            return isinstance(o, Res) and self.idx == o.idx

        def __hash__(self):
            return hash(self.idx)

    class NamedAtom:
        def __init__(self, name):
            self.name = name

        def get_id(self):
            return self.name

    residues = [Res(i) for i in range(nres)]
    rings = collections.OrderedDict()
    for r in range(R):
        rings[r] = {'ring_id': r, 'center': np.array(pc.ring_center[r]), 'normal': np.array(pc.ring_normal[r]),
                    'atoms': [NamedAtom(f'C{k}') for k in range(6)], 'residue': residues[ring_res[r]]}
    amides = collections.OrderedDict()
    for e in range(A):
        amides[e] = {'amide_id': e, 'center': pc.amide_center[e].copy(), 'normal': pc.amide_normal[e].copy(),
                     'atoms': [NamedAtom(x) for x in ('N', 'C', 'O', 'CA')], 'residue': residues[amide_res[e]]}
    assert rings[0]['center'].dtype == np.float64 and amides[0]['center'].dtype == np.float32
    res_plus = rng.random(nres) < 0.8
    res_sel = res_plus & (rng.random(nres) < 0.3)
    ring_plus = res_plus[ring_res]; ring_sel = res_sel[ring_res]
    amide_plus = res_plus[amide_res]; amide_sel = res_sel[amide_res]
    self_ = types.SimpleNamespace(
        biopython_str=types.SimpleNamespace(rings=rings, amides=amides),
        selection_ring_ids={r for r in range(R) if ring_sel[r]},
        selection_plus_ring_ids={r for r in range(R) if ring_plus[r]},
        selection_amide_ids={e for e in range(A) if amide_sel[e]},
        selection_plus_amide_ids={e for e in range(A) if amide_plus[e]},
        plane_plane_contacts=[], group_group_contacts=[], group_plane_contacts=[])
    with np.errstate(all='ignore'):
        I['__calculate_plane_plane_contacts'](self_)
        I['__calculate_group_group_contacts'](self_)
        I['__calculate_group_plane_contacts'](self_)

    def dump(lst):
        return [{'bgn_id': int(x.bgn_id), 'end_id': int(x.end_id), 'distance': float(x.distance),
                 'distance_dtype': str(np.asarray(x.distance).dtype),
                 'contact_type': list(x.contact_type), 'text': x.text,
                 'bgn_res': int(x.bgn_res.idx), 'end_res': int(x.end_res.idx),
                 'bgn_res_atoms': list(x.bgn_res_atoms), 'end_res_atoms': list(x.end_res_atoms)} for x in lst]

    np.savez_compressed(os.path.join(OUT, 'planes_input.npz'),
                        ring_center=pc.ring_center, ring_normal=pc.ring_normal, ring_res=ring_res,
                        ring_sel=ring_sel.astype(np.uint8), ring_plus=ring_plus.astype(np.uint8),
                        amide_center=pc.amide_center, amide_normal=pc.amide_normal, amide_res=amide_res,
                        amide_sel=amide_sel.astype(np.uint8), amide_plus=amide_plus.astype(np.uint8),
                        nres=np.int64(nres))
    json.dump({'plane_plane': dump(self_.plane_plane_contacts),
               'group_group': dump(self_.group_group_contacts),
               'group_plane': dump(self_.group_plane_contacts),
               'ring_ring_inter_integer_sift': {str(r.idx): r.ring_ring_inter_integer_sift for r in residues if any(r.ring_ring_inter_integer_sift)},
               'amide_amide_inter_integer_sift': {str(r.idx): r.amide_amide_inter_integer_sift for r in residues if any(r.amide_amide_inter_integer_sift)},
               'amide_ring_inter_integer_sift': {str(r.idx): r.amide_ring_inter_integer_sift for r in residues if any(r.amide_ring_inter_integer_sift)},
               'ring_amide_inter_integer_sift': {str(r.idx): r.ring_amide_inter_integer_sift for r in residues if any(r.ring_amide_inter_integer_sift)},
               }, open(os.path.join(OUT, 'planes_expected.json'), 'w'))
    print('planes:', len(self_.plane_plane_contacts), len(self_.group_group_contacts), len(self_.group_plane_contacts))

    # --------------------------------------------------------- selection_parser
    class Chain:
        def __init__(self, cid):
            self.id = cid

    class Residue:
        def __init__(self, chain, resname, het, seq, icode, poly):
            self.chain, self.resname, self.id, self.is_polypeptide, self.child_list = chain, resname, (het, seq, icode), poly, []

        def get_parent(self):
            return self.chain

    class PAtom:
        def __init__(self, idx, res, name, element):
            self.idx, self.res, self.name, self.element = idx, res, name, element
            res.child_list.append(self)

        def get_parent(self):
            return self.res

    chains = {c_: Chain(c_) for c_ in 'AB'}
    table = [  # chain, resname, het, seq, icode, poly, [(name, element)...]
        ('A', 'ALA', ' ', 12, ' ', True, [('N', 'N'), ('CA', 'C'), ('C', 'C'), ('O', 'O'), ('CB', 'C')]),
        ('A', 'GLY', ' ', 12, 'B', True, [('N', 'N'), ('CA', 'C'), ('C', 'C'), ('O', 'O')]),
        ('A', 'SER', ' ', 0, ' ', True, [('N', 'N'), ('CA', 'C'), ('OG', 'O')]),
        ('A', 'HEM', 'H_HEM', 508, ' ', False, [('FE', 'FE'), ('C1A', 'C'), ('C2A', 'C'), ('NA', 'N'), ('O1A', 'O'), ("O2'", 'O')]),
        ('A', 'HOH', 'W', 523, ' ', False, [('O', 'O')]),
        ('B', 'ALA', ' ', 12, ' ', True, [('N', 'N'), ('CA', 'C'), ('C', 'C'), ('O', 'O'), ('CB', 'C')]),
        ('B', 'GOL', 'H_GOL', 601, ' ', False, [('C1', 'C'), ('C2', 'C'), ('C3', 'C'), ('O1', 'O'), ('O2', 'O'), ('O3', 'O')]),
        ('B', 'LIG', 'H_LIG', 602, ' ', False, [('C1', 'C'), ('C2', 'C'), ('N1', 'N'), ('O1', 'O'), ('S1', 'S')]),
        ('B', 'ZN', 'H_ZN', 603, ' ', False, [('ZN', 'ZN')]),
        ('B', 'DA', ' ', 700, ' ', False, [('P', 'P'), ('C1\'', 'C'), ('N9', 'N'), ('C4', 'C'), ('C5', 'C')]),
        ('B', '+U', 'H_+U', 701, ' ', False, [('P', 'P'), ('C1\'', 'C'), ('N1', 'N'), ('C4', 'C'), ('C5', 'C')]),
    ]
    atoms, rows = [], []
    for (cid, rn, het, seq, ic, poly, ats) in table:
        res = Residue(chains[cid], rn, het, seq, ic, poly)
        for (nm, el) in ats:
            atoms.append(PAtom(len(atoms), res, nm, el))
            rows.append({'chain': cid, 'resname': rn, 'het': het, 'seq': seq, 'icode': ic, 'poly': poly, 'name': nm, 'element': el})
    selectors = [['/A/508/'], ['/A//'], ['//12/CA'], ['/A/12B/'], ['/A/0/'], ['RESNAME:HEM'], ['RESNAME: GOL '],
                 ['LIGANDS'], ['/B//', '/A/12/N'], ['///'], ["/A/508/O2'"], ['/B/603/ZN'], [' /A/523/ '],
                 ['A/1/'], ['/A/1'], ['RESNAME:ABCD'], ['/A/x!/'], ['/A/12/C@'], ['/C//'], ['/A/999/'], ['/A/1B2/'],
                 ['HET:HEM'], ['/A/12BB/']]
    sel_out = []
    for s in selectors:
        try:
            got = utils.selection_parser(list(s), atoms)
            sel_out.append({'selectors': s, 'atoms': sorted(a_.idx for a_ in got)})
        except exceptions.SelectionError as e:
            sel_out.append({'selectors': s, 'error': 'SelectionError', 'args': [str(x) for x in e.args]})
    json.dump({'atoms': rows, 'cases': sel_out}, open(os.path.join(OUT, 'selection_parser.json'), 'w'), indent=1)
    print('golden vectors written to', HERE)


if __name__ == '__main__':
    main()
