"""Host-side export of the result bags: the JSON records of ``get_contacts`` and the legacy CSV tables.

Everything here works on the struct-of-arrays bags the HIP library returns (one NumPy array per column) and on the
string tables of the PackedComplex; no per-contact Python objects are built on the way.  The output is what the
reference's ``get_contacts`` (interactions.py:172-212, 2063-2113) and ``write_*`` methods (interactions.py:135-170,
349-366, 405-466 with ``_calc_residue_sifts`` 471-574) produce, and is pinned byte for byte against the text those
methods wrote when executed on the same structure (tests/golden/core_exports.json.gz).

Order of records inside a bag: atom-atom by (bgn, end) packed index with bgn = the lower one (the canonical order of
this library; the reference's own order is its KD-tree's), the plane bags by (bgn id, end id) — the reference's creation
order.  Tables that the reference writes by iterating a ``set`` (atom sifts, residue sifts, polar matching) come out in
packed atom / residue order.
"""
from __future__ import annotations

import csv
import logging
import os

import numpy as np

from . import config

_SIFT_FLAG_COLUMNS = ('clash', 'covalent', 'vdw_clash', 'vdw', 'proximal', 'hbond', 'weak_hbond', 'halogen_bond', 'ionic',
                      'metal_complex', 'aromatic', 'hydrophobic', 'carbonyl', 'polar', 'weak_polar')
_CLASS_PREFIX = ('', 'inter_only_', 'intra_only_', 'water_only_')
_PI_COLUMNS = ('carbonpi', 'cationpi', 'donorpi', 'halogenpi', 'metsulphurpi')
_BS_CONTACT_TYPES = ('INTER', 'INTRA_SELECTION', 'SELECTION_WATER', 'WATER_WATER')      # interactions.py:166


def residue_sift_header():
    """Column names of '<id>_residue_sifts.csv' (config.py:714-1141 of the reference: 426 names, regular)."""
    h = ['residue', 'is_polypeptide']
    for integer in ('', 'integer_'):
        for part in ('', 'mc_', 'sc_'):
            for cls in _CLASS_PREFIX:
                h += [f'residue_{part}{integer}{cls}{flag}' for flag in _SIFT_FLAG_COLUMNS]
    for integer in ('', 'integer_'):
        h += [f'residue_ring_ring_inter_{integer}{t}' for t in config.PLANE_PLANE_NAMES[:9]]
        for who in ('ring_atom', 'atom_ring', 'mc_atom_ring', 'sc_atom_ring'):
            h += [f'residue_{who}_inter_{integer}{t}' for t in _PI_COLUMNS]
    for integer in ('', '_integer'):
        h += [f'residue_{who}_inter{integer}' for who in ('amide_ring', 'ring_amide', 'amide_amide')]
    return h


class Labels:
    """Per-atom / per-residue strings of a PackedComplex, computed once per structure."""

    def __init__(self, pc, component_types):
        pc.ensure_labels()
        self.pc = pc
        nr = pc.n_residues
        seq = [int(x) for x in pc.res_seq]
        self.res_json = [(pc.res_name[r], seq[r], pc.res_chain[r], pc.res_icode[r]) for r in range(nr)]
        self.res_comp_type = [component_types[pc.res_name[r]] for r in range(nr)] if nr else []   # KeyError like I:186
        self.res_macro = ['{}/{}/'.format(pc.res_chain[r], seq[r] if pc.res_icode[r] == ' ' else str(seq[r]) + pc.res_icode[r])
                          for r in range(nr)]
        self.atom_res = pc.res_id.tolist()
        self.atom_name = list(pc.atom_name)

    def atom_macro(self, i):
        return self.res_macro[self.atom_res[i]] + self.atom_name[i]

    def atom_dict(self, i):
        n, s, c, ic = self.res_json[self.atom_res[i]]
        return {'label_comp_id': n, 'auth_seq_id': s, 'auth_asym_id': c, 'auth_atom_id': self.atom_name[i],
                'pdbx_PDB_ins_code': ic, 'label_comp_type': self.res_comp_type[self.atom_res[i]]}

    def plane_dict(self, r, atom_ids):
        if r < 0:
            raise TypeError('Cannot make a json object from non-Atom/Residue object.')   # ring without residue (I:1479; U:563)
        n, s, c, ic = self.res_json[r]
        return {'label_comp_id': n, 'auth_seq_id': s, 'auth_asym_id': c, 'pdbx_PDB_ins_code': ic,
                'label_comp_type': self.res_comp_type[r], 'auth_atom_id': atom_ids}


def ring_atom_ids(pc, r):
    """Comma-joined sorted atom names of a ring (interactions.py:1059, 2078)."""
    return ','.join(sorted(pc.atom_name[a] for a in pc.ring_atoms[r])) if pc.ring_atoms else ''


def amide_atom_ids(pc, a):
    return ','.join(sorted(pc.atom_name[i] for i in pc.amide_atoms[a] if i >= 0))


def rounded(dist):
    """``round(np.float64(d), 2)`` of every element (NumPy's rint(x * 100) / 100, as np.float64.__round__ does)."""
    return np.round(np.asarray(dist, np.float64), 2).tolist()


import contextlib


@contextlib.contextmanager
def paused_gc():
    """Pause the cyclic garbage collector while a large list of small containers is built (none of them is garbage, and the
    collector would re-scan them every few hundred allocations)."""
    import gc
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was_enabled:
            gc.enable()


_PYEXPORT = False


def _pyexport():
    """The optional C helper (built by ``arpeggio_amd.build``); None when it is not there or ARP_NO_PYEXPORT is set."""
    global _PYEXPORT
    if _PYEXPORT is False:
        _PYEXPORT = None
        if not os.environ.get('ARP_NO_PYEXPORT'):
            try:
                from .. import _pyexport as mod
                _PYEXPORT = mod
            except ImportError:
                pass
    return _PYEXPORT


def contacts_json(pc, bags, component_types, share_atoms=False):
    """The list ``get_contacts`` returns (interactions.py:172-212).  Millions of small containers are created and none of
    them is garbage: the cyclic collector, which would re-scan them every few hundred allocations, is paused meanwhile.
    ``share_atoms``: the atom-atom records of one atom share ONE 'bgn' / 'end' dictionary (and the records of one
    fingerprint one 'contact' list) instead of a copy each — the same JSON, less than half the time, but a caller that
    edits one record's inner dictionary edits them all (the reference builds a fresh one per record: the default)."""
    with paused_gc():
        return _contacts_json(pc, bags, component_types, share_atoms)


def _contacts_json(pc, bags, component_types, share_atoms=False):
    lab = Labels(pc, component_types)
    out = []
    b = bags.get('atom_atom')
    if b is not None and len(b['i']):
        used = np.unique(np.concatenate([b['i'], b['j']]))
        adict = {int(a): lab.atom_dict(int(a)) for a in used.tolist()}
        names = {int(s): [n for k, n in enumerate(config.SIFT_NAMES) if (int(s) >> k) & 1] for s in np.unique(b['sift']).tolist()}
        ct = config.CONTACT_TYPE_NAMES
        fast = _pyexport()
        if fast is not None:      # the same records, built in C (csrc/arp_pyexport.c): ~0.7 us each instead of ~3 us
            atoms = [None] * pc.n_atoms
            for a, dct in adict.items():
                atoms[a] = dct
            by_sift = [None] * (1 << 15)
            for s_, nm in names.items():
                by_sift[s_] = nm
            c32 = lambda a: np.ascontiguousarray(a, np.int32)
            out += fast.atom_atom_records(c32(b['i']), c32(b['j']), np.round(np.asarray(b['dist'], np.float64), 2),
                                          np.ascontiguousarray(b['sift'], np.uint16), np.ascontiguousarray(b['ctype'], np.uint8),
                                          atoms, by_sift, list(ct), bool(share_atoms))
        else:
            cp, cl = ((lambda x: x), (lambda x: x)) if share_atoms else (dict, list)
            out += [{'bgn': cp(adict[i]), 'end': cp(adict[j]), 'type': 'atom-atom', 'distance': d, 'contact': cl(names[s]),
                     'interacting_entities': ct[c]}
                    for i, j, d, s, c in zip(b['i'].tolist(), b['j'].tolist(), rounded(b['dist']), b['sift'].tolist(), b['ctype'].tolist())]

    ring_ids, amide_ids = {}, {}

    def ring(r):
        if r not in ring_ids:
            ring_ids[r] = (int(pc.ring_res[r]), ring_atom_ids(pc, r))
        return lab.plane_dict(*ring_ids[r])

    def amide(a):
        if a not in amide_ids:
            amide_ids[a] = (int(pc.amide_res[a]), amide_atom_ids(pc, a))
        return lab.plane_dict(*amide_ids[a])

    ct = config.CONTACT_TYPE_NAMES
    b = bags.get('plane_plane')
    if b is not None and len(b['bgn']):
        pp = config.PLANE_PLANE_NAMES
        for r1, r2, d, t1, t2, c in zip(b['bgn'].tolist(), b['end'].tolist(), rounded(b['dist']), b['type1'].tolist(),
                                        b['type2'].tolist(), b['ctype'].tolist()):
            out.append({'bgn': ring(r1), 'end': ring(r2), 'type': 'plane-plane', 'distance': d,
                        'contact': [pp[t1]] + ([pp[t2]] if t2 < config.PP_SAME else []), 'interacting_entities': ct[c]})
    b = bags.get('atom_plane')
    if b is not None and len(b['atom']):
        ap = config.ATOM_PLANE_NAMES
        for a, r, d, m, c in zip(b['atom'].tolist(), b['ring'].tolist(), rounded(b['dist']), b['mask'].tolist(), b['ctype'].tolist()):
            out.append({'bgn': lab.atom_dict(a), 'end': ring(r), 'type': 'atom-plane', 'distance': d,
                        'contact': [n for k, n in enumerate(ap) if (m >> k) & 1], 'interacting_entities': ct[c]})
    b = bags.get('group_group')
    if b is not None and len(b['bgn']):
        for a1, a2, d, c in zip(b['bgn'].tolist(), b['end'].tolist(), rounded(b['dist']), b['ctype'].tolist()):
            out.append({'bgn': amide(a1), 'end': amide(a2), 'type': 'group-group', 'distance': d, 'contact': ['AMIDEAMIDE'],
                        'interacting_entities': ct[c]})
    b = bags.get('group_plane')
    if b is not None and len(b['amide']):
        for a, r, d, c in zip(b['amide'].tolist(), b['ring'].tolist(), rounded(b['dist']), b['ctype'].tolist()):
            out.append({'bgn': amide(a), 'end': ring(r), 'type': 'group-plane', 'distance': d, 'contact': ['AMIDERING'],
                        'interacting_entities': ct[c]})
    return out


def write_contacts_json(path, pc, bags, component_types, indent=4):
    """``json.dump(get_contacts(), fh, indent=indent, sort_keys=True)`` (scripts/process_protein_cli.py:184-188) without
    building the records in Python: the atom-atom bag — millions of records on a whole-structure run — is formatted by the
    native library from the result arrays, the few ring / amide records by the json module.  Same bytes."""
    import ctypes as C
    import json

    from .. import _capi
    L = _capi.load_host()      # (host-only entry point: the HIP library, or the g++-built libarpeggio_host.so when that is absent)
    pc.ensure_labels()
    tail = contacts_json(pc, {k: v for k, v in bags.items() if k != 'atom_atom'}, component_types)
    pad = ' ' * indent
    tail_text = ',\n'.join('\n'.join(pad + line for line in json.dumps(r, indent=indent, sort_keys=True).split('\n')) for r in tail)
    b = bags.get('atom_atom')
    n = 0 if b is None else len(b['i'])
    z = np.zeros(0, np.int32)
    ci = np.ascontiguousarray(b['i'], np.int32) if n else z
    cj = np.ascontiguousarray(b['j'], np.int32) if n else z
    dist = np.ascontiguousarray(b['dist'], np.float64) if n else np.zeros(0)     # rounded like np.round(x, 2) by the writer (flags bit 0)
    sift = np.ascontiguousarray(b['sift'], np.uint16) if n else np.zeros(0, np.uint16)
    ctype = np.ascontiguousarray(b['ctype'], np.uint8) if n else np.zeros(0, np.uint8)

    keep = []

    def strings(items):
        """``const char* const*`` over ``items``: one NUL-separated buffer and a NumPy array of addresses into it."""
        enc = [str(v).encode('utf-8') for v in items]
        buf = C.create_string_buffer(b'\0'.join(enc) + b'\0')
        off = np.zeros(max(len(enc), 1), np.uint64)
        if len(enc) > 1:
            np.cumsum(np.fromiter((len(e) + 1 for e in enc[:-1]), np.uint64, len(enc) - 1), out=off[1:])
        ptrs = off + np.uint64(C.addressof(buf))
        keep.append((buf, ptrs))
        return _capi._p(ptrs)

    nr = pc.n_residues
    comp = [component_types[pc.res_name[r]] for r in range(nr)]       # KeyError like I:186
    atom_res = np.ascontiguousarray(pc.res_id, np.int32)
    res_seq = np.ascontiguousarray(pc.res_seq, np.int32)
    rc = L.arp_write_contacts_json(os.fsencode(path), int(indent), 1, n, _capi._p(ci), _capi._p(cj), _capi._p(dist), _capi._p(sift),
                                   _capi._p(ctype), pc.n_atoms, _capi._p(atom_res), strings(pc.atom_name), nr, strings(pc.res_name),
                                   _capi._p(res_seq), strings(pc.res_chain), strings(pc.res_icode), strings(comp),
                                   strings(config.SIFT_NAMES), strings(config.CONTACT_TYPE_NAMES), tail_text.encode('utf-8'), len(tail))
    if rc != 0:
        raise OSError(f'arp_write_contacts_json({path!r}) failed ({rc})')


def _writer(path):
    fh = open(path, 'w')
    return fh, csv.writer(fh, delimiter=',', quotechar='"', quoting=csv.QUOTE_MINIMAL)


_CONTACT_HEADER = ['atom_bgn', 'atom_end', 'distance'] + list(config.SIFT_NAMES) + ['interacting_entities']


def _csv_field(text):
    """One field as csv.writer(quoting=QUOTE_MINIMAL) writes it: quoted, with doubled quotes, only when it holds the delimiter,
    the quote character or a line break."""
    if any(c in text for c in ',"\r\n'):
        return '"' + text.replace('"', '""') + '"'
    return text


def write_contact_file_csv_module(path, pc, lab, bag, rows=None):
    """The table row by row through the csv module (what write_contact_file replaces; kept for the test that compares them)."""
    bits = ((bag['sift'][:, None] >> np.arange(15, dtype=np.uint16)[None, :]) & 1).astype(np.uint8)
    idx = range(len(bag['i'])) if rows is None else rows.tolist()
    i, j, c = bag['i'].tolist(), bag['j'].tolist(), bag['ctype'].tolist()
    dist = bag['dist']      # float32 scalars print with their shortest representation, as the reference's np.float32 does
    fh, w = _writer(path)
    with fh:
        w.writerow(_CONTACT_HEADER)
        w.writerows([lab.atom_macro(i[k]), lab.atom_macro(j[k]), dist[k]] + bits[k].tolist() + [config.CONTACT_TYPE_NAMES[c[k]]]
                    for k in idx)


def write_contact_file(path, pc, lab, bag, rows=None):
    """One '<id>_contacts.csv' style table (interactions.py:606-641); ``rows`` = optional index subset.  The same bytes as the
    csv module writes, assembled from per-atom, per-mask and per-type text fragments (a whole-structure run has a million rows)."""
    sel = slice(None) if rows is None else np.asarray(rows, np.int64)
    i, j = np.asarray(bag['i'])[sel], np.asarray(bag['j'])[sel]
    used = np.unique(np.concatenate([i, j])) if len(i) else np.zeros(0, np.int64)
    macro = {int(a): _csv_field(lab.atom_macro(int(a))) for a in used.tolist()}
    sift = np.asarray(bag['sift'])[sel]
    frag = {int(m): ',' + ','.join('1' if (int(m) >> k) & 1 else '0' for k in range(15)) + ',' for m in np.unique(sift).tolist()}
    ctn = [_csv_field(t) for t in config.CONTACT_TYPE_NAMES]
    dist = np.asarray(bag['dist'], np.float32)[sel]
    dtxt = [str(x) for x in dist]        # str(np.float32): the shortest representation, as the reference's float32 prints
    with open(path, 'w', newline='') as fh:
        fh.write(','.join(_CONTACT_HEADER) + '\r\n')
        ct = np.asarray(bag['ctype'])[sel].tolist()
        body = '\r\n'.join(map('%s,%s,%s%s%s'.__mod__, zip(map(macro.__getitem__, i.tolist()), map(macro.__getitem__, j.tolist()), dtxt,
                                                           map(frag.__getitem__, sift.tolist()), map(ctn.__getitem__, ct))))
        if body:
            fh.write(body + '\r\n')


def write_contacts(wd, sid, pc, bag, component_types, selection_given):
    """interactions.py:151-170: '<id>_contacts.csv' and, when a selection was given, '<id>_bs_contacts.csv'."""
    lab = Labels(pc, component_types)
    write_contact_file(os.path.join(wd, sid + '_contacts.csv'), pc, lab, bag)
    if not selection_given:
        return
    keep = np.isin(bag['ctype'], [config.CONTACT_TYPE_NAMES.index(t) for t in _BS_CONTACT_TYPES])
    write_contact_file(os.path.join(wd, sid + '_bs_contacts.csv'), pc, lab, bag, np.nonzero(keep)[0])


def write_atom_types(wd, sid, pc, component_types):
    """interactions.py:135-149: every atom with the sorted list of its type names."""
    lab = Labels(pc, component_types)
    names = config.ATOM_TYPE_NAMES
    cache = {}
    fh, w = _writer(os.path.join(wd, sid + '_atomtypes.csv'))
    with fh:
        w.writerow(['atom', 'atom_types'])
        for i, m in enumerate(pc.type_mask.tolist()):
            if m not in cache:
                cache[m] = sorted(n for b, n in enumerate(names) if (m >> b) & 1)
            w.writerow([lab.atom_macro(i), cache[m]])


def residue_sift_table(pc, atom_sift_bits, atom_isift, plane_sifts):
    """`_calc_residue_sifts` (interactions.py:471-574) as arrays: dict name -> int array [n_residues, width]; the order of
    the names is the column order of write_residue_sifts (interactions.py:448-464)."""
    nr = pc.n_residues
    rid = np.asarray(pc.res_id, np.int64)
    poly = (np.asarray(pc.res_flags) & config.R_POLYPEPTIDE) != 0
    mc = np.asarray([n in config.MAINCHAIN_ATOMS for n in pc.atom_name], bool) if pc.n_atoms else np.zeros(0, bool)
    rows = {'': np.ones(pc.n_atoms, bool), 'mc_': mc & poly[rid], 'sc_': ~mc & poly[rid]}
    isift = atom_isift.astype(np.int64)                     # [n, 4, 15]
    integer = {}
    for part, sel in rows.items():
        acc = np.zeros((nr, 4, 15), np.int64)
        np.add.at(acc, rid[sel], isift[sel])
        for k, cls in enumerate(_CLASS_PREFIX):
            integer[part + 'integer_sift' + ('_' + cls[:-1] if cls else '')] = acc[:, k, :]
    out = {}
    for part in ('', 'mc_', 'sc_'):                         # binary: flatten of the integer ones (I:493-496, 553-561)
        for cls in ('', '_inter_only', '_intra_only', '_water_only'):
            out[part + 'sift' + cls] = (integer[part + 'integer_sift' + cls] != 0).astype(np.int64)
    for part in ('', 'mc_', 'sc_'):
        for cls in ('', '_inter_only', '_intra_only', '_water_only'):
            out[part + 'integer_sift' + cls] = integer[part + 'integer_sift' + cls]
    ring_names = ('ring_ring_inter', 'ring_atom_inter', 'atom_ring_inter', 'mc_atom_ring_inter', 'sc_atom_ring_inter')
    for nme in ring_names:
        out[nme + '_sift'] = (plane_sifts[nme + '_integer_sift'] != 0).astype(np.int64)
    for nme in ring_names:
        out[nme + '_integer_sift'] = plane_sifts[nme + '_integer_sift']
    amide_names = ('amide_ring_inter', 'ring_amide_inter', 'amide_amide_inter')
    for nme in amide_names:
        out[nme + '_sift'] = (plane_sifts[nme + '_integer_sift'] != 0).astype(np.int64)
    for nme in amide_names:
        out[nme + '_integer_sift'] = plane_sifts[nme + '_integer_sift']
    return out


def write_residue_sifts(wd, sid, pc, component_types, residues, table):
    """interactions.py:435-466: one row per residue of selection_plus."""
    lab = Labels(pc, component_types)
    mat = np.concatenate([table[k] for k in table], axis=1)
    poly = ((np.asarray(pc.res_flags) & config.R_POLYPEPTIDE) != 0).tolist()
    fh, w = _writer(os.path.join(wd, sid + '_residue_sifts.csv'))
    with fh:
        w.writerow(residue_sift_header())
        w.writerows([lab.res_macro[r], poly[r]] + mat[r].tolist() for r in np.asarray(residues).tolist())


def potential_counts(pc):
    """atom.potential_hbonds == atom.potential_polars as `_initialize_atom_sift` leaves them (interactions.py:1780-1816):
    acceptors add their lone pairs (``lone_pair_electrons / 2`` — a float — unless the electron count is 0), donors their
    bound hydrogens.  Python numbers, int or float exactly as the reference's arithmetic yields them."""
    T = config.ATOM_TYPE_BIT
    nh = np.diff(pc.h_off).tolist()
    if pc.lone_pair_electrons is None:
        if (pc.type_mask & T['hbond acceptor']).any():
            logging.warning('potential hbond / polar counts: the pack carries no lone_pair_electrons (OpenBabel valence data, I:1804); '
                            'acceptors are counted with 0 lone pairs')
        lone = [0] * pc.n_atoms
    else:
        lone = pc.lone_pair_electrons.tolist()
    out = []
    for m, h, lp in zip(pc.type_mask.tolist(), nh, lone):
        v = 0
        if m & T['hbond acceptor']:
            v = v + (lp / 2 if lp != 0 else lp)
        if m & T['hbond donor']:
            v = v + h
        out.append(v)
    return out


def potential_fsift(pc):
    """atom.potential_fsift (interactions.py:1795-1852) as 10-bit masks (bit k = FEATURE_SIFT[k])."""
    T = config.ATOM_TYPE_BIT
    m = pc.type_mask.astype(np.uint32)
    hal = (pc.flags & config.F_HALOGEN) != 0
    met = (pc.flags & config.F_METAL) != 0
    has = lambda *names: (m & sum(T[n] for n in names)) != 0     # noqa: E731
    hb = has('hbond acceptor', 'hbond donor')
    weak = has('weak hbond acceptor', 'weak hbond donor', 'hbond donor', 'hbond acceptor') | hal
    cols = (hb, weak, has('xbond acceptor', 'xbond donor'), has('pos ionisable', 'neg ionisable'), has('hbond acceptor') | met,
            has('aromatic'), has('hydrophobe'), has('carbonyl oxygen', 'carbonyl carbon'), hb, weak)
    return sum(c.astype(np.uint16) << k for k, c in enumerate(cols)).astype(np.uint16)


def write_polar_matching(wd, sid, pc, component_types, atoms, counts):
    """interactions.py:405-433: '<id>_polarmatch.csv' and '<id>_specific_polarmatch.csv' (no header rows)."""
    lab = Labels(pc, component_types)
    pot = potential_counts(pc)
    c = counts.tolist()      # [hbonds, hbonds_intra, hbonds_inter, hbonds_water, polars, polars_intra, polars_inter, polars_water]
    fa, wa = _writer(os.path.join(wd, sid + '_polarmatch.csv'))
    fs, ws = _writer(os.path.join(wd, sid + '_specific_polarmatch.csv'))
    with fa, fs:
        for i in np.asarray(atoms).tolist():
            wa.writerow([lab.atom_macro(i), pot[i], pot[i], c[i][0], c[i][4]])
            ws.writerow([lab.atom_macro(i), pot[i], pot[i], c[i][2], c[i][1], c[i][3], c[i][6], c[i][5], c[i][7]])
