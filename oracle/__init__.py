"""ORACLE — CPU restatement of the reference's hot path (TEST INFRASTRUCTURE ONLY).

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of bench.py may
import this package, and only as the checker.  The product (arpeggio_amd/) never does.

Parity status: pinned against executed reference code for the whole path (run_arpeggio and everything below it;
tests/golden/make_golden_core.py) — see the header of ref_c.c and tests/golden/README.md; third-party semantics the
reference calls but does not contain (KD-tree membership test and delivery order) are restated as recalled.

``ref_c.c`` is the restatement (plain C, gcc); this module is its ctypes binding.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, '_build', 'liborc.so')
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, 'ref_c.c')
    hdr = os.path.join(_HERE, '..', 'include', 'arpeggio_hip.h')
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
    if force or stale:
        subprocess.run(['make', '-C', _HERE, '-B' if force else '-s'], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Complex(C.Structure):
    _fields_ = [
        ('n', C.c_int64), ('xyz', C.c_void_p), ('vdw', C.c_void_p), ('cov', C.c_void_p),
        ('tmask', C.c_void_p), ('flags', C.c_void_p), ('res_id', C.c_void_p),
        ('nres', C.c_int64), ('res_flags', C.c_void_p), ('res_prev', C.c_void_p), ('res_next', C.c_void_p),
        ('bond_off', C.c_void_p), ('bond_idx', C.c_void_p), ('h_off', C.c_void_p), ('h_xyz', C.c_void_p),
        ('sb_nbr', C.c_void_p), ('in_sel', C.c_void_p), ('in_plus', C.c_void_p),
        ('nring', C.c_int64), ('ring_center', C.c_void_p), ('ring_normal', C.c_void_p), ('ring_res', C.c_void_p),
        ('ring_sel', C.c_void_p), ('ring_plus', C.c_void_p),
        ('namide', C.c_int64), ('amide_center', C.c_void_p), ('amide_normal', C.c_void_p), ('amide_res', C.c_void_p),
        ('amide_sel', C.c_void_p), ('amide_plus', C.c_void_p),
    ]


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_dist_f32.restype = C.c_float
        L.orc_dot_f32.restype = C.c_float
        L.orc_norm_f32.restype = C.c_float
        L.orc_dot_f64.restype = C.c_double
        L.orc_norm_f64.restype = C.c_double
        for f in ('orc_get_angle_f64', 'orc_get_angle_f32', 'orc_get_angle_mixed', 'orc_group_angle_f64',
                  'orc_group_angle_f32n_f64p'):
            getattr(L, f).restype = C.c_double
        L.orc_group_angle_f32.restype = C.c_float
        L.orc_is_hbond_like.restype = C.c_int
        L.orc_is_hbond_like.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_double]
        L.orc_contact_type.restype = C.c_int
        L.orc_pp_class.restype = C.c_int
        L.orc_pp_class.argtypes = [C.c_double, C.c_double]
        for f in ('orc_search_all_brute', 'orc_search_all_grid', 'orc_search_point', 'orc_atom_contacts',
                  'orc_atom_plane', 'orc_plane_plane', 'orc_group_group', 'orc_group_plane'):
            getattr(L, f).restype = C.c_int64
        L.orc_make_selection.restype = C.c_int
        L.orc_pair_contact.restype = C.c_int
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _u8(a, n):
    if a is None:
        return np.ones(n, np.uint8)
    return np.ascontiguousarray(a, dtype=np.uint8)


class OracleComplex:
    """A PackedComplex bound to the C struct, plus selection state."""

    def __init__(self, pc, in_sel=None, in_plus=None):
        self.pc = pc
        n = pc.n_atoms
        self.in_sel = _u8(in_sel, n)
        self.in_plus = _u8(in_plus, n)
        self.ring_sel = np.zeros(pc.n_rings, np.uint8)
        self.ring_plus = np.zeros(pc.n_rings, np.uint8)
        self.amide_sel = np.zeros(pc.n_amides, np.uint8)
        self.amide_plus = np.zeros(pc.n_amides, np.uint8)
        self._bind()

    def _bind(self):
        pc = self.pc
        s = _Complex()
        s.n = pc.n_atoms
        s.xyz, s.vdw, s.cov = _p(pc.xyz), _p(pc.vdw), _p(pc.cov)
        s.tmask, s.flags, s.res_id = _p(pc.type_mask), _p(pc.flags), _p(pc.res_id)
        s.nres = pc.n_residues
        s.res_flags, s.res_prev, s.res_next = _p(pc.res_flags), _p(pc.res_prev), _p(pc.res_next)
        s.bond_off, s.bond_idx = _p(pc.bond_off), _p(pc.bond_idx)
        s.h_off, s.h_xyz, s.sb_nbr = _p(pc.h_off), _p(pc.h_xyz), _p(pc.sb_nbr)
        s.in_sel, s.in_plus = _p(self.in_sel), _p(self.in_plus)
        s.nring = pc.n_rings
        s.ring_center, s.ring_normal, s.ring_res = _p(pc.ring_center), _p(pc.ring_normal), _p(pc.ring_res)
        s.ring_sel, s.ring_plus = _p(self.ring_sel), _p(self.ring_plus)
        s.namide = pc.n_amides
        s.amide_center, s.amide_normal, s.amide_res = _p(pc.amide_center), _p(pc.amide_normal), _p(pc.amide_res)
        s.amide_sel, s.amide_plus = _p(self.amide_sel), _p(self.amide_plus)
        self.s = s

    # I:1384-1451
    def make_selection(self, in_sel=None, radius=6.0, use_grid=True):
        n = self.pc.n_atoms
        self.in_sel = _u8(in_sel, n)
        plus = np.zeros(n, np.uint8)
        self._bind()
        rc = lib().orc_make_selection(C.byref(self.s), _p(self.in_sel), C.c_double(radius), int(use_grid),
                                      _p(plus), _p(self.ring_sel), _p(self.ring_plus),
                                      _p(self.amide_sel), _p(self.amide_plus))
        if rc != 0:
            raise MemoryError('orc_make_selection')
        self.in_plus = plus
        self._bind()
        return plus

    # I:693-936
    def atom_contacts(self, cutoff=5.0, vdw_comp=0.1, include_sequence_adjacent=False, use_grid=True, cap_hint=None):
        """``cap_hint``: a capacity known to be enough (e.g. the count of an earlier call on the same structure): the
        search and the per-pair evaluation then run ONCE instead of once to count and once to fill (bench.py's baseline)."""
        L = lib()
        stats = np.zeros(8, np.int64)
        err = C.c_int(0)
        args = (C.byref(self.s), C.c_double(cutoff), C.c_double(vdw_comp), int(include_sequence_adjacent), int(use_grid))
        if cap_hint is None:
            cnt = L.orc_atom_contacts(*args, C.c_int64(0), None, None, None, None, None, _p(stats), C.byref(err))
        else:
            cnt = int(cap_hint)
        cap = max(int(cnt), 1)
        oi, oj = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        od, osf, oc = np.zeros(cap, np.float32), np.zeros(cap, np.uint16), np.zeros(cap, np.uint8)
        cnt = L.orc_atom_contacts(*args, C.c_int64(cap), _p(oi), _p(oj), _p(od), _p(osf), _p(oc), _p(stats), C.byref(err))
        if int(cnt) > cap:       # a cap_hint that was too small: once more with the count the call reported (never a short list)
            cap = int(cnt)
            oi, oj = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
            od, osf, oc = np.zeros(cap, np.float32), np.zeros(cap, np.uint16), np.zeros(cap, np.uint8)
            cnt = L.orc_atom_contacts(*args, C.c_int64(cap), _p(oi), _p(oj), _p(od), _p(osf), _p(oc), _p(stats), C.byref(err))
        k = int(cnt)
        out = dict(i=oi[:k], j=oj[:k], dist=od[:k], sift=osf[:k], ctype=oc[:k], stats=stats, err=err.value)
        return sort_pairs(out)

    def pair_contact(self, b, e, vdw_comp=0.1, include_sequence_adjacent=False):
        d, s, ct, err = C.c_float(), C.c_uint16(), C.c_uint8(), C.c_int(0)
        ok = lib().orc_pair_contact(C.byref(self.s), int(b), int(e), C.c_double(vdw_comp),
                                    int(include_sequence_adjacent), C.byref(d), C.byref(s), C.byref(ct), C.byref(err))
        return (bool(ok), np.float32(d.value), int(s.value), int(ct.value), err.value)

    # I:947-1062
    def atom_plane(self):
        L = lib()
        cnt = L.orc_atom_plane(C.byref(self.s), C.c_int64(0), None, None, None, None, None, None)
        cap = max(int(cnt), 1)
        oa, orr = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        od, ot = np.zeros(cap, np.float64), np.zeros(cap, np.float64)
        om, oc = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8)
        cnt = L.orc_atom_plane(C.byref(self.s), C.c_int64(cap), _p(oa), _p(orr), _p(od), _p(ot), _p(om), _p(oc))
        k = int(cnt)
        out = dict(atom=oa[:k], ring=orr[:k], dist=od[:k], theta=ot[:k], mask=om[:k], ctype=oc[:k])
        o = np.lexsort((out['atom'], out['ring']))
        return {kk: v[o] for kk, v in out.items()}

    # I:1064-1194 (records in the reference's creation order)
    def plane_plane(self):
        L = lib()
        R = self.pc.n_rings
        cap = 64
        while True:
            ob, oe = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
            od, odi = np.zeros(cap, np.float64), np.zeros(cap, np.float64)
            t1, t2 = np.zeros(cap, np.float64), np.zeros(cap, np.float64)
            y1, y2, oc = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8), np.zeros(cap, np.uint8)
            cnt = int(L.orc_plane_plane(C.byref(self.s), C.c_int64(cap), _p(ob), _p(oe), _p(od), _p(odi), _p(t1), _p(t2),
                                        _p(y1), _p(y2), _p(oc)))
            if cnt <= cap:
                break
            cap = cnt  # dedupe needs every record stored: re-run with room for all
        k = cnt
        return dict(bgn=ob[:k], end=oe[:k], dist=od[:k], dihedral=odi[:k], theta_bgn=t1[:k], theta_end=t2[:k],
                    type1=y1[:k], type2=y2[:k], ctype=oc[:k])

    # I:1217-1300
    def group_group(self):
        L = lib()
        cnt = L.orc_group_group(C.byref(self.s), C.c_int64(0), None, None, None, None, None, None)
        cap = max(int(cnt), 1)
        ob, oe = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        od, odi, ot = np.zeros(cap, np.float32), np.zeros(cap, np.float32), np.zeros(cap, np.float32)
        oc = np.zeros(cap, np.uint8)
        cnt = L.orc_group_group(C.byref(self.s), C.c_int64(cap), _p(ob), _p(oe), _p(od), _p(odi), _p(ot), _p(oc))
        k = int(cnt)
        return dict(bgn=ob[:k], end=oe[:k], dist=od[:k], dihedral=odi[:k], theta=ot[:k], ctype=oc[:k])

    # I:1302-1382
    def group_plane(self):
        L = lib()
        cnt = L.orc_group_plane(C.byref(self.s), C.c_int64(0), None, None, None, None, None, None)
        cap = max(int(cnt), 1)
        oa, orr = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        od, odi, ot = np.zeros(cap, np.float64), np.zeros(cap, np.float64), np.zeros(cap, np.float64)
        oc = np.zeros(cap, np.uint8)
        cnt = L.orc_group_plane(C.byref(self.s), C.c_int64(cap), _p(oa), _p(orr), _p(od), _p(odi), _p(ot), _p(oc))
        k = int(cnt)
        return dict(amide=oa[:k], ring=orr[:k], dist=od[:k], dihedral=odi[:k], theta=ot[:k], ctype=oc[:k])


def atom_accumulators(n, contacts):
    """Per-atom sift masks (n,4) and hbond/polar counters (n,8) from a contact list (I:821-852, 923-934)."""
    ci = np.ascontiguousarray(contacts['i'], np.int32)
    cj = np.ascontiguousarray(contacts['j'], np.int32)
    cs = np.ascontiguousarray(contacts['sift'], np.uint16)
    cc = np.ascontiguousarray(contacts['ctype'], np.uint8)
    sift = np.zeros((max(n, 1), 4), np.uint16)
    cnt = np.zeros((max(n, 1), 8), np.int32)
    lib().orc_atom_accumulators(C.c_int64(n), C.c_int64(len(ci)), _p(ci), _p(cj), _p(cs), _p(cc), _p(sift), _p(cnt))
    return dict(sift=sift[:n], counts=cnt[:n])


def atom_integer_sifts(n, contacts):
    """update_atom_integer_sift (U:224-242) walked over the contact list in the order given: uint8 [n, 4, 15]."""
    ci = np.ascontiguousarray(contacts['i'], np.int32)
    cj = np.ascontiguousarray(contacts['j'], np.int32)
    cs = np.ascontiguousarray(contacts['sift'], np.uint16)
    cc = np.ascontiguousarray(contacts['ctype'], np.uint8)
    out = np.zeros((max(n, 1), 4, 15), np.uint8)
    lib().orc_atom_integer_sifts(C.c_int64(n), C.c_int64(len(ci)), _p(ci), _p(cj), _p(cs), _p(cc), _p(out))
    return out[:n]


def sort_pairs(out, ki='i', kj='j'):
    """Canonical order: ascending (i, j)."""
    o = np.lexsort((out[kj], out[ki]))
    return {k: (v[o] if isinstance(v, np.ndarray) and v.shape[:1] == o.shape else v) for k, v in out.items()}


def search_all(xyz, radius, active=None, grid=True):
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    n = xyz.shape[0]
    act = None if active is None else np.ascontiguousarray(active, np.uint8)
    L = lib()
    cand = C.c_int64(0)
    if grid:
        cnt = L.orc_search_all_grid(C.c_int64(n), _p(xyz), _p(act), C.c_double(radius), C.c_int64(0), None, None, C.byref(cand))
    else:
        cnt = L.orc_search_all_brute(C.c_int64(n), _p(xyz), _p(act), C.c_double(radius), C.c_int64(0), None, None)
    cap = max(int(cnt), 1)
    oi, oj = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    if grid:
        L.orc_search_all_grid(C.c_int64(n), _p(xyz), _p(act), C.c_double(radius), C.c_int64(cap), _p(oi), _p(oj), C.byref(cand))
    else:
        L.orc_search_all_brute(C.c_int64(n), _p(xyz), _p(act), C.c_double(radius), C.c_int64(cap), _p(oi), _p(oj))
    k = int(cnt)
    o = np.lexsort((oj[:k], oi[:k]))
    return oi[:k][o], oj[:k][o], int(cand.value)


def search_point(xyz, center, radius, active=None):
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    n = xyz.shape[0]
    act = None if active is None else np.ascontiguousarray(active, np.uint8)
    ctr = np.ascontiguousarray(center, np.float64)
    out = np.zeros(n if n else 1, np.int32)
    cnt = lib().orc_search_point(C.c_int64(n), _p(xyz), _p(act), _p(ctr), C.c_double(radius), C.c_int64(n), _p(out))
    return out[:int(cnt)]


# ---- timing-only multi-core variant of one whole-structure pass (bench.py's extra CPU figure) -----------------------
_omp_lib = None


def pass_openmp(oc, cutoff=5.0, comp=0.1, include_sequence_adjacent=False, threads=0):
    """One run_arpeggio pass of the C restatement with its cell loops spread over OpenMP threads (``threads`` = 0: all
    cores): same searches, same per-pair evaluation, contacts kept in per-thread buffers.  Returns dict(candidates_6A,
    pairs_6A, candidates, contacts, sift_checksum).  Not a checker: nothing is compared against it except its counts."""
    global _omp_lib
    if _omp_lib is None:
        build()
        _omp_lib = C.CDLL(os.path.join(_HERE, '_build', 'liborc_omp.so'))
        _omp_lib.orc_pass_openmp.restype = C.c_int
    st = np.zeros(5, np.int64)
    rc = _omp_lib.orc_pass_openmp(C.byref(oc.s), C.c_double(cutoff), C.c_double(comp), C.c_int(int(include_sequence_adjacent)),
                                  C.c_int(int(threads)), _p(st))
    if rc != 0:
        raise MemoryError('orc_pass_openmp failed')
    return dict(candidates_6A=int(st[0]), pairs_6A=int(st[1]), candidates=int(st[2]), contacts=int(st[3]), sift_checksum=int(st[4]))
