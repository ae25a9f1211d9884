"""ref_py — second, independent restatement of the per-pair loop in the reference's own idiom: Python control flow
and *real NumPy calls* (np.linalg.norm, np.dot, np.arccos on float32 / float64 arrays), so that it inherits whatever
NumPy + BLAS actually do on the machine it runs on.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Purpose: (1) cross-check oracle/ref_c.c, whose arithmetic is an explicit model of these NumPy calls (float32 dot with
float64 accumulation, FMA-chained float64 dot, NEP-50 comparisons); (2) the interpreter-bound "B1" baseline of
BASELINE.md.  Only small inputs: it is as slow as the reference.

Follows arpeggio/core/interactions.py:693-936 and arpeggio/core/utils.py:73-179, 696-745 on a PackedComplex.
"""
from __future__ import annotations

import numpy as np

from arpeggio_amd.core import config

T = config.ATOM_TYPE_BIT
S = {name: 1 << k for k, name in enumerate(config.SIFT_NAMES)}
CT = {name: k for k, name in enumerate(config.CONTACT_TYPE_NAMES)}
TH = config.CONTACT_TYPES


def get_angle(pa, pb, pc):
    """utils.py:696-745 with NumPy scalars: the dtype follows the operands exactly as in the reference."""
    v1 = pa - pb
    v2 = pc - pb
    with np.errstate(all='ignore'):
        m1 = np.sqrt(v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2])
        n1 = np.array([v1[0] / m1, v1[1] / m1, v1[2] / m1])
        m2 = np.sqrt(v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2])
        n2 = np.array([v2[0] / m2, v2[1] / m2, v2[2] / m2])
        ang = np.arccos(n1[0] * n2[0] + n1[1] * n2[1] + n1[2] * n2[2])
    return np.pi if np.isnan(ang) else ang


class RefPy:
    def __init__(self, pc, in_sel=None, in_plus=None):
        self.pc = pc
        n = pc.n_atoms
        self.sel = np.ones(n, bool) if in_sel is None else np.asarray(in_sel, bool)
        self.plus = np.ones(n, bool) if in_plus is None else np.asarray(in_plus, bool)
        self.h = [pc.h_xyz[pc.h_off[i]:pc.h_off[i + 1]] for i in range(n)]                  # float64 rows
        self.bonded = [set(pc.bond_idx[pc.bond_off[i]:pc.bond_off[i + 1]].tolist()) for i in range(n)]

    # utils.py:73-116
    def _hbond(self, donor, acceptor, comp, angle_min):
        pc = self.pc
        for h in self.h[donor]:
            if np.linalg.norm(h - pc.xyz[acceptor]) <= config.VDW_RADII['H'] + float(pc.vdw[acceptor]) + comp:
                if get_angle(pc.xyz[donor], h, pc.xyz[acceptor]) >= angle_min:
                    return True
        return False

    # utils.py:119-155
    def _halogen_weak(self, donor, halogen, comp):
        pc = self.pc
        nbr = int(pc.sb_nbr[halogen])
        if nbr < 0:
            return False
        for h in self.h[donor]:
            if np.linalg.norm(pc.xyz[halogen] - h) <= config.VDW_RADII['H'] + float(pc.vdw[halogen]) + comp:
                a = get_angle(pc.xyz[nbr], pc.xyz[halogen], h)
                if TH['weak hbond']['cx angle min rad'] <= a <= TH['weak hbond']['cx angle max rad']:
                    return True
        return False

    # utils.py:158-179
    def _xbond(self, donor, acceptor):
        pc = self.pc
        nbr = int(pc.sb_nbr[donor])
        if nbr < 0:
            raise AttributeError("'NoneType' object has no attribute 'GetId'")
        return get_angle(pc.xyz[nbr], pc.xyz[donor], pc.xyz[acceptor]) >= TH['xbond']['angle theta 1 rad']

    # interactions.py:643-691
    def _ctype(self, b, e):
        pc = self.pc
        bs, es = self.sel[b], self.sel[e]
        bw, ew = bool(pc.flags[b] & config.F_WATER), bool(pc.flags[e] & config.F_WATER)
        ct = None
        if not bs and not es:
            ct = 'INTRA_NON_SELECTION'
        if bs and es:
            ct = 'INTRA_SELECTION'
        if (bs and not es) or (es and not bs):
            ct = 'INTER'
        if (bs and ew) or (es and bw):
            ct = 'SELECTION_WATER'
        if (not bs and ew) or (not es and bw):
            ct = 'NON_SELECTION_WATER'
        if bw and ew:
            ct = 'WATER_WATER'
        return ct

    def pair(self, b, e, comp=0.1, include_sequence_adjacent=False):
        """One iteration of interactions.py:707-936 for the ordered pair (bgn=b, end=e); None when it `continue`s."""
        pc = self.pc
        fb, fe = int(pc.flags[b]), int(pc.flags[e])
        tb, te = int(pc.type_mask[b]), int(pc.type_mask[e])
        if (fb | fe) & config.F_HYDROGEN:
            return None
        ct = self._ctype(b, e)
        sum_cov = float(pc.cov[b]) + float(pc.cov[e])
        sum_vdw = float(pc.vdw[b]) + float(pc.vdw[e])
        rb, re = int(pc.res_id[b]), int(pc.res_id[e])
        if rb == re:
            return None
        if not include_sequence_adjacent and pc.res_flags[re] & config.R_POLYPEPTIDE:
            if pc.res_flags[rb] & config.R_HAS_SEQ and pc.res_flags[re] & config.R_HAS_SEQ:
                if pc.res_next[rb] == re or pc.res_prev[rb] == re or pc.res_next[re] == rb or pc.res_prev[re] == rb:
                    return None
        d = np.linalg.norm(pc.xyz[b] - pc.xyz[e])                 # float32 scalar
        s = 0
        if e in self.bonded[b]:
            s |= S['covalent']
        elif d < sum_cov:                                          # float32 vs Python float: NEP 50 -> float32 compare
            s |= S['clash']
        elif d < sum_vdw:
            s |= S['vdw_clash']
        elif d <= sum_vdw + comp:
            s |= S['vdw']
        else:
            s |= S['proximal']
        if d <= TH['metal']['distance']:
            if tb & T['hbond acceptor'] and fe & config.F_METAL:
                s |= S['metal_complex']
            elif te & T['hbond acceptor'] and fb & config.F_METAL:
                s |= S['metal_complex']
        if not s & S['clash'] and d <= config.CONTACT_TYPES_DIST_MAX:
            bw, ew = fb & config.F_WATER, fe & config.F_WATER
            if bw and d <= sum_vdw + comp:
                if te & (T['hbond acceptor'] | T['hbond donor']):
                    s |= S['hbond'] | S['polar']
            elif ew and d <= sum_vdw + comp:
                if tb & (T['hbond acceptor'] | T['hbond donor']):
                    s |= S['hbond'] | S['polar']
            else:
                if tb & T['hbond donor'] and te & T['hbond acceptor']:
                    if self._hbond(b, e, comp, TH['hbond']['angle rad']):
                        s |= S['hbond']
                    if d <= TH['hbond']['polar distance']:
                        s |= S['polar']
                elif te & T['hbond donor'] and tb & T['hbond acceptor']:
                    if self._hbond(e, b, comp, TH['hbond']['angle rad']):
                        s |= S['hbond']
                    if d <= TH['hbond']['polar distance']:
                        s |= S['polar']
            weak = False
            wp = d <= TH['weak hbond']['weak polar distance']
            if tb & T['hbond acceptor'] and te & T['weak hbond donor']:
                weak = self._hbond(e, b, comp, TH['weak hbond']['angle rad'])
                s |= S['weak_polar'] if wp else 0
            if tb & T['weak hbond donor'] and te & T['hbond acceptor']:
                weak = self._hbond(b, e, comp, TH['weak hbond']['angle rad'])
                s |= S['weak_polar'] if wp else 0
            if tb & T['weak hbond acceptor'] and fb & config.F_HALOGEN and te & (T['hbond donor'] | T['weak hbond donor']):
                weak = self._halogen_weak(e, b, comp)
                s |= S['weak_polar'] if wp else 0
            if te & T['weak hbond acceptor'] and fe & config.F_HALOGEN and tb & (T['hbond donor'] | T['weak hbond donor']):
                weak = self._halogen_weak(b, e, comp)
                s |= S['weak_polar'] if wp else 0
            if weak:
                s |= S['weak_hbond']
            if d <= sum_vdw + comp:
                if tb & T['xbond donor'] and te & T['xbond acceptor']:
                    s |= S['xbond'] if self._xbond(b, e) else 0
                elif te & T['xbond donor'] and tb & T['xbond acceptor']:
                    s |= S['xbond'] if self._xbond(e, b) else 0
            if d <= TH['ionic']['distance']:
                if (tb & T['pos ionisable'] and te & T['neg ionisable']) or (tb & T['neg ionisable'] and te & T['pos ionisable']):
                    s |= S['ionic']
            if d <= TH['carbonyl']['distance']:
                if (tb & T['carbonyl oxygen'] and te & T['carbonyl carbon']) or (te & T['carbonyl oxygen'] and tb & T['carbonyl carbon']):
                    s |= S['carbonyl']
            if tb & te & T['aromatic'] and d <= TH['aromatic']['distance']:
                s |= S['aromatic']
            if tb & te & T['hydrophobe'] and d <= TH['hydrophobic']['distance']:
                s |= S['hydrophobic']
        return d, s, CT[ct]

    def atom_contacts(self, cutoff=5.0, comp=0.1, include_sequence_adjacent=False):
        """Brute-force search_all over selection_plus (float64 d^2 <= r^2), canonical orientation, sorted by (i, j)."""
        pc = self.pc
        idx = np.nonzero(self.plus)[0]
        x = pc.xyz[idx].astype(np.float64)
        out = []
        for a in range(len(idx)):
            dv = x[a + 1:] - x[a]
            d2 = dv[:, 0] * dv[:, 0] + dv[:, 1] * dv[:, 1]
            d2 = d2 + dv[:, 2] * dv[:, 2]
            for k in np.nonzero(d2 <= cutoff * cutoff)[0]:
                b, e = int(idx[a]), int(idx[a + 1 + k])
                r = self.pair(b, e, comp, include_sequence_adjacent)
                if r is not None:
                    out.append((b, e) + r)
        out.sort(key=lambda t: (t[0], t[1]))
        return dict(i=np.array([t[0] for t in out], np.int32), j=np.array([t[1] for t in out], np.int32),
                    dist=np.array([t[2] for t in out], np.float32), sift=np.array([t[3] for t in out], np.uint16),
                    ctype=np.array([t[4] for t in out], np.uint8))


# ---- geometric part of initialize() -----------------------------------------------------------
def ring_geometry(xyz, ring_atoms):
    """interactions.py:1697-1733 -> OBRing::findCenterAndNormal (OpenBabel ring.cpp, third party, restated from its
    published source; PARITY UNPINNED — OpenBabel is not installed here): centre = sum of the atom vectors times
    1/n, normal = sum of cross(v_j - centre, v_j+1 - centre) times 1/n, normalised unless its length is < 2e-6
    (vector3::normalize / IsNearZero; operator/= multiplies by the reciprocal).  float64 on the float32 coordinates."""
    x = np.asarray(xyz, np.float32).reshape(-1, 3).astype(np.float64)
    ctr, nrm = np.zeros((len(ring_atoms), 3)), np.zeros((len(ring_atoms), 3))
    for r, atoms in enumerate(ring_atoms):
        na = len(atoms)
        c = np.zeros(3)
        for a in atoms:
            c = c + x[a]
        inv = 1.0 / float(na)
        c = c * inv
        n = np.zeros(3)
        for j in range(na):
            v1 = x[atoms[j]] - c
            v2 = x[atoms[0 if j + 1 == na else j + 1]] - c
            n = n + np.array([v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0]])
        n = n * inv
        length = np.sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2])
        if not abs(length) < 2e-6:
            n = n * (1.0 / length)
        ctr[r], nrm[r] = c, n
    return ctr, nrm


def amide_geometry(xyz, amide_atoms):
    """interactions.py:1566-1580 with the reference's own NumPy calls: bond centroid of C-N (float32) and the plane
    normal from np.linalg.svd of the centred C, O, N coordinates (float32; the sign is whatever LAPACK returns)."""
    x = np.asarray(xyz, np.float32).reshape(-1, 3)
    at = np.asarray(amide_atoms).reshape(-1, 4)
    ctr, nrm = np.zeros((len(at), 3), np.float32), np.zeros((len(at), 3), np.float32)
    for k, (n_, c_, o_, _) in enumerate(at):
        con = np.array([x[c_], x[o_], x[n_]])                # C-O-N
        cn = np.array([x[c_], x[n_]])                        # C-N
        amide_centroid = con.sum(0) / float(len(con))
        bond_centroid = cn.sum(0) / float(len(cn))
        cog = con - amide_centroid
        u, s_, vh = np.linalg.svd(cog)
        v = vh.conj().transpose()
        a, b, c = v[:, -1]
        ctr[k], nrm[k] = bond_centroid, np.array([a, b, c])
    return ctr, nrm


def ring_residues(xyz, res_id, ring_center):
    """interactions.py:1453-1492: atoms within 3.0 A of the ring centre (Bio.PDB.kdtrees: float64 sum of squares <= r*r),
    nearest by np.linalg.norm(atom.coord - centre) with a strict '<' (first of equals wins; brute force in packed atom
    order here, KD-tree order in the reference), its residue; -1 / -1.0 when no atom qualifies."""
    x = np.asarray(xyz, np.float32).reshape(-1, 3)
    ctr = np.asarray(ring_center, np.float64).reshape(-1, 3)
    res, dist = np.full(len(ctr), -1, np.int32), np.full(len(ctr), -1.0)
    xd = x.astype(np.float64)
    for r, c in enumerate(ctr):
        d = xd - c
        near = np.nonzero(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2] <= 9.0)[0]
        closest = (None, None)
        for a in near:
            distance = np.linalg.norm(x[a] - c)
            if closest[1] is None or distance < closest[1]:
                closest = (a, distance)
        if closest[0] is not None:
            res[r], dist[r] = res_id[closest[0]], closest[1]
    return res, dist
