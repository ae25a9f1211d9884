/*
 * ref_c.c — CPU restatement (ORACLE) of arpeggio 1.4.4's contact-detection hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (arpeggio_amd/) may
 * import, link or call this file; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker.
 *
 * Parity status: PINNED against executed reference code for everything the reference itself computes on the
 * path.  The reference cannot be imported as a package here (BioPython / OpenBabel / gemmi are absent; it ships no
 * tests or fixtures), so tests/golden/make_golden.py and make_golden_core.py compile its own function and method
 * bodies with `ast` from the files where they lie under /root/reference and run them on data holders (stored
 * attributes, neighbour lists, bond orders — no arithmetic of the path): run_arpeggio, _make_selection,
 * _calculate_atom_contacts, __get_contact_type, the atom-plane / plane-plane / group-group / group-plane loops,
 * is_hbond, is_weak_hbond, is_halogen_weak_hbond, is_xbond, get_single_bond_neighbour, get_angle, group_angle,
 * group_group_angle and the three update_atom_* accumulators; 136 structures in three delivery orders, 240 k
 * atom-atom records, all compared bit for bit with this file (tests/test_golden_core.py, tests/test_oracle_golden.py)
 * and — directly, without this file in between — with the HIP kernels.
 * What stays "recalled" is third-party semantics the reference calls but does not contain: the KD-tree's
 * membership test (Bio.PDB.kdtrees: float64 dx*dx + dy*dy + dz*dz <= r*r, inclusive) and its delivery order / pair
 * orientation (results are defined on the canonical one, bgn = lower packed index), OpenBabel's iteration order
 * of bonds.
 *
 * Citations: I: = arpeggio/core/interactions.py, U: = arpeggio/core/utils.py,
 * C: = arpeggio/core/config.py of the reference tree.
 *
 * Arithmetic model (NumPy 2.2.6 + bundled OpenBLAS, probed):
 *   - np.linalg.norm / np.dot on float32 3-vectors: float32 products, float64
 *     accumulation, one rounding to float32, float32 sqrt.
 *   - np.linalg.norm / np.dot on float64 3-vectors: FMA chain
 *     fma(z,z', fma(y,y', x*x')), float64 sqrt.
 *   - NumPy scalar expressions (U:712-745): one rounding per operation in the
 *     operand dtype; float32 op Python-float -> float32 (NEP 50).
 * Build with -ffp-contract=off so that the only fused operations are the
 * explicit fma() calls below.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/arpeggio_hip.h"

#define ORC_PI 3.141592653589793 /* np.pi */

/* config.CONTACT_TYPES (C:592-660) and friends */
#define K_DIST_MAX 4.5       /* C:590 CONTACT_TYPES_DIST_MAX */
#define K_HBOND_POLAR 3.5    /* C:597 */
#define K_HBOND_ANGLE 1.57   /* C:598 */
#define K_WEAK_POLAR 3.5     /* C:605 */
#define K_WEAK_ANGLE 2.27    /* C:606 */
#define K_CX_MIN 0.52        /* C:608 */
#define K_CX_MAX 2.62        /* C:610 */
#define K_AROMATIC 4.0       /* C:616 */
#define K_CENTROID 6.0       /* C:617 */
#define K_ATOM_AROMATIC 4.5  /* C:618 */
#define K_MET_SULPHUR 6.0    /* C:619 */
#define K_AMIDE_CENTROID 6.0 /* C:624 */
#define K_XBOND_THETA1 2.09  /* C:632 */
#define K_IONIC 4.0          /* C:643 */
#define K_HYDROPHOBIC 4.5    /* C:648 */
#define K_CARBONYL 3.6       /* C:653 */
#define K_METAL 2.8          /* C:658 */
#define K_VDW_H 1.2          /* C:23-25 VDW_RADII['H'] */

typedef struct {
    int64_t n;
    const float* xyz;
    const double* vdw;
    const double* cov;
    const uint16_t* tmask;
    const uint16_t* flags;
    const int32_t* res_id;
    int64_t nres;
    const uint8_t* res_flags;
    const int32_t* res_prev;
    const int32_t* res_next;
    const int32_t* bond_off;
    const int32_t* bond_idx;
    const int32_t* h_off;
    const double* h_xyz;
    const int32_t* sb_nbr;
    const uint8_t* in_sel;
    const uint8_t* in_plus;
    int64_t nring;
    const double* ring_center;
    const double* ring_normal;
    const int32_t* ring_res;
    const uint8_t* ring_sel;
    const uint8_t* ring_plus;
    int64_t namide;
    const float* amide_center;
    const float* amide_normal;
    const int32_t* amide_res;
    const uint8_t* amide_sel;
    const uint8_t* amide_plus;
} orc_complex;

/* ------------------------------------------------------------------------- */
/* elementary arithmetic                                                      */
/* ------------------------------------------------------------------------- */

/* np.dot(a, b) for float32[3] (OpenBLAS sdot tail loop). */
float orc_dot_f32(const float* a, const float* b) {
    double acc = 0.0;
    for (int i = 0; i < 3; ++i) {
        float p = a[i] * b[i];
        acc += (double)p;
    }
    return (float)acc;
}

/* np.dot(a, b) for float64[3] (OpenBLAS ddot tail loop, FMA-contracted). */
double orc_dot_f64(const double* a, const double* b) {
    double acc = a[0] * b[0];
    acc = fma(a[1], b[1], acc);
    acc = fma(a[2], b[2], acc);
    return acc;
}

float orc_norm_f32(const float* v) { return sqrtf(orc_dot_f32(v, v)); }
double orc_norm_f64(const double* v) { return sqrt(orc_dot_f64(v, v)); }

/* I:745  np.linalg.norm(atom_bgn.coord - atom_end.coord), float32 */
float orc_dist_f32(const float* a, const float* b) {
    float d[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
    return orc_norm_f32(d);
}

/* kd-tree membership test of Bio.PDB.kdtrees (float64 d^2 <= r^2, inclusive). */
static inline int within_f64(const float* a, const float* b, double r2) {
    double dx = (double)a[0] - (double)b[0];
    double dy = (double)a[1] - (double)b[1];
    double dz = (double)a[2] - (double)b[2];
    double r = dx * dx;
    r += dy * dy;
    r += dz * dz;
    return r <= r2;
}

/* U:696-745 get_angle with all three points float64 after promotion
 * (any point float64 in BOTH difference vectors -> float64 arithmetic). */
double orc_get_angle_f64(const double* a, const double* b, const double* c) {
    double v1[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
    double v2[3] = {c[0] - b[0], c[1] - b[1], c[2] - b[2]};
    double m1 = sqrt(v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2]);
    double n1[3] = {v1[0] / m1, v1[1] / m1, v1[2] / m1};
    double m2 = sqrt(v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2]);
    double n2[3] = {v2[0] / m2, v2[1] / m2, v2[2] / m2};
    double res = n1[0] * n2[0] + n1[1] * n2[1] + n1[2] * n2[2];
    double ang = acos(res);
    if (isnan(ang)) ang = ORC_PI; /* U:741-743 */
    return ang;
}

/* U:696-745 with three float32 points (is_xbond, U:174): float32 throughout.
 * Returns double so that the NaN -> np.pi substitution keeps Python-float pi. */
double orc_get_angle_f32(const float* a, const float* b, const float* c) {
    float v1[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
    float v2[3] = {c[0] - b[0], c[1] - b[1], c[2] - b[2]};
    float m1 = sqrtf(v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2]);
    float n1[3] = {v1[0] / m1, v1[1] / m1, v1[2] / m1};
    float m2 = sqrtf(v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2]);
    float n2[3] = {v2[0] / m2, v2[1] / m2, v2[2] / m2};
    float res = n1[0] * n2[0] + n1[1] * n2[1] + n1[2] * n2[2];
    float ang = acosf(res);
    if (isnan(ang)) return ORC_PI;
    return (double)ang;
}

/* U:151 get_angle(nbr.coord f32, halogen.coord f32, hydrogen_coord f64):
 * v1 stays float32, v2 is float64, the dot product is float64. */
double orc_get_angle_mixed(const float* a, const float* b, const double* c) {
    float v1[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
    double v2[3] = {c[0] - (double)b[0], c[1] - (double)b[1], c[2] - (double)b[2]};
    float m1 = sqrtf(v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2]);
    float n1[3] = {v1[0] / m1, v1[1] / m1, v1[2] / m1};
    double m2 = sqrt(v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2]);
    double n2[3] = {v2[0] / m2, v2[1] / m2, v2[2] / m2};
    double res = (double)n1[0] * n2[0] + (double)n1[1] * n2[1] + (double)n1[2] * n2[2];
    double ang = acos(res);
    if (isnan(ang)) ang = ORC_PI;
    return ang;
}

/* degrees + signed folding of U:656-660 / U:689-693, float64 */
static double fold_deg_f64(double rad) {
    if (rad > ORC_PI / 2) rad = rad - ORC_PI;
    return rad * 180 / ORC_PI;
}
/* same in float32 (Python floats are weak: cast to float32, NEP 50) */
static float fold_deg_f32(float rad) {
    if (rad > (float)(ORC_PI / 2)) rad = rad - (float)ORC_PI;
    float t = rad * 180.0f;
    return t / (float)ORC_PI;
}

/* abs(group_angle(group, point, True, True)) U:638-660, float64 normal & point */
double orc_group_angle_f64(const double* normal, const double* point) {
    double c = orc_dot_f64(normal, point) / (orc_norm_f64(normal) * orc_norm_f64(point));
    return fabs(fold_deg_f64(acos(c)));
}
/* float32 normal & point (amide-amide) */
float orc_group_angle_f32(const float* normal, const float* point) {
    float c = orc_dot_f32(normal, point) / (orc_norm_f32(normal) * orc_norm_f32(point));
    return fabsf(fold_deg_f32(acosf(c)));
}
/* float32 normal, float64 point/normal (amide-ring): dot promotes to float64,
 * norm(normal) stays float32 and is promoted in the product. */
double orc_group_angle_f32n_f64p(const float* normal, const double* point) {
    double nd[3] = {(double)normal[0], (double)normal[1], (double)normal[2]};
    double c = orc_dot_f64(nd, point) / ((double)orc_norm_f32(normal) * orc_norm_f64(point));
    return fabs(fold_deg_f64(acos(c)));
}

/* ------------------------------------------------------------------------- */
/* U:73-179 predicates                                                        */
/* ------------------------------------------------------------------------- */

static inline void f32_to_f64(const float* s, double* d) {
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
}

/* U:73-93 / U:96-116: is_hbond (angle_min 1.57) and is_weak_hbond (2.27) */
int orc_is_hbond_like(const float* donor_xyz, const double* h_xyz, int nh,
                      const float* acc_xyz, double acc_vdw, double comp, double angle_min) {
    double d[3], a[3];
    f32_to_f64(donor_xyz, d);
    f32_to_f64(acc_xyz, a);
    for (int k = 0; k < nh; ++k) {
        const double* h = h_xyz + 3 * k;
        double v[3] = {h[0] - a[0], h[1] - a[1], h[2] - a[2]};
        double h_dist = orc_norm_f64(v);
        if (h_dist <= K_VDW_H + acc_vdw + comp) {
            if (orc_get_angle_f64(d, h, a) >= angle_min) return 1;
        }
    }
    return 0;
}

/* U:119-155 */
static int is_halogen_weak_hbond(const orc_complex* c, int donor, int halogen, double comp) {
    int nbr = c->sb_nbr[halogen];
    if (nbr < 0) return 0; /* U:139-141 */
    const float* hal = c->xyz + 3 * (int64_t)halogen;
    const float* nb = c->xyz + 3 * (int64_t)nbr;
    double hd[3];
    f32_to_f64(hal, hd);
    for (int k = c->h_off[donor]; k < c->h_off[donor + 1]; ++k) {
        const double* h = c->h_xyz + 3 * (int64_t)k;
        double v[3] = {hd[0] - h[0], hd[1] - h[1], hd[2] - h[2]};
        double h_dist = orc_norm_f64(v);
        if (h_dist <= K_VDW_H + c->vdw[halogen] + comp) {
            double ang = orc_get_angle_mixed(nb, hal, h);
            if (K_CX_MIN <= ang && ang <= K_CX_MAX) return 1;
        }
    }
    return 0;
}

/* U:158-179; *err set when the reference would raise AttributeError (U:173) */
static int is_xbond(const orc_complex* c, int donor, int acceptor, int* err) {
    int nbr = c->sb_nbr[donor];
    if (nbr < 0) { *err = ARP_E_XBOND_NBR; return 0; }
    double theta = orc_get_angle_f32(c->xyz + 3 * (int64_t)nbr, c->xyz + 3 * (int64_t)donor,
                                     c->xyz + 3 * (int64_t)acceptor);
    if (theta == ORC_PI) return 1; /* NaN path: Python float pi >= 2.09 */
    return ((float)theta >= (float)K_XBOND_THETA1) ? 1 : 0;
}

/* I:643-691 */
int orc_contact_type(int bs, int es, int bw, int ew) {
    int ct = -1;
    if (!bs && !es) ct = ARP_CT_INTRA_NON_SELECTION;
    if (bs && es) ct = ARP_CT_INTRA_SELECTION;
    if ((bs && !es) || (es && !bs)) ct = ARP_CT_INTER;
    if ((bs && ew) || (es && bw)) ct = ARP_CT_SELECTION_WATER;
    if ((!bs && ew) || (!es && bw)) ct = ARP_CT_NON_SELECTION_WATER;
    if (bw && ew) ct = ARP_CT_WATER_WATER;
    return ct;
}

/* ------------------------------------------------------------------------- */
/* I:693-936: one iteration of the loop body for the ordered pair (b, e).     */
/* Returns 1 if an AtomAtomContact is appended, 0 if a `continue` fires.      */
/* ------------------------------------------------------------------------- */
int orc_pair_contact(const orc_complex* c, int b, int e, double comp, int seq_adj,
                     float* out_dist, uint16_t* out_sift, uint8_t* out_ctype, int* err) {
    const uint16_t fb = c->flags[b], fe = c->flags[e];
    const uint16_t tb = c->tmask[b], te = c->tmask[e];
    /* I:712 */
    if ((fb & ARP_F_HYDROGEN) || (fe & ARP_F_HYDROGEN)) return 0;
    /* I:715 */
    int bw = (fb & ARP_F_WATER) != 0, ew = (fe & ARP_F_WATER) != 0;
    int ct = orc_contact_type(c->in_sel[b] != 0, c->in_sel[e] != 0, bw, ew);
    /* I:717-718 */
    double sum_cov = c->cov[b] + c->cov[e];
    double sum_vdw = c->vdw[b] + c->vdw[e];
    /* I:726-730 */
    int rb = c->res_id[b], re = c->res_id[e];
    if (rb == re) return 0;
    /* I:733-741 (res_end.is_polypeptide tested twice, res_bgn never) */
    if (!seq_adj) {
        if (c->res_flags[re] & ARP_R_POLYPEPTIDE) {
            if ((c->res_flags[rb] & ARP_R_HAS_SEQ) && (c->res_flags[re] & ARP_R_HAS_SEQ)) {
                if (c->res_next[rb] == re || c->res_prev[rb] == re ||
                    c->res_next[re] == rb || c->res_prev[re] == rb)
                    return 0;
            }
        }
    }
    /* I:745 */
    const float* xb = c->xyz + 3 * (int64_t)b;
    const float* xe = c->xyz + 3 * (int64_t)e;
    float d = orc_dist_f32(xb, xe);
    uint16_t s = 0;
    /* I:748-757 */
    int cov = 0;
    for (int k = c->bond_off[b]; k < c->bond_off[b + 1]; ++k)
        if (c->bond_idx[k] == e) { cov = 1; break; }
    /* I:756-773: float32 distance against Python floats -> float32 compare */
    double vdw_comp = sum_vdw + comp;
    if (cov) s |= ARP_S_COVALENT;
    else if (d < (float)sum_cov) s |= ARP_S_CLASH;
    else if (d < (float)sum_vdw) s |= ARP_S_VDW_CLASH;
    else if (d <= (float)vdw_comp) s |= ARP_S_VDW;
    else s |= ARP_S_PROXIMAL;
    /* I:777-783 */
    if (d <= (float)K_METAL) {
        if ((tb & ARP_T_HBOND_ACCEPTOR) && (fe & ARP_F_METAL)) s |= ARP_S_METAL_COMPLEX;
        else if ((te & ARP_T_HBOND_ACCEPTOR) && (fb & ARP_F_METAL)) s |= ARP_S_METAL_COMPLEX;
    }
    /* I:786 */
    if (!(s & ARP_S_CLASH) && d <= (float)K_DIST_MAX) {
        const double* hb = c->h_xyz + 3 * (int64_t)c->h_off[b];
        const double* he = c->h_xyz + 3 * (int64_t)c->h_off[e];
        int nhb = c->h_off[b + 1] - c->h_off[b], nhe = c->h_off[e + 1] - c->h_off[e];
        /* I:791-819 */
        if (bw && d <= (float)vdw_comp) {
            if (te & (ARP_T_HBOND_ACCEPTOR | ARP_T_HBOND_DONOR)) s |= ARP_S_HBOND | ARP_S_POLAR;
        } else if (ew && d <= (float)vdw_comp) {
            if (tb & (ARP_T_HBOND_ACCEPTOR | ARP_T_HBOND_DONOR)) s |= ARP_S_HBOND | ARP_S_POLAR;
        } else {
            if ((tb & ARP_T_HBOND_DONOR) && (te & ARP_T_HBOND_ACCEPTOR)) {
                if (orc_is_hbond_like(xb, hb, nhb, xe, c->vdw[e], comp, K_HBOND_ANGLE)) s |= ARP_S_HBOND;
                if (d <= (float)K_HBOND_POLAR) s |= ARP_S_POLAR;
            } else if ((te & ARP_T_HBOND_DONOR) && (tb & ARP_T_HBOND_ACCEPTOR)) {
                if (orc_is_hbond_like(xe, he, nhe, xb, c->vdw[b], comp, K_HBOND_ANGLE)) s |= ARP_S_HBOND;
                if (d <= (float)K_HBOND_POLAR) s |= ARP_S_POLAR;
            }
        }
        /* I:857-886: four independent ifs, each overwrites SIFt[6] */
        int weak = 0;
        if ((tb & ARP_T_HBOND_ACCEPTOR) && (te & ARP_T_WEAK_HBOND_DONOR)) {
            weak = orc_is_hbond_like(xe, he, nhe, xb, c->vdw[b], comp, K_WEAK_ANGLE);
            if (d <= (float)K_WEAK_POLAR) s |= ARP_S_WEAK_POLAR;
        }
        if ((tb & ARP_T_WEAK_HBOND_DONOR) && (te & ARP_T_HBOND_ACCEPTOR)) {
            weak = orc_is_hbond_like(xb, hb, nhb, xe, c->vdw[e], comp, K_WEAK_ANGLE);
            if (d <= (float)K_WEAK_POLAR) s |= ARP_S_WEAK_POLAR;
        }
        if ((tb & ARP_T_WEAK_HBOND_ACCEPTOR) && (fb & ARP_F_HALOGEN) &&
            (te & (ARP_T_HBOND_DONOR | ARP_T_WEAK_HBOND_DONOR))) {
            weak = is_halogen_weak_hbond(c, e, b, comp);
            if (d <= (float)K_WEAK_POLAR) s |= ARP_S_WEAK_POLAR;
        }
        if ((te & ARP_T_WEAK_HBOND_ACCEPTOR) && (fe & ARP_F_HALOGEN) &&
            (tb & (ARP_T_HBOND_DONOR | ARP_T_WEAK_HBOND_DONOR))) {
            weak = is_halogen_weak_hbond(c, b, e, comp);
            if (d <= (float)K_WEAK_POLAR) s |= ARP_S_WEAK_POLAR;
        }
        if (weak) s |= ARP_S_WEAK_HBOND;
        /* I:889-895 */
        if (d <= (float)vdw_comp) {
            if ((tb & ARP_T_XBOND_DONOR) && (te & ARP_T_XBOND_ACCEPTOR)) {
                if (is_xbond(c, b, e, err)) s |= ARP_S_XBOND;
            } else if ((te & ARP_T_XBOND_DONOR) && (tb & ARP_T_XBOND_ACCEPTOR)) {
                if (is_xbond(c, e, b, err)) s |= ARP_S_XBOND;
            }
        }
        /* I:898-904 */
        if (d <= (float)K_IONIC) {
            if ((tb & ARP_T_POS_IONISABLE) && (te & ARP_T_NEG_IONISABLE)) s |= ARP_S_IONIC;
            else if ((tb & ARP_T_NEG_IONISABLE) && (te & ARP_T_POS_IONISABLE)) s |= ARP_S_IONIC;
        }
        /* I:907-913 */
        if (d <= (float)K_CARBONYL) {
            if ((tb & ARP_T_CARBONYL_OXYGEN) && (te & ARP_T_CARBONYL_CARBON)) s |= ARP_S_CARBONYL;
            else if ((te & ARP_T_CARBONYL_OXYGEN) && (tb & ARP_T_CARBONYL_CARBON)) s |= ARP_S_CARBONYL;
        }
        /* I:916-917 */
        if ((tb & ARP_T_AROMATIC) && (te & ARP_T_AROMATIC) && d <= (float)K_AROMATIC) s |= ARP_S_AROMATIC;
        /* I:920-921 */
        if ((tb & ARP_T_HYDROPHOBE) && (te & ARP_T_HYDROPHOBE) && d <= (float)K_HYDROPHOBIC)
            s |= ARP_S_HYDROPHOBIC;
    }
    *out_dist = d;
    *out_sift = s;
    *out_ctype = (uint8_t)ct;
    return 1;
}

/* ------------------------------------------------------------------------- */
/* NeighborSearch stand-ins                                                   */
/* ------------------------------------------------------------------------- */

/* Ground truth: O(N^2) enumeration of pairs (i<j), inclusive float64 test. */
int64_t orc_search_all_brute(int64_t n, const float* xyz, const uint8_t* active, double radius,
                             int64_t cap, int32_t* out_i, int32_t* out_j) {
    double r2 = radius * radius;
    int64_t cnt = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (active && !active[i]) continue;
        for (int64_t j = i + 1; j < n; ++j) {
            if (active && !active[j]) continue;
            if (within_f64(xyz + 3 * i, xyz + 3 * j, r2)) {
                if (cnt < cap) { out_i[cnt] = (int32_t)i; out_j[cnt] = (int32_t)j; }
                ++cnt;
            }
        }
    }
    return cnt;
}

typedef struct {
    double ox, oy, oz, inv;
    int nx, ny, nz;
    int64_t ncell;
    int32_t* start; /* ncell+1 */
    int32_t* perm;  /* atoms sorted by cell, ascending id within a cell */
} orc_grid;

static int cell_of(const orc_grid* g, double v, double o, int nmax) {
    int cidx = (int)floor((v - o) * g->inv);
    if (cidx < 0) cidx = 0;
    if (cidx >= nmax) cidx = nmax - 1;
    return cidx;
}

static int grid_build(orc_grid* g, int64_t n, const float* xyz, const uint8_t* active, double edge) {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (active && !active[i]) continue;
        ++m;
        for (int k = 0; k < 3; ++k) {
            double v = xyz[3 * i + k];
            if (v < lo[k]) lo[k] = v;
            if (v > hi[k]) hi[k] = v;
        }
    }
    memset(g, 0, sizeof(*g));
    if (m == 0) { g->nx = g->ny = g->nz = 1; g->ncell = 1; lo[0] = lo[1] = lo[2] = 0; hi[0] = hi[1] = hi[2] = 0; }
    edge *= 1.000001;
    for (;;) {
        g->nx = (int)floor((hi[0] - lo[0]) / edge) + 1;
        g->ny = (int)floor((hi[1] - lo[1]) / edge) + 1;
        g->nz = (int)floor((hi[2] - lo[2]) / edge) + 1;
        g->ncell = (int64_t)g->nx * g->ny * g->nz;
        if (g->ncell <= (int64_t)1 << 26) break;
        edge *= 1.26;
    }
    g->ox = lo[0]; g->oy = lo[1]; g->oz = lo[2]; g->inv = 1.0 / edge;
    g->start = (int32_t*)calloc((size_t)g->ncell + 1, sizeof(int32_t));
    g->perm = (int32_t*)malloc((size_t)(m > 0 ? m : 1) * sizeof(int32_t));
    if (!g->start || !g->perm) return -1;
    for (int64_t i = 0; i < n; ++i) {
        if (active && !active[i]) continue;
        int cx = cell_of(g, xyz[3 * i], g->ox, g->nx), cy = cell_of(g, xyz[3 * i + 1], g->oy, g->ny),
            cz = cell_of(g, xyz[3 * i + 2], g->oz, g->nz);
        g->start[((int64_t)cz * g->ny + cy) * g->nx + cx + 1]++;
    }
    for (int64_t c = 0; c < g->ncell; ++c) g->start[c + 1] += g->start[c];
    int32_t* fill = (int32_t*)calloc((size_t)g->ncell, sizeof(int32_t));
    if (!fill) return -1;
    for (int64_t i = 0; i < n; ++i) { /* ascending i => ascending id within a cell */
        if (active && !active[i]) continue;
        int cx = cell_of(g, xyz[3 * i], g->ox, g->nx), cy = cell_of(g, xyz[3 * i + 1], g->oy, g->ny),
            cz = cell_of(g, xyz[3 * i + 2], g->oz, g->nz);
        int64_t cc = ((int64_t)cz * g->ny + cy) * g->nx + cx;
        g->perm[g->start[cc] + fill[cc]++] = (int32_t)i;
    }
    free(fill);
    return 0;
}

static void grid_free(orc_grid* g) { free(g->start); free(g->perm); }

/* Cell-list search_all; same pair set as orc_search_all_brute (checked in tests).
 * candidates (optional) receives the number of distance tests performed. */
int64_t orc_search_all_grid(int64_t n, const float* xyz, const uint8_t* active, double radius,
                            int64_t cap, int32_t* out_i, int32_t* out_j, int64_t* candidates) {
    orc_grid g;
    if (grid_build(&g, n, xyz, active, radius) != 0) return -1;
    double r2 = radius * radius;
    int64_t cnt = 0, cand = 0;
    for (int cz = 0; cz < g.nz; ++cz)
        for (int cy = 0; cy < g.ny; ++cy)
            for (int cx = 0; cx < g.nx; ++cx) {
                int64_t c0 = ((int64_t)cz * g.ny + cy) * g.nx + cx;
                for (int32_t p = g.start[c0]; p < g.start[c0 + 1]; ++p) {
                    int32_t i = g.perm[p];
                    /* half stencil: own cell (later entries) + 13 forward cells */
                    for (int dz = 0; dz <= 1; ++dz)
                        for (int dy = (dz ? -1 : 0); dy <= 1; ++dy)
                            for (int dx = ((dz || dy) ? -1 : 0); dx <= 1; ++dx) {
                                int x2 = cx + dx, y2 = cy + dy, z2 = cz + dz;
                                if (x2 < 0 || x2 >= g.nx || y2 < 0 || y2 >= g.ny || z2 >= g.nz) continue;
                                int64_t c1 = ((int64_t)z2 * g.ny + y2) * g.nx + x2;
                                int32_t q0 = (c1 == c0) ? p + 1 : g.start[c1];
                                for (int32_t q = q0; q < g.start[c1 + 1]; ++q) {
                                    int32_t j = g.perm[q];
                                    ++cand;
                                    if (within_f64(xyz + 3 * (int64_t)i, xyz + 3 * (int64_t)j, r2)) {
                                        if (cnt < cap) {
                                            out_i[cnt] = i < j ? i : j;
                                            out_j[cnt] = i < j ? j : i;
                                        }
                                        ++cnt;
                                    }
                                }
                            }
                }
            }
    grid_free(&g);
    if (candidates) *candidates = cand;
    return cnt;
}

/* NeighborSearch.search(center, radius): atoms with float64 d^2 <= r^2 (brute force). */
int64_t orc_search_point(int64_t n, const float* xyz, const uint8_t* active, const double* center,
                         double radius, int64_t cap, int32_t* out) {
    double r2 = radius * radius;
    int64_t cnt = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (active && !active[i]) continue;
        double dx = center[0] - (double)xyz[3 * i], dy = center[1] - (double)xyz[3 * i + 1],
               dz = center[2] - (double)xyz[3 * i + 2];
        double r = dx * dx;
        r += dy * dy;
        r += dz * dz;
        if (r <= r2) { if (cnt < cap) out[cnt] = (int32_t)i; ++cnt; }
    }
    return cnt;
}

/* ------------------------------------------------------------------------- */
/* I:1384-1451 _make_selection (set semantics; order-free)                    */
/* ------------------------------------------------------------------------- */
int orc_make_selection(const orc_complex* c, const uint8_t* in_sel, double radius, int use_grid,
                       uint8_t* out_plus, uint8_t* ring_sel, uint8_t* ring_plus,
                       uint8_t* amide_sel, uint8_t* amide_plus) {
    int64_t n = c->n;
    memcpy(out_plus, in_sel, (size_t)n); /* I:1407 */
    /* I:1420-1424 over ALL atoms (hydrogens included) */
    int64_t cap = 0, cnt;
    int32_t *pi = NULL, *pj = NULL;
    cnt = use_grid ? orc_search_all_grid(n, c->xyz, NULL, radius, 0, NULL, NULL, NULL)
                   : orc_search_all_brute(n, c->xyz, NULL, radius, 0, NULL, NULL);
    if (cnt < 0) return -1;
    cap = cnt > 0 ? cnt : 1;
    pi = (int32_t*)malloc((size_t)cap * 4);
    pj = (int32_t*)malloc((size_t)cap * 4);
    if (!pi || !pj) return -1;
    if (use_grid) orc_search_all_grid(n, c->xyz, NULL, radius, cap, pi, pj, NULL);
    else orc_search_all_brute(n, c->xyz, NULL, radius, cap, pi, pj);
    for (int64_t k = 0; k < cnt; ++k)
        if (in_sel[pi[k]] || in_sel[pj[k]]) { out_plus[pi[k]] = 1; out_plus[pj[k]] = 1; }
    free(pi); free(pj);
    /* I:1413, 1431: residue sets */
    uint8_t* rs = (uint8_t*)calloc((size_t)(c->nres > 0 ? c->nres : 1), 1);
    uint8_t* rp = (uint8_t*)calloc((size_t)(c->nres > 0 ? c->nres : 1), 1);
    if (!rs || !rp) return -1;
    for (int64_t i = 0; i < n; ++i) {
        if (in_sel[i]) rs[c->res_id[i]] = 1;
        if (out_plus[i]) rp[c->res_id[i]] = 1;
    }
    /* I:1416-1417, 1434-1437 (residue None never qualifies) */
    for (int64_t r = 0; r < c->nring; ++r) {
        int rr = c->ring_res[r];
        ring_sel[r] = (rr >= 0 && rs[rr]) ? 1 : 0;
        ring_plus[r] = (rr >= 0 && rp[rr]) ? 1 : 0;
    }
    for (int64_t a = 0; a < c->namide; ++a) {
        int rr = c->amide_res[a];
        amide_sel[a] = (rr >= 0 && rs[rr]) ? 1 : 0;
        amide_plus[a] = (rr >= 0 && rp[rr]) ? 1 : 0;
    }
    free(rs); free(rp);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* I:693-936 whole loop: search_all(cutoff) on selection_plus, canonical       */
/* orientation bgn = lower packed index.                                       */
/* ------------------------------------------------------------------------- */
int64_t orc_atom_contacts(const orc_complex* c, double cutoff, double comp, int seq_adj, int use_grid,
                          int64_t cap, int32_t* out_i, int32_t* out_j, float* out_dist,
                          uint16_t* out_sift, uint8_t* out_ctype, int64_t* stats, int* err) {
    int64_t n = c->n;
    /* tree over selection_plus; hydrogens are dropped at I:712, so leave them out */
    uint8_t* act = (uint8_t*)malloc((size_t)(n > 0 ? n : 1));
    if (!act) return -1;
    for (int64_t i = 0; i < n; ++i) act[i] = c->in_plus[i] && !(c->flags[i] & ARP_F_HYDROGEN);
    int64_t cand = 0;
    int64_t np = use_grid ? orc_search_all_grid(n, c->xyz, act, cutoff, 0, NULL, NULL, &cand)
                          : orc_search_all_brute(n, c->xyz, act, cutoff, 0, NULL, NULL);
    if (np < 0) { free(act); return -1; }
    int64_t pc = np > 0 ? np : 1;
    int32_t* pi = (int32_t*)malloc((size_t)pc * 4);
    int32_t* pj = (int32_t*)malloc((size_t)pc * 4);
    if (!pi || !pj) { free(act); return -1; }
    if (use_grid) orc_search_all_grid(n, c->xyz, act, cutoff, pc, pi, pj, NULL);
    else orc_search_all_brute(n, c->xyz, act, cutoff, pc, pi, pj);
    free(act);
    int64_t cnt = 0;
    *err = 0;
    for (int64_t k = 0; k < np; ++k) {
        float d; uint16_t s; uint8_t ct;
        if (orc_pair_contact(c, pi[k], pj[k], comp, seq_adj, &d, &s, &ct, err)) {
            if (cnt < cap) { out_i[cnt] = pi[k]; out_j[cnt] = pj[k]; out_dist[cnt] = d; out_sift[cnt] = s; out_ctype[cnt] = ct; }
            ++cnt;
        }
    }
    free(pi); free(pj);
    if (stats) { stats[0] = cand; stats[1] = np; stats[2] = cnt; }
    return cnt;
}

/* contact-type chain shared by I:985-997, 1095-1108, 1252-1265, 1333-1346 */
static int plane_ctype(int a_sel, int b_sel, int a_plus, int b_plus) {
    int ct = -1;
    if (!a_sel && !b_sel) ct = ARP_CT_INTRA_NON_SELECTION;
    if (a_plus && b_plus) ct = ARP_CT_INTRA_BINDING_SITE;
    if (a_sel && b_sel) ct = ARP_CT_INTRA_SELECTION;
    if ((a_sel && !b_sel) || (b_sel && !a_sel)) ct = ARP_CT_INTER;
    return ct;
}

/* I:947-1062 */
int64_t orc_atom_plane(const orc_complex* c, int64_t cap, int32_t* out_atom, int32_t* out_ring,
                       double* out_dist, double* out_theta, uint8_t* out_mask, uint8_t* out_ctype) {
    int64_t cnt = 0;
    for (int64_t r = 0; r < c->nring; ++r) {
        if (!c->ring_plus[r]) continue; /* I:957 */
        const double* ctr = c->ring_center + 3 * r;
        const double* nrm = c->ring_normal + 3 * r;
        for (int64_t a = 0; a < c->n; ++a) {
            if (!c->in_plus[a]) continue; /* tree is built on selection_plus (I:1442) */
            const float* x = c->xyz + 3 * a;
            double dx = ctr[0] - (double)x[0], dy = ctr[1] - (double)x[1], dz = ctr[2] - (double)x[2];
            double rr = dx * dx; rr += dy * dy; rr += dz * dz;
            if (!(rr <= K_MET_SULPHUR * K_MET_SULPHUR)) continue; /* I:960 */
            if (c->flags[a] & ARP_F_HYDROGEN) continue;           /* I:964 */
            /* I:972 distance = norm(atom.coord - center), float64 */
            double v[3] = {(double)x[0] - ctr[0], (double)x[1] - ctr[1], (double)x[2] - ctr[2]};
            double dist = orc_norm_f64(v);
            if (c->tmask[a] & ARP_T_AROMATIC) continue;           /* I:975 */
            int ct = plane_ctype(c->ring_sel[r], c->in_sel[a], c->ring_plus[r], c->in_plus[a]);
            double p[3] = {ctr[0] - (double)x[0], ctr[1] - (double)x[1], ctr[2] - (double)x[2]};
            double theta = orc_group_angle_f64(nrm, p); /* I:1005 */
            uint8_t m = 0;
            if (dist <= K_ATOM_AROMATIC && theta <= 30.0) {
                if ((c->flags[a] & ARP_F_ELEM_C) && (c->tmask[a] & ARP_T_WEAK_HBOND_DONOR)) m |= ARP_AP_CARBONPI;
                if (c->tmask[a] & ARP_T_POS_IONISABLE) m |= ARP_AP_CATIONPI;
                if (c->tmask[a] & ARP_T_HBOND_DONOR) m |= ARP_AP_DONORPI;
                if (c->tmask[a] & ARP_T_XBOND_DONOR) m |= ARP_AP_HALOGENPI;
            }
            if (dist <= K_MET_SULPHUR) {
                if ((c->flags[a] & ARP_F_RES_MET) && (c->flags[a] & ARP_F_ELEM_S)) m |= ARP_AP_METSULPHURPI;
            }
            if (!m) continue; /* I:1026 */
            if (cnt < cap) {
                out_atom[cnt] = (int32_t)a; out_ring[cnt] = (int32_t)r; out_dist[cnt] = dist;
                out_theta[cnt] = theta; out_mask[cnt] = m; out_ctype[cnt] = (uint8_t)ct;
            }
            ++cnt;
        }
    }
    return cnt;
}

/* I:1127-1148 */
int orc_pp_class(double dihedral, double theta) {
    if (dihedral <= 30.0 && theta <= 30.0) return ARP_PP_FF;
    else if (dihedral <= 30.0 && theta <= 60.0) return ARP_PP_OF;
    else if (dihedral <= 30.0 && theta <= 90.0) return ARP_PP_EE;
    else if (30.0 < dihedral && dihedral <= 60.0 && theta <= 30.0) return ARP_PP_FT;
    else if (30.0 < dihedral && dihedral <= 60.0 && theta <= 60.0) return ARP_PP_OT;
    else if (30.0 < dihedral && dihedral <= 60.0 && theta <= 90.0) return ARP_PP_ET;
    else if (60.0 < dihedral && dihedral <= 90.0 && theta <= 30.0) return ARP_PP_FE;
    else if (60.0 < dihedral && dihedral <= 90.0 && theta <= 60.0) return ARP_PP_OE;
    else if (60.0 < dihedral && dihedral <= 90.0 && theta <= 90.0) return ARP_PP_EF;
    return ARP_PP_NONE;
}

/* I:1064-1194, visiting ordered pairs exactly like the reference and applying its
 * dedupe; records come out in creation order.  Every record created during the
 * outer iteration r1 has bgn == r1, so the records of one bgn ring are contiguous:
 * seg[r] / seg[r+1] bracket them and replace the reference's linear scan (I:1181). */
int64_t orc_plane_plane(const orc_complex* c, int64_t cap, int32_t* out_bgn, int32_t* out_end,
                        double* out_dist, double* out_dihedral, double* out_theta_bgn,
                        double* out_theta_end, uint8_t* out_type1, uint8_t* out_type2,
                        uint8_t* out_ctype) {
    int64_t cnt = 0;
    int64_t* seg = (int64_t*)calloc((size_t)c->nring + 2, sizeof(int64_t));
    if (!seg) return -1;
    for (int64_t r1 = 0; r1 < c->nring; ++r1) {
        seg[r1] = cnt;
        for (int64_t r2 = 0; r2 < c->nring; ++r2) {
            if (!c->ring_plus[r1] || !c->ring_plus[r2]) continue; /* I:1081 */
            if (r1 == r2) continue;                                /* I:1085 */
            int intra = c->ring_res[r1] == c->ring_res[r2];        /* I:1091 */
            int ct = plane_ctype(c->ring_sel[r1], c->ring_sel[r2], c->ring_plus[r1], c->ring_plus[r2]);
            const double* c1 = c->ring_center + 3 * r1;
            const double* c2 = c->ring_center + 3 * r2;
            double p[3] = {c1[0] - c2[0], c1[1] - c2[1], c1[2] - c2[2]};
            double dist = orc_norm_f64(p);
            if (dist > K_CENTROID) continue; /* I:1113 */
            const double* n1 = c->ring_normal + 3 * r1;
            const double* n2 = c->ring_normal + 3 * r2;
            double cosd = orc_dot_f64(n1, n2) / (orc_norm_f64(n1) * orc_norm_f64(n2));
            double dihedral = fabs(fold_deg_f64(acos(cosd))); /* I:1122 */
            double theta = orc_group_angle_f64(n1, p);        /* I:1123 */
            int ty = orc_pp_class(dihedral, theta);
            if (intra && ty == ARP_PP_EE) continue; /* I:1154 */
            /* I:1181-1194 dedupe: an earlier record can only be (bgn=r2, end=r1), r2 < r1 */
            int64_t found = -1;
            if (r2 < r1)
                for (int64_t k = seg[r2]; k < seg[r2 + 1] && k < cap; ++k)
                    if (out_bgn[k] == r2 && out_end[k] == r1) { found = k; break; }
            if (found >= 0) {
                out_type2[found] = (out_type1[found] != ty) ? (uint8_t)ty : ARP_PP_SAME; /* appended if different */
                out_theta_end[found] = theta;
            } else {
                if (cnt < cap) {
                    out_bgn[cnt] = (int32_t)r1; out_end[cnt] = (int32_t)r2; out_dist[cnt] = dist;
                    out_dihedral[cnt] = dihedral; out_theta_bgn[cnt] = theta; out_theta_end[cnt] = NAN;
                    out_type1[cnt] = (uint8_t)ty; out_type2[cnt] = ARP_PP_SKIPPED; out_ctype[cnt] = (uint8_t)ct;
                }
                ++cnt;
            }
        }
        seg[r1 + 1] = cnt;
    }
    free(seg);
    return cnt;
}

/* I:1217-1300 (float32 arithmetic) */
int64_t orc_group_group(const orc_complex* c, int64_t cap, int32_t* out_bgn, int32_t* out_end,
                        float* out_dist, float* out_dihedral, float* out_theta, uint8_t* out_ctype) {
    int64_t cnt = 0;
    for (int64_t a1 = 0; a1 < c->namide; ++a1) {
        if (!c->amide_plus[a1]) continue;
        for (int64_t a2 = 0; a2 < c->namide; ++a2) {
            if (a1 == a2) continue;
            if (!c->amide_plus[a2]) continue;
            int ct = plane_ctype(c->amide_sel[a1], c->amide_sel[a2], c->amide_plus[a1], c->amide_plus[a2]);
            const float* c1 = c->amide_center + 3 * a1;
            const float* c2 = c->amide_center + 3 * a2;
            float p[3] = {c1[0] - c2[0], c1[1] - c2[1], c1[2] - c2[2]};
            float dist = orc_norm_f32(p);
            if (dist > (float)K_AMIDE_CENTROID) continue; /* I:1270 */
            const float* n1 = c->amide_normal + 3 * a1;
            const float* n2 = c->amide_normal + 3 * a2;
            float cosd = orc_dot_f32(n1, n2) / (orc_norm_f32(n1) * orc_norm_f32(n2));
            float dihedral = fabsf(fold_deg_f32(acosf(cosd)));
            float theta = orc_group_angle_f32(n1, p);
            if (dihedral > 30.0f || theta > 30.0f) continue; /* I:1282 */
            if (cnt < cap) {
                out_bgn[cnt] = (int32_t)a1; out_end[cnt] = (int32_t)a2; out_dist[cnt] = dist;
                out_dihedral[cnt] = dihedral; out_theta[cnt] = theta; out_ctype[cnt] = (uint8_t)ct;
            }
            ++cnt;
        }
    }
    return cnt;
}

/* I:1302-1382 (float32 amide against float64 ring -> float64) */
int64_t orc_group_plane(const orc_complex* c, int64_t cap, int32_t* out_amide, int32_t* out_ring,
                        double* out_dist, double* out_dihedral, double* out_theta, uint8_t* out_ctype) {
    int64_t cnt = 0;
    for (int64_t a = 0; a < c->namide; ++a) {
        if (!c->amide_plus[a]) continue;
        for (int64_t r = 0; r < c->nring; ++r) {
            if (!c->ring_plus[r]) continue;
            int ct = plane_ctype(c->amide_sel[a], c->ring_sel[r], c->amide_plus[a], c->ring_plus[r]);
            const float* ca = c->amide_center + 3 * a;
            const double* cr = c->ring_center + 3 * r;
            double p[3] = {(double)ca[0] - cr[0], (double)ca[1] - cr[1], (double)ca[2] - cr[2]};
            double dist = orc_norm_f64(p);
            if (dist > K_AMIDE_CENTROID) continue; /* I:1351 */
            const float* na = c->amide_normal + 3 * a;
            const double* nr = c->ring_normal + 3 * r;
            double nad[3] = {(double)na[0], (double)na[1], (double)na[2]};
            double cosd = orc_dot_f64(nad, nr) / ((double)orc_norm_f32(na) * orc_norm_f64(nr));
            double dihedral = fabs(fold_deg_f64(acos(cosd)));
            double theta = orc_group_angle_f32n_f64p(na, p);
            if (dihedral > 30.0 || theta > 30.0) continue; /* I:1363 */
            if (cnt < cap) {
                out_amide[cnt] = (int32_t)a; out_ring[cnt] = (int32_t)r; out_dist[cnt] = dist;
                out_dihedral[cnt] = dihedral; out_theta[cnt] = theta; out_ctype[cnt] = (uint8_t)ct;
            }
            ++cnt;
        }
    }
    return cnt;
}

/* U:182-242 accumulators on 15-bit masks: state = {sift, inter, intra, water} as
 * bit masks plus integer_sift[4][15]; restated for the golden test only. */
void orc_update_atom_sift(uint16_t state[4], uint8_t integer_sift[4][15], uint16_t addition, int ctype) {
    int inter = ctype == ARP_CT_INTER; /* == 'INTER' (U:193) */
    int intra = ctype == ARP_CT_INTRA_NON_SELECTION || ctype == ARP_CT_INTRA_SELECTION ||
                ctype == ARP_CT_INTRA_BINDING_SITE; /* 'INTRA' in */
    int water = ctype == ARP_CT_SELECTION_WATER || ctype == ARP_CT_NON_SELECTION_WATER ||
                ctype == ARP_CT_WATER_WATER; /* 'WATER' in */
    /* update_atom_integer_sift is called first (I:924-925): old binary sift + addition (U:233) */
    for (int k = 0; k < 15; ++k) {
        int add = (addition >> k) & 1;
        integer_sift[0][k] = (uint8_t)(((state[0] >> k) & 1) + add);
        if (inter) integer_sift[1][k] = (uint8_t)(((state[1] >> k) & 1) + add);
        if (intra) integer_sift[2][k] = (uint8_t)(((state[2] >> k) & 1) + add);
        if (water) integer_sift[3][k] = (uint8_t)(((state[3] >> k) & 1) + add);
    }
    state[0] |= addition;
    if (inter) state[1] |= addition;
    if (intra) state[2] |= addition;
    if (water) state[3] |= addition;
}

/* I:821-852 + I:923-934 over a finished contact list: per-atom masks {sift, inter, intra, water} and the eight
 * hbond / polar counters.  Restated independently of the GPU kernel: walks the contacts in the given order. */
void orc_atom_accumulators(int64_t n, int64_t np, const int32_t* ci, const int32_t* cj, const uint16_t* cs,
                           const uint8_t* cct, uint16_t* out_sift4, int32_t* out_counts8) {
    memset(out_sift4, 0, (size_t)n * 4 * sizeof(uint16_t));
    memset(out_counts8, 0, (size_t)n * 8 * sizeof(int32_t));
    for (int64_t p = 0; p < np; ++p) {
        const int ct = cct[p];
        const char* names[] = {"INTRA_NON_SELECTION", "INTRA_SELECTION", "INTER", "SELECTION_WATER", "NON_SELECTION_WATER",
                               "WATER_WATER", "INTRA_BINDING_SITE"};
        const char* t = names[ct];
        const int is_inter = strcmp(t, "INTER") == 0;          /* contact_type == 'INTER' (U:193) */
        const int has_intra = strstr(t, "INTRA") != NULL;      /* 'INTRA' in contact_type */
        const int has_inter = strstr(t, "INTER") != NULL;      /* 'INTER' in contact_type (I:830) */
        const int has_water = strstr(t, "WATER") != NULL;
        const int32_t atoms[2] = {ci[p], cj[p]};
        for (int k = 0; k < 2; ++k) {
            uint16_t* s4 = out_sift4 + 4 * (int64_t)atoms[k];
            int32_t* c8 = out_counts8 + 8 * (int64_t)atoms[k];
            s4[0] |= cs[p];
            if (is_inter) s4[1] |= cs[p];
            if (has_intra) s4[2] |= cs[p];
            if (has_water) s4[3] |= cs[p];
            for (int f = 0; f < 2; ++f) {                       /* f = 0: hbond (SIFt[5]), 1: polar (SIFt[13]) */
                if (!(cs[p] & (f ? ARP_S_POLAR : ARP_S_HBOND))) continue;
                c8[4 * f] += 1;
                if (has_intra) c8[4 * f + 1] += 1;
                else if (has_inter) c8[4 * f + 2] += 1;
                else if (has_water) c8[4 * f + 3] += 1;
            }
        }
    }
}

/* I:923-925 + U:224-242 over a contact list IN THE ORDER GIVEN (the reference's values depend on the order in which
 * its KD-tree delivers the pairs; this project defines them on the canonical order, contacts sorted by (i, j)):
 * integer_sift{,_inter_only,_intra_only,_water_only}[15] per atom = binary sift before the atom's last pair of that
 * class + that pair's SIFt. */
void orc_atom_integer_sifts(int64_t n, int64_t np, const int32_t* ci, const int32_t* cj, const uint16_t* cs,
                            const uint8_t* cct, uint8_t* out_isift /* [n][4][15] */) {
    uint16_t* state = (uint16_t*)calloc((size_t)(n > 0 ? n : 1) * 4, sizeof(uint16_t));
    memset(out_isift, 0, (size_t)n * 60);
    for (int64_t p = 0; p < np; ++p) {
        /* both integer updates run before the binary ones (I:924-929), and an atom never pairs with itself */
        orc_update_atom_sift(state + 4 * (int64_t)ci[p], (uint8_t(*)[15])(out_isift + 60 * (int64_t)ci[p]), cs[p], cct[p]);
        orc_update_atom_sift(state + 4 * (int64_t)cj[p], (uint8_t(*)[15])(out_isift + 60 * (int64_t)cj[p]), cs[p], cct[p]);
    }
    free(state);
}

/* ------------------------------------------------------------------------- */
/* Timing-only multi-core variant of one whole-structure pass (bench.py's     */
/* "stronger than the reference" CPU figure): the same grid     */
/* search, pair test and per-pair evaluation as above with the cell loops     */
/* spread over OpenMP threads; contacts go to per-thread buffers and only     */
/* their number and a checksum leave.  Compiled into liborc_omp.so with       */
/* -fopenmp; the checker library (liborc.so) does not contain it.             */
/* ------------------------------------------------------------------------- */
#ifdef _OPENMP
#include <omp.h>

typedef struct { int32_t i, j; float d; uint16_t s; uint8_t ct; } orc_rec;

int orc_pass_openmp(const orc_complex* c, double cutoff, double comp, int seq_adj, int nthreads, int64_t* stats) {
    const int64_t n = c->n;
    if (nthreads > 0) omp_set_num_threads(nthreads);
    /* _make_selection: search_all(6.0) over all atoms, either atom selected -> both in selection_plus (I:1420-1424) */
    uint8_t* plus = (uint8_t*)malloc((size_t)(n > 0 ? n : 1));
    uint8_t* act = (uint8_t*)malloc((size_t)(n > 0 ? n : 1));
    if (!plus || !act) return -1;
    memcpy(plus, c->in_sel, (size_t)n);
    orc_grid g;
    if (grid_build(&g, n, c->xyz, NULL, 6.0) != 0) return -1;
    int64_t cand6 = 0, pairs6 = 0;
    const double r6 = 36.0;
#pragma omp parallel for collapse(2) schedule(dynamic, 4) reduction(+ : cand6, pairs6)
    for (int cz = 0; cz < g.nz; ++cz)
        for (int cy = 0; cy < g.ny; ++cy)
            for (int cx = 0; cx < g.nx; ++cx) {
                int64_t c0 = ((int64_t)cz * g.ny + cy) * g.nx + cx;
                for (int32_t p = g.start[c0]; p < g.start[c0 + 1]; ++p) {
                    int32_t i = g.perm[p];
                    for (int dz = 0; dz <= 1; ++dz)
                        for (int dy = (dz ? -1 : 0); dy <= 1; ++dy)
                            for (int dx = ((dz || dy) ? -1 : 0); dx <= 1; ++dx) {
                                int x2 = cx + dx, y2 = cy + dy, z2 = cz + dz;
                                if (x2 < 0 || x2 >= g.nx || y2 < 0 || y2 >= g.ny || z2 >= g.nz) continue;
                                int64_t c1 = ((int64_t)z2 * g.ny + y2) * g.nx + x2;
                                int32_t q0 = (c1 == c0) ? p + 1 : g.start[c1];
                                for (int32_t q = q0; q < g.start[c1 + 1]; ++q) {
                                    int32_t j = g.perm[q];
                                    ++cand6;
                                    if (within_f64(c->xyz + 3 * (int64_t)i, c->xyz + 3 * (int64_t)j, r6)) {
                                        ++pairs6;
                                        if (c->in_sel[i] || c->in_sel[j]) { plus[i] = 1; plus[j] = 1; }   /* (stores of 1 only) */
                                    }
                                }
                            }
                }
            }
    grid_free(&g);
    /* _calculate_atom_contacts on selection_plus without hydrogens */
    for (int64_t i = 0; i < n; ++i) act[i] = plus[i] && !(c->flags[i] & ARP_F_HYDROGEN);
    orc_complex cc = *c;
    cc.in_plus = plus;
    if (grid_build(&g, n, c->xyz, act, cutoff) != 0) return -1;
    int64_t cand5 = 0, contacts = 0, checksum = 0;
    const double r2 = cutoff * cutoff;
    int bad = 0;
#pragma omp parallel reduction(+ : cand5, contacts, checksum)
    {
        int64_t cap = 1 << 16, cnt = 0;
        orc_rec* buf = (orc_rec*)malloc((size_t)cap * sizeof(orc_rec));
        int err = 0;
#pragma omp for collapse(2) schedule(dynamic, 4)
        for (int cz = 0; cz < g.nz; ++cz)
            for (int cy = 0; cy < g.ny; ++cy)
                for (int cx = 0; cx < g.nx; ++cx) {
                    int64_t c0 = ((int64_t)cz * g.ny + cy) * g.nx + cx;
                    for (int32_t p = g.start[c0]; p < g.start[c0 + 1]; ++p) {
                        int32_t i = g.perm[p];
                        for (int dz = 0; dz <= 1; ++dz)
                            for (int dy = (dz ? -1 : 0); dy <= 1; ++dy)
                                for (int dx = ((dz || dy) ? -1 : 0); dx <= 1; ++dx) {
                                    int x2 = cx + dx, y2 = cy + dy, z2 = cz + dz;
                                    if (x2 < 0 || x2 >= g.nx || y2 < 0 || y2 >= g.ny || z2 >= g.nz) continue;
                                    int64_t c1 = ((int64_t)z2 * g.ny + y2) * g.nx + x2;
                                    int32_t q0 = (c1 == c0) ? p + 1 : g.start[c1];
                                    for (int32_t q = q0; q < g.start[c1 + 1]; ++q) {
                                        int32_t j = g.perm[q];
                                        ++cand5;
                                        if (!within_f64(c->xyz + 3 * (int64_t)i, c->xyz + 3 * (int64_t)j, r2)) continue;
                                        float d; uint16_t s; uint8_t ct;
                                        int b = i < j ? i : j, e = i < j ? j : i;
                                        if (buf && orc_pair_contact(&cc, b, e, comp, seq_adj, &d, &s, &ct, &err)) {
                                            if (cnt == cap) {
                                                cap *= 2;
                                                orc_rec* nb = (orc_rec*)realloc(buf, (size_t)cap * sizeof(orc_rec));
                                                if (!nb) { free(buf); buf = NULL; continue; }
                                                buf = nb;
                                            }
                                            buf[cnt].i = b; buf[cnt].j = e; buf[cnt].d = d; buf[cnt].s = s; buf[cnt].ct = ct;
                                            ++cnt;
                                            checksum += s;
                                        }
                                    }
                                }
                    }
                }
        contacts += cnt;
        if (!buf) {
#pragma omp atomic write
            bad = 1;
        }
        free(buf);
    }
    grid_free(&g);
    free(plus); free(act);
    if (stats) { stats[0] = cand6; stats[1] = pairs6; stats[2] = cand5; stats[3] = contacts; stats[4] = checksum; }
    return bad ? -1 : 0;
}
#endif
