// arp_planes.h — ring / amide kernels:
//   k_atom_plane   __calculate_atom_plane_contacts   interactions.py:947-1062
//   k_plane_plane  __calculate_plane_plane_contacts  interactions.py:1064-1194
//   k_group_group  __calculate_group_group_contacts  interactions.py:1217-1300
//   k_group_plane  __calculate_group_plane_contacts  interactions.py:1302-1382
// with utils.group_angle / group_group_angle (utils.py:638-693).
//
// The reference walks O(R^2), O(A^2) and O(A*R) Python loops; here each home item
// visits only the 27 cells of a uniform grid around it (cell edge >= 6.0 A, the
// centroid cut-off of config.py:617,619,624).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "arp_grid.h"
#include "arp_numerics.h"
#include "arp_pairs.h"

// contact-type chain shared by interactions.py:985-997, 1095-1108, 1252-1265, 1333-1346
__device__ __forceinline__ int plane_ctype(bool a_sel, bool b_sel, bool a_plus, bool b_plus) {
    int ct = 0;
    if (!a_sel && !b_sel) ct = ARP_CT_INTRA_NON_SELECTION;
    if (a_plus && b_plus) ct = ARP_CT_INTRA_BINDING_SITE;
    if (a_sel && b_sel) ct = ARP_CT_INTRA_SELECTION;
    if ((a_sel && !b_sel) || (b_sel && !a_sel)) ct = ARP_CT_INTER;
    return ct;
}

__device__ __forceinline__ num::d3 ld3(const double* p, int i) {
    return {p[3 * (size_t)i], p[3 * (size_t)i + 1], p[3 * (size_t)i + 2]};
}
__device__ __forceinline__ num::f3 lf3(const float* p, int i) {
    return {p[3 * (size_t)i], p[3 * (size_t)i + 1], p[3 * (size_t)i + 2]};
}

struct CellBox {  // integer cell coordinates of a point, unclamped
    int cx, cy, cz;
};
__device__ __forceinline__ CellBox cell_box(const GridDesc& g, num::d3 p) {
    return {cell_coord_raw(p.x, g.ox, g.inv, g.nx), cell_coord_raw(p.y, g.oy, g.inv, g.ny),
            cell_coord_raw(p.z, g.oz, g.inv, g.nz)};
}
// ... of a point of structure sid in a grid that holds several structures (clamped into the structure's own cells: one
// step further is the empty gap around it, never another structure)
__device__ __forceinline__ CellBox cell_box(const GridDesc& g, num::d3 p, int sid) {
    if (!g.place) return cell_box(g, p);
    const BatchPlace b = g.place[sid];
    return {b.cx + cell_coord(p.x, b.ox, g.inv, b.nx), b.cy + cell_coord(p.y, b.oy, g.inv, b.ny), b.cz + cell_coord(p.z, b.oz, g.inv, b.nz)};
}

// ---- 27-cell stencil, flattened over the lanes of one wavefront ------------------------------
// Lane r < 9 owns one (dy, dz) row of the stencil: three x-neighbour cells are contiguous in
// the cell-sorted array, so a row is one run [js, js+len).  All 18 bounds are fetched in one
// load round; candidate k of the concatenated runs is then mapped to a sorted position.
struct Stencil {
    int js[9];
    int pre[10];  // pre[r] = candidates before row r; pre[9] = total
};
__device__ __forceinline__ Stencil stencil_load(const GridDesc& g, const int* __restrict__ start, CellBox cb, int lane) {
    int my_js = 0, my_len = 0;
    if (lane < 9) {
        const int y2 = cb.cy + (lane % 3) - 1, z2 = cb.cz + (lane / 3) - 1;
        const int xlo = max(cb.cx - 1, 0), xhi = min(cb.cx + 1, g.nx - 1);
        if (y2 >= 0 && y2 < g.ny && z2 >= 0 && z2 < g.nz && xlo <= xhi) {
            const int rowbase = (z2 * g.ny + y2) * g.nx;
            my_js = start[rowbase + xlo];
            my_len = start[rowbase + xhi + 1] - my_js;
        }
    }
    Stencil st;
    st.pre[0] = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        st.js[r] = __builtin_amdgcn_readlane(my_js, r);
        st.pre[r + 1] = st.pre[r] + __builtin_amdgcn_readlane(my_len, r);
    }
    return st;
}
__device__ __forceinline__ int stencil_pos(const Stencil& st, int k) {
    int pos = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r)
        if (k >= st.pre[r] && k < st.pre[r + 1]) pos = st.js[r] + (k - st.pre[r]);
    return pos;
}
// Output records of the ring / amide kernels go through a per-wave LDS queue and leave with ONE atomicAdd on the
// bag counter per flush (same-address atomics run at ~90 per microsecond on this chip: one per ring made the
// 10 k-ring set of BASELINE configs[4] a 119 us kernel).  A record is two ids, up to four values and three bytes.
struct __attribute__((aligned(16))) PlaneRec {
    int i0, i1;
    unsigned u;      // u0 | u1 << 8 | u2 << 16
    unsigned pad;
    double d0, d1, d2, d3;
};
#define PLANE_QCAP 128
struct PlaneQueue {
    PlaneRec* q;   // this wave's PLANE_QCAP records of LDS
    int n;
    __device__ __forceinline__ void push(bool emit, const PlaneRec& r, int lane) {
        const unsigned long long me = __ballot(emit);
        if (!me) return;
        if (emit) q[n + __popcll(me & ((1ull << lane) - 1ull))] = r;
        n += __popcll(me);
    }
    __device__ __forceinline__ bool nearly_full() const { return n > PLANE_QCAP - 64; }
    template <class W>
    __device__ __forceinline__ void flush(u64* __restrict__ n_out, long long cap, int lane, W write) {
        if (n == 0) return;
        __builtin_amdgcn_wave_barrier();
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(n_out, (unsigned long long)n);
        base = __shfl(base, 0);
        for (int k = lane; k < n; k += 64)
            if ((long long)(base + k) < cap) write((long long)(base + k), q[k]);
        __builtin_amdgcn_wave_barrier();
        n = 0;
    }
    // end of the kernel: the four waves of the block leave with one atomicAdd (every wave of the block calls this)
    template <class W>
    __device__ __forceinline__ void flush_block(int* s_n, u64* s_base, u64* __restrict__ n_out, long long cap, int lane, W write) {
        const int w = threadIdx.x >> 6;
        if (lane == 0) s_n[w] = n;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int tot = s_n[0] + s_n[1] + s_n[2] + s_n[3];
            *s_base = tot ? atomicAdd(n_out, (unsigned long long)tot) : 0ull;
        }
        __syncthreads();
        unsigned long long base = *s_base;
        for (int k = 0; k < w; ++k) base += (unsigned long long)s_n[k];
        for (int k = lane; k < n; k += 64)
            if ((long long)(base + k) < cap) write((long long)(base + k), q[k]);
        n = 0;
    }
};
#define PAIR_QCAP 192
struct PlaneShared {   // LDS of one 256-thread block of a ring / amide kernel
    PlaneRec q[4][PLANE_QCAP];
    int2 pq[4][PAIR_QCAP];     // candidate pairs that passed the cheap distance pre-filter, waiting for a full wave
    u64 base;
    int n[4];
};

// Two-phase evaluation.  The geometry of a ring / amide pair is float64 acos / sqrt / division chains — hundreds of
// instructions — while an item has only a handful of partners inside its stencil, so evaluating in place keeps a few
// lanes of the wave busy.  Phase 1 therefore only enumerates: every lane tests one candidate with a cheap squared
// distance (a superset of the reference's cut-off test) and the survivors are compacted into a per-wave LDS queue of
// {home, partner} ids across items; phase 2 runs whenever 64 pairs have gathered (and once at the end) with one pair
// per lane, full lanes, and applies the reference's exact sequence.
struct PairQueue {
    int2* q;
    int n;
    template <class Eval>
    __device__ __forceinline__ void push(bool ok, int x, int y, int lane, Eval eval) {
        const unsigned long long m = __ballot(ok);
        if (!m) return;
        if (ok) q[n + __popcll(m & ((1ull << lane) - 1ull))] = make_int2(x, y);
        n += __popcll(m);
        __builtin_amdgcn_wave_barrier();
        if (n >= 64) {
            n -= 64;
            const int2 pr = q[n + lane];
            __builtin_amdgcn_wave_barrier();
            eval(true, pr.x, pr.y);
        }
    }
    template <class Eval>
    __device__ __forceinline__ void drain(int lane, Eval eval) {
        __builtin_amdgcn_wave_barrier();
        if (n > 0) {   // (wave-uniform)
            const bool live = lane < n;
            const int2 pr = live ? q[lane] : make_int2(0, 0);
            eval(live, pr.x, pr.y);
        }
        n = 0;
    }
};

// =====================================================================================================================
// Argument blocks of the four loops
// =====================================================================================================================
struct AtomPlaneArgs {          // __calculate_atom_plane_contacts, I:947-1062
    GridDesc g;                 // atom grid the candidates come from (all-atom 6 A grid, or the contact grid of the pass)
    const int* start;
    const float4* s_xyzm;
    const int4* s_aux;
    int nring;
    const double* ring_c;
    const double* ring_n;
    const int* ring_res;
    const uint8_t* ring_sel;
    const uint8_t* ring_plus;
    const uint8_t* plus;
    const uint8_t* ring_home;
    const int* ring_gid;
    const int* gid;
    long long cap;
    int* out_atom;
    int* out_ring;
    double* out_dist;
    double* out_theta;
    uint8_t* out_mask;
    uint8_t* out_ct;
    u64* n_out;
    // list mode (atoms addressed by local id through the static columns instead of a sorted position)
    const float4* st_xyzm;
    const uint8_t* sel;
    int sel_all;
};
struct PlanePlaneArgs {         // __calculate_plane_plane_contacts, I:1064-1194
    GridDesc g;
    const int* start;
    const int* perm;
    int nring;
    const double* ring_c;
    const double* ring_n;
    const int* ring_res;
    const uint8_t* ring_sel;
    const uint8_t* ring_plus;
    const uint8_t* ring_home;
    const int* ring_gid;
    long long cap;
    int* out_bgn;
    int* out_end;
    double* out_dist;
    double* out_dih;
    double* out_t1;
    double* out_t2;
    uint8_t* out_y1;
    uint8_t* out_y2;
    uint8_t* out_ct;
    u64* n_out;
};
struct GroupGroupArgs {         // __calculate_group_group_contacts, I:1217-1300
    GridDesc g;
    const int* start;
    const int* perm;
    int namide;
    const float* am_c;
    const float* am_n;
    const uint8_t* am_sel;
    const uint8_t* am_plus;
    const uint8_t* am_home;
    const int* am_gid;
    long long cap;
    int* out_bgn;
    int* out_end;
    float* out_dist;
    float* out_dih;
    float* out_theta;
    uint8_t* out_ct;
    u64* n_out;
};
struct GroupPlaneArgs {         // __calculate_group_plane_contacts, I:1302-1382
    GridDesc g;                 // RING grid
    const int* start;
    const int* perm;
    int namide;
    const float* am_c;
    const float* am_n;
    const uint8_t* am_sel;
    const uint8_t* am_plus;
    const double* ring_c;
    const double* ring_n;
    const uint8_t* ring_sel;
    const uint8_t* ring_plus;
    const uint8_t* am_home;
    const int* am_gid;
    const int* ring_gid;
    long long cap;
    int* out_amide;
    int* out_ring;
    double* out_dist;
    double* out_dih;
    double* out_theta;
    uint8_t* out_ct;
    u64* n_out;
};

// =====================================================================================================================
// Phase 2 of every loop: the reference's exact operation sequence for ONE candidate pair per lane, record into the
// wave's output queue.  `live` = this lane holds a pair that passed the per-pass filters.
// =====================================================================================================================
struct ApWrite {
    const AtomPlaneArgs& A;
    __device__ __forceinline__ void operator()(long long slot, const PlaneRec& t) const {
        A.out_atom[slot] = t.i0; A.out_ring[slot] = t.i1;
        A.out_dist[slot] = t.d0; A.out_theta[slot] = t.d1;
        A.out_mask[slot] = (uint8_t)(t.u & 255u); A.out_ct[slot] = (uint8_t)((t.u >> 8) & 255u);
    }
};
// ring r against the atom (x, y, z, meta) with local id lid, already inside the tree radius (I:960)
__device__ __forceinline__ void ap_eval(const AtomPlaneArgs& A, bool live, int r, float4 v, int lid, PlaneQueue& Q, int lane) {
    bool emit = false;
    PlaneRec rec;
    if (live) {
        const num::d3 ctr_ = ld3(A.ring_c, r), nrm = ld3(A.ring_n, r);
        const num::d3 x = {(double)v.x, (double)v.y, (double)v.z};
        const uint32_t m = __float_as_uint(v.w);
        const double dist = num::norm(num::sub(x, ctr_));                        // I:972
        const int ct = plane_ctype(A.ring_sel[r], m & M_SEL, true, true);         // I:985-997
        const double theta = num::group_angle(nrm, num::sub(ctr_, x));           // I:1005
        uint32_t mask = 0;
        if (dist <= 4.5 && theta <= 30.0) {                                      // I:1007
            if ((m & M_ELEM_C) && (m & ARP_T_WEAK_HBOND_DONOR)) mask |= ARP_AP_CARBONPI;
            if (m & ARP_T_POS_IONISABLE) mask |= ARP_AP_CATIONPI;
            if (m & ARP_T_HBOND_DONOR) mask |= ARP_AP_DONORPI;
            if (m & ARP_T_XBOND_DONOR) mask |= ARP_AP_HALOGENPI;
        }
        if (dist <= 6.0) {                                                       // I:1021
            if ((m & M_RES_MET) && (m & M_ELEM_S)) mask |= ARP_AP_METSULPHURPI;
        }
        emit = mask != 0;                                                        // I:1026
        rec.i0 = A.gid ? A.gid[lid] : lid;
        rec.i1 = A.ring_gid ? A.ring_gid[r] : r;
        rec.d0 = dist; rec.d1 = theta;
        rec.u = mask | ((unsigned)ct << 8);
    }
    if (Q.nearly_full()) Q.flush(A.n_out, A.cap, lane, ApWrite{A});
    Q.push(emit, rec, lane);
}

struct PpWrite {
    const PlanePlaneArgs& A;
    __device__ __forceinline__ void operator()(long long slot, const PlaneRec& t) const {
        A.out_bgn[slot] = t.i0; A.out_end[slot] = t.i1;
        A.out_dist[slot] = t.d0; A.out_dih[slot] = t.d1; A.out_t1[slot] = t.d2; A.out_t2[slot] = t.d3;
        A.out_y1[slot] = (uint8_t)(t.u & 255u); A.out_y2[slot] = (uint8_t)((t.u >> 8) & 255u);
        A.out_ct[slot] = (uint8_t)((t.u >> 16) & 255u);
    }
};
// rings a < b: reproduces both visits (a,b) and (b,a) of the reference's ordered double loop and its dedupe (I:1181-1194)
__device__ __forceinline__ void pp_eval(const PlanePlaneArgs& A, bool live, int a, int b, PlaneQueue& Q, int lane) {
    bool emit = false;
    PlaneRec rec;
    if (live) {
        const num::d3 ca = ld3(A.ring_c, a), na = ld3(A.ring_n, a), cbv = ld3(A.ring_c, b), nb = ld3(A.ring_n, b);
        const num::d3 pab = num::sub(ca, cbv);
        const double dist = num::norm(pab);                 // I:1111 (same value for both visits)
        if (!(dist > 6.0)) {                                // I:1113
            const bool intra = A.ring_res[a] == A.ring_res[b];  // I:1091
            const int ct = plane_ctype(A.ring_sel[a], A.ring_sel[b], true, true);
            const double cosd = num::dot(na, nb) / (num::norm(na) * num::norm(nb));
            const double dih = num::fold_deg(acos(cosd));                       // I:1122
            const double t_ab = num::group_angle(na, pab);                      // I:1123, visit (a,b)
            const double t_ba = num::group_angle(nb, num::sub(cbv, ca));        // visit (b,a)
            const int y_ab = num::pp_class(dih, t_ab), y_ba = num::pp_class(dih, t_ba);
            const bool skip_ab = intra && y_ab == ARP_PP_EE;  // I:1154
            const bool skip_ba = intra && y_ba == ARP_PP_EE;
            emit = !(skip_ab && skip_ba);
            const bool first = !skip_ab;
            double t1, t2;
            int y1, y2;
            if (first) {  // record created by visit (a,b); visit (b,a) may append its class
                t1 = t_ab; t2 = skip_ba ? NAN : t_ba;
                y1 = y_ab; y2 = skip_ba ? ARP_PP_SKIPPED : (y_ba == y_ab ? ARP_PP_SAME : y_ba);
            } else {      // first visit skipped: the reverse visit creates the record
                t1 = t_ba; t2 = NAN;
                y1 = y_ba; y2 = ARP_PP_SKIPPED;
            }
            const int ga = A.ring_gid ? A.ring_gid[a] : a, gb = A.ring_gid ? A.ring_gid[b] : b;
            rec.i0 = first ? ga : gb;
            rec.i1 = first ? gb : ga;
            rec.d0 = dist; rec.d1 = dih; rec.d2 = t1; rec.d3 = t2;
            rec.u = (unsigned)y1 | ((unsigned)y2 << 8) | ((unsigned)ct << 16);
        }
    }
    if (Q.nearly_full()) Q.flush(A.n_out, A.cap, lane, PpWrite{A});
    Q.push(emit, rec, lane);
}

struct GgWrite {
    const GroupGroupArgs& A;
    __device__ __forceinline__ void operator()(long long slot, const PlaneRec& t) const {   // float values travel as doubles (exact both ways)
        A.out_bgn[slot] = t.i0; A.out_end[slot] = t.i1;
        A.out_dist[slot] = (float)t.d0; A.out_dih[slot] = (float)t.d1; A.out_theta[slot] = (float)t.d2;
        A.out_ct[slot] = (uint8_t)(t.u & 255u);
    }
};
// ordered pair of amides, float32 (I:1227-1300)
__device__ __forceinline__ void gg_eval(const GroupGroupArgs& A, bool live, int a, int b, PlaneQueue& Q, int lane) {
    bool emit = false;
    PlaneRec rec;
    if (live) {
        const num::f3 ca = lf3(A.am_c, a), na = lf3(A.am_n, a), cbv = lf3(A.am_c, b), nb = lf3(A.am_n, b);
        const num::f3 pab = num::sub(ca, cbv);
        const float dist = num::norm(pab);                 // I:1268
        if (!(dist > (float)6.0)) {                        // I:1270
            const float cosd = num::dot(na, nb) / (num::norm(na) * num::norm(nb));
            const float dih = num::fold_deg(acosf(cosd));  // I:1278
            const float theta = num::group_angle(na, pab); // I:1279
            emit = !(dih > 30.0f || theta > 30.0f);        // I:1282
            rec.i0 = A.am_gid ? A.am_gid[a] : a; rec.i1 = A.am_gid ? A.am_gid[b] : b;
            rec.d0 = (double)dist; rec.d1 = (double)dih; rec.d2 = (double)theta;
            rec.u = (unsigned)plane_ctype(A.am_sel[a], A.am_sel[b], true, true);
        }
    }
    if (Q.nearly_full()) Q.flush(A.n_out, A.cap, lane, GgWrite{A});
    Q.push(emit, rec, lane);
}

struct GpWrite {
    const GroupPlaneArgs& A;
    __device__ __forceinline__ void operator()(long long slot, const PlaneRec& t) const {
        A.out_amide[slot] = t.i0; A.out_ring[slot] = t.i1;
        A.out_dist[slot] = t.d0; A.out_dih[slot] = t.d1; A.out_theta[slot] = t.d2;
        A.out_ct[slot] = (uint8_t)(t.u & 255u);
    }
};
// amide a, ring r (I:1312-1382)
__device__ __forceinline__ void gp_eval(const GroupPlaneArgs& A, bool live, int a, int r, PlaneQueue& Q, int lane) {
    bool emit = false;
    PlaneRec rec;
    if (live) {
        const num::f3 ca = lf3(A.am_c, a), na = lf3(A.am_n, a);
        const num::d3 cad = num::to_d3(ca);
        const num::d3 cr = ld3(A.ring_c, r), nr = ld3(A.ring_n, r);
        const num::d3 par = num::sub(cad, cr);
        const double dist = num::norm(par);         // I:1349
        if (!(dist > 6.0)) {                        // I:1351
            const double cosd = num::dot(num::to_d3(na), nr) / ((double)num::norm(na) * num::norm(nr));
            const double dih = num::fold_deg(acos(cosd));    // I:1359
            const double theta = num::group_angle(na, par);  // I:1360
            emit = !(dih > 30.0 || theta > 30.0);            // I:1363
            rec.i0 = A.am_gid ? A.am_gid[a] : a; rec.i1 = A.ring_gid ? A.ring_gid[r] : r;
            rec.d0 = dist; rec.d1 = dih; rec.d2 = theta;
            rec.u = (unsigned)plane_ctype(A.am_sel[a], A.ring_sel[r], true, true);
        }
    }
    if (Q.nearly_full()) Q.flush(A.n_out, A.cap, lane, GpWrite{A});
    Q.push(emit, rec, lane);
}

// =====================================================================================================================
// Phase 1: enumeration of the candidates of one home item (a wavefront per item, lanes over the stencil).  `Sink` receives
// (ok, x, y) for every lane; DYNAMIC = also apply the per-pass filters (selection_plus membership, ownership), which the
// static candidate lists leave to the pass.
// =====================================================================================================================
template <bool DYNAMIC, class Sink>
__device__ __forceinline__ void ap_enumerate(const AtomPlaneArgs& A, int r, int lane, Sink sink) {   // 27-cell stencil of a >= 6 A atom grid
    if (DYNAMIC && (!A.ring_plus[r] || (A.ring_home && !A.ring_home[r]))) return;   // I:957; multi-GPU: the ring's owner emits
    const num::d3 ctr_ = ld3(A.ring_c, r);
    const Stencil st = stencil_load(A.g, A.start, cell_box(A.g, ctr_, A.g.place ? A.g.sid_ring[r] : 0), lane);
    for (int kb = 0; kb < st.pre[9]; kb += 64) {
        const int k = kb + lane;
        bool ok = false;
        int j = 0;
        if (k < st.pre[9]) {
            j = stencil_pos(st, k);
            const float4 v = A.s_xyzm[j];
            const uint32_t m = __float_as_uint(v.w);
            // I:960 tree membership (float64, inclusive), I:964 hydrogens, I:975 aromatic atoms, I:968 selection_plus
            ok = num::dist2_kd(ctr_, num::d3{(double)v.x, (double)v.y, (double)v.z}) <= 36.0 && !(m & (M_HYDROGEN | ARP_T_AROMATIC));
            if (DYNAMIC) ok = ok && A.plus[A.s_aux[j].x];
        }
        sink(ok, r, j);
    }
}
// the same on the CONTACT grid of the pass (cell edge = the interacting cut-off, atoms of selection_plus without
// hydrogens — exactly the atoms I:964-968 let through): with a 5 A edge the 6 A query reaches two cells, so the stencil is
// (2R + 1)^2 rows of 2R + 1 contiguous cells, R = floor(6 / edge) + 1; lane r fetches the bounds of row r, the rows are
// swept one after the other
template <class Sink>
__device__ __forceinline__ void ap_enumerate_cg(const AtomPlaneArgs& A, int r, int lane, Sink sink) {
    if (!A.ring_plus[r] || (A.ring_home && !A.ring_home[r])) return;
    const GridDesc g = A.g;
    const int R = (int)floor(6.0 * g.inv) + 1, W = 2 * R + 1, nrows = W * W;
    const num::d3 ctr_ = ld3(A.ring_c, r);
    const CellBox cb = cell_box(g, ctr_, g.place ? g.sid_ring[r] : 0);
    for (int row0 = 0; row0 < nrows; row0 += 64) {
        int my_js = 0, my_len = 0;
        const int row = row0 + lane;
        if (row < nrows) {
            const int y2 = cb.cy + (row % W) - R, z2 = cb.cz + (row / W) - R;
            const int xlo = max(cb.cx - R, 0), xhi = min(cb.cx + R, g.nx - 1);
            if (y2 >= 0 && y2 < g.ny && z2 >= 0 && z2 < g.nz && xlo <= xhi) {
                const int rowbase = (z2 * g.ny + y2) * g.nx;
                my_js = A.start[rowbase + xlo];
                my_len = A.start[rowbase + xhi + 1] - my_js;
            }
        }
        const int rows_here = min(64, nrows - row0);
        for (int rr = 0; rr < rows_here; ++rr) {
            const int js = __builtin_amdgcn_readlane(my_js, rr), len = __builtin_amdgcn_readlane(my_len, rr);
            for (int kb = 0; kb < len; kb += 64) {
                const int k = kb + lane;
                bool ok = false;
                if (k < len) {
                    const float4 v = A.s_xyzm[js + k];
                    ok = num::dist2_kd(ctr_, num::d3{(double)v.x, (double)v.y, (double)v.z}) <= 36.0 &&
                         !(__float_as_uint(v.w) & ARP_T_AROMATIC);
                }
                sink(ok, r, js + k);
            }
        }
    }
}
// The static candidate list of the atom-plane loop, ATOM by atom: every lane takes one atom of the structure — as uploaded, by
// local id: no atom grid, no spatial order of the atoms is needed — and walks the 27-cell stencil of the 6 A grid of ring CENTRES
// around it (I:960 is symmetric in the two points).  What it reads exists as soon as the structure and the centre grids do, so the
// list is made with the upload, not by the first pass; a structure with few rings costs its atoms nine pairs of start-table words.
// The wave walks its 64 stencils together: row by row, entry t of every lane's row at once (the sink is a wave-wide affair).
struct AtomRingListArgs {
    GridDesc g;                 // 6 A grid of the ring centres
    const int* start;
    const int* perm;
    int n;                      // atoms
    const float4* xyz;          // by local id, w unused
    const uint16_t* tmask;
    const uint16_t* flags;
    const double* ring_c;
};
template <class Sink>
__device__ __forceinline__ void ar_enumerate(const AtomRingListArgs& A, int i, int lane, Sink sink) {
    const bool have = i < A.n;
    const float4 v = have ? A.xyz[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t m = have ? ((uint32_t)(A.tmask[i] & M_TMASK) | ((uint32_t)(A.flags[i] & 0x7F) << M_FLAG_SHIFT)) : M_HYDROGEN;
    const bool cand = have && !(m & (M_HYDROGEN | ARP_T_AROMATIC));       // I:964 hydrogens, I:975 aromatic atoms
    if (!__any(cand)) return;
    const GridDesc g = A.g;
    const num::d3 p = {(double)v.x, (double)v.y, (double)v.z};
    const CellBox cb = cand ? cell_box(g, p, g.place ? g.sid_atom[i] : 0) : CellBox{0, 0, 0};
    const int xlo = max(cb.cx - 1, 0), xhi = min(cb.cx + 1, g.nx - 1);
    // the bounds of the nine rows in ONE load round (eighteen loads in flight), then the rows one after the other
    int js[9], len[9];
#pragma unroll
    for (int row = 0; row < 9; ++row) {
        const int y2 = cb.cy + (row % 3) - 1, z2 = cb.cz + (row / 3) - 1;
        const bool in = cand && y2 >= 0 && y2 < g.ny && z2 >= 0 && z2 < g.nz && xlo <= xhi;
        const int rowbase = in ? (z2 * g.ny + y2) * g.nx : 0;
        const int a = A.start[rowbase + (in ? xlo : 0)], b = A.start[rowbase + (in ? xhi + 1 : 0)];
        js[row] = a;
        len[row] = b - a;
    }
    int total = 0;
#pragma unroll
    for (int row = 0; row < 9; ++row) total += len[row];
    if (!__any(total > 0)) return;
#pragma unroll
    for (int row = 0; row < 9; ++row) {
        for (int t = 0; __any(t < len[row]); ++t) {
            bool ok = false;
            int r = 0;
            if (t < len[row]) {
                r = A.perm[js[row] + t];
                ok = num::dist2_kd(ld3(A.ring_c, r), p) <= 36.0;      // I:960 tree membership (float64, inclusive)
            }
            sink(ok, r, i);
        }
    }
}
template <bool DYNAMIC, class Sink>
__device__ __forceinline__ void pp_enumerate(const PlanePlaneArgs& A, int a, int lane, Sink sink) {
    if (DYNAMIC && (!A.ring_plus[a] || (A.ring_home && !A.ring_home[a]))) return;   // I:1081; multi-GPU: owner of the lower ring id emits
    const num::d3 ca = ld3(A.ring_c, a);
    const Stencil st = stencil_load(A.g, A.start, cell_box(A.g, ca, A.g.place ? A.g.sid_ring[a] : 0), lane);
    for (int kb = 0; kb < st.pre[9]; kb += 64) {
        const int k = kb + lane;
        bool ok = false;
        int b = 0;
        if (k < st.pre[9]) {
            b = A.perm[stencil_pos(st, k)];
            // unordered pair once (I:1081, 1085); squared-distance superset of I:1113, settled exactly in pp_eval
            ok = b > a && num::dist2_kd(ca, ld3(A.ring_c, b)) <= 36.0 * (1.0 + 1e-9);
            if (DYNAMIC) ok = ok && A.ring_plus[b];
        }
        sink(ok, a, b);
    }
}
template <bool DYNAMIC, class Sink>
__device__ __forceinline__ void gg_enumerate(const GroupGroupArgs& A, int a, int lane, Sink sink) {
    if (DYNAMIC && (!A.am_plus[a] || (A.am_home && !A.am_home[a]))) return;   // multi-GPU: owner of the bgn amide emits
    const num::d3 cad = num::to_d3(lf3(A.am_c, a));
    const Stencil st = stencil_load(A.g, A.start, cell_box(A.g, cad, A.g.place ? A.g.sid_amide[a] : 0), lane);
    for (int kb = 0; kb < st.pre[9]; kb += 64) {
        const int k = kb + lane;
        bool ok = false;
        int b = 0;
        if (k < st.pre[9]) {
            b = A.perm[stencil_pos(st, k)];
            // I:1233, 1237; float64 squared-distance superset of the float32 test at I:1270 (settled exactly in gg_eval)
            ok = b != a && num::dist2_kd(cad, num::to_d3(lf3(A.am_c, b))) <= 36.0 * (1.0 + 1e-5);
            if (DYNAMIC) ok = ok && A.am_plus[b];
        }
        sink(ok, a, b);
    }
}
template <bool DYNAMIC, class Sink>
__device__ __forceinline__ void gp_enumerate(const GroupPlaneArgs& A, int a, int lane, Sink sink) {
    if (DYNAMIC && (!A.am_plus[a] || (A.am_home && !A.am_home[a]))) return;   // multi-GPU: owner of the amide emits
    const num::d3 cad = num::to_d3(lf3(A.am_c, a));
    const Stencil st = stencil_load(A.g, A.start, cell_box(A.g, cad, A.g.place ? A.g.sid_amide[a] : 0), lane);
    for (int kb = 0; kb < st.pre[9]; kb += 64) {
        const int k = kb + lane;
        bool ok = false;
        int r = 0;
        if (k < st.pre[9]) {
            r = A.perm[stencil_pos(st, k)];
            ok = num::dist2_kd(cad, ld3(A.ring_c, r)) <= 36.0 * (1.0 + 1e-9);   // superset of I:1351
            if (DYNAMIC) ok = ok && A.ring_plus[r];                              // I:1318
        }
        sink(ok, a, r);
    }
}

// =====================================================================================================================
// Direct form: enumerate and evaluate in one kernel (standalone entry points, sharded stage path)
// =====================================================================================================================
__device__ __forceinline__ void atom_plane_body(const AtomPlaneArgs& A, int vblock, int vgrid, PlaneShared* sh, bool contact_grid) {
    PlaneQueue Q{sh->q[threadIdx.x >> 6], 0};
    PairQueue P{sh->pq[threadIdx.x >> 6], 0};
    const int lane = threadIdx.x & 63;
    auto eval = [&](bool live, int r, int j) {
        const float4 v = live ? A.s_xyzm[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        ap_eval(A, live, r, v, live ? A.s_aux[j].x : 0, Q, lane);
    };
    auto sink = [&](bool ok, int r, int j) { P.push(ok, r, j, lane, eval); };
    const int wave = (vblock * blockDim.x + threadIdx.x) >> 6, nwave = (vgrid * blockDim.x) >> 6;
    for (int r = wave; r < A.nring; r += nwave) {
        if (contact_grid) ap_enumerate_cg(A, r, lane, sink);
        else ap_enumerate<true>(A, r, lane, sink);
    }
    P.drain(lane, eval);
    Q.flush_block(sh->n, &sh->base, A.n_out, A.cap, lane, ApWrite{A});
}
__device__ __forceinline__ void plane_plane_body(const PlanePlaneArgs& A, int vblock, int vgrid, PlaneShared* sh) {
    PlaneQueue Q{sh->q[threadIdx.x >> 6], 0};
    PairQueue P{sh->pq[threadIdx.x >> 6], 0};
    const int lane = threadIdx.x & 63;
    auto eval = [&](bool live, int a, int b) { pp_eval(A, live, a, b, Q, lane); };
    auto sink = [&](bool ok, int a, int b) { P.push(ok, a, b, lane, eval); };
    const int wave = (vblock * blockDim.x + threadIdx.x) >> 6, nwave = (vgrid * blockDim.x) >> 6;
    for (int a = wave; a < A.nring; a += nwave) pp_enumerate<true>(A, a, lane, sink);
    P.drain(lane, eval);
    Q.flush_block(sh->n, &sh->base, A.n_out, A.cap, lane, PpWrite{A});
}
__device__ __forceinline__ void group_group_body(const GroupGroupArgs& A, int vblock, int vgrid, PlaneShared* sh) {
    PlaneQueue Q{sh->q[threadIdx.x >> 6], 0};
    PairQueue P{sh->pq[threadIdx.x >> 6], 0};
    const int lane = threadIdx.x & 63;
    auto eval = [&](bool live, int a, int b) { gg_eval(A, live, a, b, Q, lane); };
    auto sink = [&](bool ok, int a, int b) { P.push(ok, a, b, lane, eval); };
    const int wave = (vblock * blockDim.x + threadIdx.x) >> 6, nwave = (vgrid * blockDim.x) >> 6;
    for (int a = wave; a < A.namide; a += nwave) gg_enumerate<true>(A, a, lane, sink);
    P.drain(lane, eval);
    Q.flush_block(sh->n, &sh->base, A.n_out, A.cap, lane, GgWrite{A});
}
__device__ __forceinline__ void group_plane_body(const GroupPlaneArgs& A, int vblock, int vgrid, PlaneShared* sh) {
    PlaneQueue Q{sh->q[threadIdx.x >> 6], 0};
    PairQueue P{sh->pq[threadIdx.x >> 6], 0};
    const int lane = threadIdx.x & 63;
    auto eval = [&](bool live, int a, int r) { gp_eval(A, live, a, r, Q, lane); };
    auto sink = [&](bool ok, int a, int r) { P.push(ok, a, r, lane, eval); };
    const int wave = (vblock * blockDim.x + threadIdx.x) >> 6, nwave = (vgrid * blockDim.x) >> 6;
    for (int a = wave; a < A.namide; a += nwave) gp_enumerate<true>(A, a, lane, sink);
    P.drain(lane, eval);
    Q.flush_block(sh->n, &sh->base, A.n_out, A.cap, lane, GpWrite{A});
}

__global__ __launch_bounds__(256) void k_atom_plane(AtomPlaneArgs a) {
    __shared__ PlaneShared s_sh;
    atom_plane_body(a, (int)blockIdx.x, (int)gridDim.x, &s_sh, false);
}
__global__ __launch_bounds__(256) void k_plane_plane(PlanePlaneArgs a) {
    __shared__ PlaneShared s_sh;
    plane_plane_body(a, (int)blockIdx.x, (int)gridDim.x, &s_sh);
}
__global__ __launch_bounds__(256) void k_group_group(GroupGroupArgs a) {
    __shared__ PlaneShared s_sh;
    group_group_body(a, (int)blockIdx.x, (int)gridDim.x, &s_sh);
}
__global__ __launch_bounds__(256) void k_group_plane(GroupPlaneArgs a) {
    __shared__ PlaneShared s_sh;
    group_plane_body(a, (int)blockIdx.x, (int)gridDim.x, &s_sh);
}

// =====================================================================================================================
// Static candidate lists.  Which ring / amide / atom pairs lie within 6 A of each other depends on the coordinates only;
// it is established ONCE per uploaded structure (k_plane_lists: the four enumerations above without the per-pass filters,
// appended to four {x, y} lists in HBM) — a spatial index like the ring grid it is built from.  A pass then spends no
// time walking stencils: it reads the lists, 64 pairs per wavefront, applies the filters of the pass (selection_plus
// membership of both partners, ownership) and evaluates the geometry with full lanes (planes_from_lists).
// =====================================================================================================================
struct PlaneLists {
    int2* pairs[4];            // 0 atom-plane {ring, atom local id}, 1 plane-plane {a, b}, 2 group-group {a, b}, 3 group-plane {amide, ring}
    long long cap[4];
    u64* count;                // [4], device; may exceed cap (overflow: the host re-sizes and rebuilds)
};
// Appends the lanes with ok to list k.  The entries of a block gather in LDS (slots from an LDS counter) and leave with
// ONE global atomicAdd when the block ends (flush_list): all the waves of the launch adding to the same four counters
// were bound by the same-address atomic rate (~90 per microsecond), 74 us for config 3.  A block that overflows its LDS
// queue appends the rest directly (one global atomicAdd per wave and call).
#define LIST_QCAP 1024
struct ListQueue { int2 q[LIST_QCAP]; int n, cut; unsigned long long base; };   // cut: first slot that was NOT written to LDS
struct ListSink {
    const PlaneLists& L;
    int k, lane;
    ListQueue* Q;
    __device__ __forceinline__ void operator()(bool ok, int x, int y) const {
        const unsigned long long m = __ballot(ok);
        if (!m) return;
        int slot0 = 0;
        if (lane == 0) slot0 = atomicAdd(&Q->n, __popcll(m));
        slot0 = __shfl(slot0, 0);
        const int slot = slot0 + __popcll(m & ((1ull << lane) - 1ull));
        if (slot0 + __popcll(m) <= LIST_QCAP) {
            if (ok) Q->q[slot] = make_int2(x, y);
            return;
        }
        // overflow: this call's LDS slots stay unwritten (the queue is cut at the first such slot), its entries go to the list directly
        unsigned long long base = 0;
        if (lane == 0) {
            atomicMin(&Q->cut, slot0);
            base = atomicAdd(L.count + k, (unsigned long long)__popcll(m));
        }
        base = __shfl(base, 0);
        const long long g = (long long)base + __popcll(m & ((1ull << lane) - 1ull));
        if (ok && g < L.cap[k]) L.pairs[k][g] = make_int2(x, y);
    }
};
// (every thread of the block, after its enumeration)
__device__ __forceinline__ void flush_list(const PlaneLists& L, int k, ListQueue* Q) {
    __syncthreads();
    // slots are handed out in order, so everything below the first slot of the first overflowing call was written
    const int n = min(Q->n, Q->cut);
    if (threadIdx.x == 0) Q->base = n > 0 ? atomicAdd(L.count + k, (unsigned long long)n) : 0ull;
    __syncthreads();
    const unsigned long long base = Q->base;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const long long g = (long long)base + i;
        if (g < L.cap[k]) L.pairs[k][g] = Q->q[i];
    }
}
// blocks [0, nb0) atoms -> atom-plane list, [nb0, nb1) rings -> plane-plane, [nb1, nb2) amides -> group-group, rest -> group-plane
__global__ __launch_bounds__(256) void k_plane_lists(AtomRingListArgs ar, PlanePlaneArgs pp, GroupGroupArgs gg, GroupPlaneArgs gp,
                                                     PlaneLists L, int nb0, int nb1, int nb2, int nb3) {
    __shared__ ListQueue s_lq;
    if (threadIdx.x == 0) { s_lq.n = 0; s_lq.cut = LIST_QCAP; }
    __syncthreads();
    const int b = (int)blockIdx.x, lane = threadIdx.x & 63;
    int kind = 3;
    if (b < nb0) {
        kind = 0;
        const int wave = (b * 256 + (int)threadIdx.x) >> 6, nwave = nb0 * 4;
        for (int i0 = wave * 64; i0 < ar.n; i0 += nwave * 64) ar_enumerate(ar, i0 + lane, lane, ListSink{L, 0, lane, &s_lq});     // {ring, local atom id}
    } else if (b < nb1) {
        kind = 1;
        const int wave = ((b - nb0) * 256 + (int)threadIdx.x) >> 6, nwave = (nb1 - nb0) * 4;
        for (int a = wave; a < pp.nring; a += nwave) pp_enumerate<false>(pp, a, lane, ListSink{L, 1, lane, &s_lq});
    } else if (b < nb2) {
        kind = 2;
        const int wave = ((b - nb1) * 256 + (int)threadIdx.x) >> 6, nwave = (nb2 - nb1) * 4;
        for (int a = wave; a < gg.namide; a += nwave) gg_enumerate<false>(gg, a, lane, ListSink{L, 2, lane, &s_lq});
    } else if (b < nb3) {
        const int wave = ((b - nb2) * 256 + (int)threadIdx.x) >> 6, nwave = (nb3 - nb2) * 4;
        for (int a = wave; a < gp.namide; a += nwave) gp_enumerate<false>(gp, a, lane, ListSink{L, 3, lane, &s_lq});
    }
    flush_list(L, kind, &s_lq);
}

// The ring / amide loops of a pass from the lists: waves take chunks of 64 list entries (a chunk never straddles two
// lists), filter, evaluate.  vblock / vgrid: position among the blocks doing this work.
__device__ __forceinline__ void planes_from_lists(const AtomPlaneArgs& ap, const PlanePlaneArgs& pp, const GroupGroupArgs& gg,
                                                  const GroupPlaneArgs& gp, const PlaneLists& L, u64* __restrict__ publish_counts,
                                                  int vblock, int vgrid, PlaneShared* sh, int kinds = 15) {      // kinds: bit k = evaluate list k
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    long long cnt[4], chunks[5];
    chunks[0] = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const u64 c = L.count[k];
        cnt[k] = (long long)(c < (u64)L.cap[k] ? c : (u64)L.cap[k]);
        chunks[k + 1] = chunks[k] + (((kinds >> k) & 1) ? (cnt[k] + 63) / 64 : 0);
        // the host checks them against the capacities at the end of the pass.  An ATOMIC, like every other write to the counter
        // block: the block that publishes the counters (pass_end) may run on another XCD — or, on a structure's first pass, in
        // the other kernel — and reads the memory-side values; a plain store would sit in this XCD's L2 until the kernel ends,
        // be missed by the publisher and land on the zeroed counter afterwards (the next structure then saw this one's counts:
        // a spurious "list too small", now and then three times in a row)
        if (vblock == 0 && threadIdx.x == 0) atomicExch(publish_counts + k, c);
    }
    const long long wave = (long long)vblock * 4 + w, nwave = (long long)vgrid * 4;
    PlaneQueue Q{sh->q[w], 0};
    int kind_done = -1;     // output queue holds records of one kind at a time
    auto flush_kind = [&](int kind) {
        if (kind == 0) Q.flush(ap.n_out, ap.cap, lane, ApWrite{ap});
        else if (kind == 1) Q.flush(pp.n_out, pp.cap, lane, PpWrite{pp});
        else if (kind == 2) Q.flush(gg.n_out, gg.cap, lane, GgWrite{gg});
        else if (kind == 3) Q.flush(gp.n_out, gp.cap, lane, GpWrite{gp});
    };
    for (long long ch = wave; ch < chunks[4]; ch += nwave) {
        const int kind = ch < chunks[1] ? 0 : ch < chunks[2] ? 1 : ch < chunks[3] ? 2 : 3;
        if (kind != kind_done) { flush_kind(kind_done); kind_done = kind; }
        const long long e = (ch - chunks[kind]) * 64 + lane;
        const bool have = e < cnt[kind];
        const int2 pr = have ? L.pairs[kind][e] : make_int2(0, 0);
        if (kind == 0) {
            const int r = pr.x, lid = pr.y;
            bool live = have && ap.ring_plus[r] && !(ap.ring_home && !ap.ring_home[r]) && (ap.sel_all || ap.plus[lid]);   // I:957, 968
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) {
                v = ap.st_xyzm[lid];
                uint32_t m = __float_as_uint(v.w);
                if (ap.sel_all || (ap.sel && ap.sel[lid])) m |= M_SEL;
                v.w = __uint_as_float(m);
            }
            ap_eval(ap, live, r, v, lid, Q, lane);
        } else if (kind == 1) {
            const bool live = have && pp.ring_plus[pr.x] && pp.ring_plus[pr.y] && !(pp.ring_home && !pp.ring_home[pr.x]);
            pp_eval(pp, live, pr.x, pr.y, Q, lane);
        } else if (kind == 2) {
            const bool live = have && gg.am_plus[pr.x] && gg.am_plus[pr.y] && !(gg.am_home && !gg.am_home[pr.x]);
            gg_eval(gg, live, pr.x, pr.y, Q, lane);
        } else {
            const bool live = have && gp.am_plus[pr.x] && gp.ring_plus[pr.y] && !(gp.am_home && !gp.am_home[pr.x]);
            gp_eval(gp, live, pr.x, pr.y, Q, lane);
        }
    }
    flush_kind(kind_done);
}

// ---- the last launch of a pass: ring / amide loops (from the lists) and the per-pair SIFt kernel in ONE grid ------------
// Blocks [0, np) evaluate the candidate lists (np a multiple of 8, so that vblock % 8 of the sift blocks is still the XCD
// the dispatcher puts them on), blocks [np, np + nsift) are the sift kernel.  One stream, no cross-stream events (those
// cost the host ~5 us each); pass_end() publishes the counters of the pass.
union SiftPlanesShared {
    PlaneShared planes;
    SiftShared sift;
};
template <int STREAM, int GID = 0>
__global__ __launch_bounds__(256, SIFT_MIN_WAVES) void k_sift_planes(SiftArgs sa, int nsift, AtomPlaneArgs ap, PlanePlaneArgs pp,
                                                                     GroupGroupArgs gg, GroupPlaneArgs gp, PlaneLists L,
                                                                     u64* publish_counts, int np, PublishArgs pub) {
    __shared__ SiftPlanesShared s_sh;
    const int b = (int)blockIdx.x;
    if (b >= np) sift_body<STREAM, GID>(sa, b - np, nsift, &s_sh.sift);
    else planes_from_lists(ap, pp, gg, gp, L, publish_counts, b, np, &s_sh.planes);
    pass_end(pub, 0);
}
// the two halves as separate kernels (sharded stage path with a caller-owned stream, structures without atoms / planes)
__global__ __launch_bounds__(256) void k_planes(AtomPlaneArgs ap, PlanePlaneArgs pp, GroupGroupArgs gg, GroupPlaneArgs gp, PlaneLists L,
                                                u64* publish_counts, PublishArgs pub, int kinds) {
    __shared__ PlaneShared s_sh;
    planes_from_lists(ap, pp, gg, gp, L, publish_counts, (int)blockIdx.x, (int)gridDim.x, &s_sh, kinds);
    pass_end(pub, 1);
}
template <int STREAM, int GID = 0>
__global__ __launch_bounds__(256, SIFT_MIN_WAVES) void k_sift(SiftArgs sa, PublishArgs pub) {
    __shared__ SiftShared s_sh;
    sift_body<STREAM, GID>(sa, (int)blockIdx.x, (int)gridDim.x, &s_sh);
    pass_end(pub, 0);
}
