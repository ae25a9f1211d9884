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

// One wavefront per ring; lanes sweep the atoms of the 27 cells around the ring centre =
// NeighborSearch.search(center, 6.0) (I:960).  The grid is the all-atom 6 A grid of the
// selection expansion; membership of the selection_plus tree (I:1442) and the hydrogen
// filter (I:964) are applied per atom.
struct AtomPlaneArgs {
    GridDesc g;
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
};
__device__ __forceinline__ void atom_plane_body(GridDesc g, const int* __restrict__ start,
                                                    const float4* __restrict__ s_xyzm, const int4* __restrict__ s_aux,
                                                    int nring, const double* __restrict__ ring_c,
                                                    const double* __restrict__ ring_n, const int* __restrict__ ring_res,
                                                    const uint8_t* __restrict__ ring_sel, const uint8_t* __restrict__ ring_plus,
                                                    const uint8_t* __restrict__ plus, const uint8_t* __restrict__ ring_home,
                                                    const int* __restrict__ ring_gid,
                                                    const int* __restrict__ gid, long long cap, int* __restrict__ out_atom,
                                                    int* __restrict__ out_ring, double* __restrict__ out_dist,
                                                    double* __restrict__ out_theta, uint8_t* __restrict__ out_mask,
                                                    uint8_t* __restrict__ out_ct, u64* __restrict__ n_out, int vblock, int vgrid, PlaneShared* sh) {
    PlaneQueue Q{sh->q[threadIdx.x >> 6], 0};
    PairQueue P{sh->pq[threadIdx.x >> 6], 0};
    auto write = [&](long long slot, const PlaneRec& t) {
        out_atom[slot] = t.i0; out_ring[slot] = t.i1;
        out_dist[slot] = t.d0; out_theta[slot] = t.d1;
        out_mask[slot] = (uint8_t)(t.u & 255u); out_ct[slot] = (uint8_t)((t.u >> 8) & 255u);
    };
    const int lane = threadIdx.x & 63;
    auto eval = [&](bool live, int r, int j) {   // ring r, atom at sorted position j (inside the tree radius, I:960)
        bool emit = false;
        PlaneRec rec;
        if (live) {
            const num::d3 ctr_ = ld3(ring_c, r), nrm = ld3(ring_n, r);
            const float4 v = s_xyzm[j];
            const num::d3 x = {(double)v.x, (double)v.y, (double)v.z};
            const uint32_t m = __float_as_uint(v.w);
            const int lid = s_aux[j].x;
            const double dist = num::norm(num::sub(x, ctr_));                        // I:972
            const int ct = plane_ctype(ring_sel[r], m & M_SEL, true, true);           // I:985-997
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
            rec.i0 = gid ? gid[lid] : lid;
            rec.i1 = ring_gid ? ring_gid[r] : r;
            rec.d0 = dist; rec.d1 = theta;
            rec.u = mask | ((unsigned)ct << 8);
        }
        if (Q.nearly_full()) Q.flush(n_out, cap, lane, write);
        Q.push(emit, rec, lane);
    };
    const int wave = (vblock * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (vgrid * blockDim.x) >> 6;
    for (int r = wave; r < nring; r += nwave) {
        if (!ring_plus[r]) continue;  // I:957
        if (ring_home && !ring_home[r]) continue;  // multi-GPU: the rank owning the ring emits
        const num::d3 ctr_ = ld3(ring_c, r);
        const Stencil st = stencil_load(g, start, cell_box(g, ctr_), lane);
        for (int kb = 0; kb < st.pre[9]; kb += 64) {
            const int k = kb + lane;
            bool ok = false;
            int j = 0;
            if (k < st.pre[9]) {
                j = stencil_pos(st, k);
                const float4 v = s_xyzm[j];
                const uint32_t m = __float_as_uint(v.w);
                // I:960 tree membership (float64, inclusive), I:964 hydrogens, I:975 aromatic atoms, I:968 selection_plus
                ok = num::dist2_kd(ctr_, num::d3{(double)v.x, (double)v.y, (double)v.z}) <= 36.0 &&
                     !(m & (M_HYDROGEN | ARP_T_AROMATIC)) && plus[s_aux[j].x];
            }
            P.push(ok, r, j, lane, eval);
        }
    }
    P.drain(lane, eval);
    Q.flush_block(sh->n, &sh->base, n_out, cap, lane, write);
}

// One wavefront per ring a; lanes = partner rings b > a of the 27 cells around it.  Reproduces
// both visits (a,b) and (b,a) of the reference's ordered double loop and its dedupe (I:1181-1194).
struct PlanePlaneArgs {
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
__device__ __forceinline__ void plane_plane_body(GridDesc g, const int* __restrict__ start, const int* __restrict__ perm,
                                                     int nring, const double* __restrict__ ring_c,
                                                     const double* __restrict__ ring_n, const int* __restrict__ ring_res,
                                                     const uint8_t* __restrict__ ring_sel, const uint8_t* __restrict__ ring_plus,
                                                     const uint8_t* __restrict__ ring_home, const int* __restrict__ ring_gid,
                                                     long long cap, int* __restrict__ out_bgn, int* __restrict__ out_end,
                                                     double* __restrict__ out_dist, double* __restrict__ out_dih,
                                                     double* __restrict__ out_t1, double* __restrict__ out_t2,
                                                     uint8_t* __restrict__ out_y1, uint8_t* __restrict__ out_y2,
                                                     uint8_t* __restrict__ out_ct, u64* __restrict__ n_out, int vblock, int vgrid, PlaneShared* sh) {
    PlaneQueue Q{sh->q[threadIdx.x >> 6], 0};
    PairQueue P{sh->pq[threadIdx.x >> 6], 0};
    auto write = [&](long long slot, const PlaneRec& t) {
        out_bgn[slot] = t.i0; out_end[slot] = t.i1;
        out_dist[slot] = t.d0; out_dih[slot] = t.d1; out_t1[slot] = t.d2; out_t2[slot] = t.d3;
        out_y1[slot] = (uint8_t)(t.u & 255u); out_y2[slot] = (uint8_t)((t.u >> 8) & 255u);
        out_ct[slot] = (uint8_t)((t.u >> 16) & 255u);
    };
    const int lane = threadIdx.x & 63;
    auto eval = [&](bool live, int a, int b) {   // rings a < b, both in selection_plus, centres within ~6 A
        bool emit = false;
        PlaneRec rec;
        if (live) {
            const num::d3 ca = ld3(ring_c, a), na = ld3(ring_n, a), cbv = ld3(ring_c, b), nb = ld3(ring_n, b);
            const num::d3 pab = num::sub(ca, cbv);
            const double dist = num::norm(pab);                 // I:1111 (same value for both visits)
            if (!(dist > 6.0)) {                                // I:1113
                const bool intra = ring_res[a] == ring_res[b];  // I:1091
                const int ct = plane_ctype(ring_sel[a], ring_sel[b], true, true);
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
                const int ga = ring_gid ? ring_gid[a] : a, gb = ring_gid ? ring_gid[b] : b;
                rec.i0 = first ? ga : gb;
                rec.i1 = first ? gb : ga;
                rec.d0 = dist; rec.d1 = dih; rec.d2 = t1; rec.d3 = t2;
                rec.u = (unsigned)y1 | ((unsigned)y2 << 8) | ((unsigned)ct << 16);
            }
        }
        if (Q.nearly_full()) Q.flush(n_out, cap, lane, write);
        Q.push(emit, rec, lane);
    };
    const int wave = (vblock * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (vgrid * blockDim.x) >> 6;
    for (int a = wave; a < nring; a += nwave) {
        if (!ring_plus[a]) continue;  // I:1081
        if (ring_home && !ring_home[a]) continue;  // multi-GPU: owner of the lower ring id emits the pair
        const num::d3 ca = ld3(ring_c, a);
        const Stencil st = stencil_load(g, start, cell_box(g, ca), lane);
        for (int kb = 0; kb < st.pre[9]; kb += 64) {
            const int k = kb + lane;
            bool ok = false;
            int b = 0;
            if (k < st.pre[9]) {
                b = perm[stencil_pos(st, k)];
                // unordered pair once (I:1081, 1085); squared-distance superset of I:1113, settled exactly in eval
                ok = b > a && ring_plus[b] && num::dist2_kd(ca, ld3(ring_c, b)) <= 36.0 * (1.0 + 1e-9);
            }
            P.push(ok, a, b, lane, eval);
        }
    }
    P.drain(lane, eval);
    Q.flush_block(sh->n, &sh->base, n_out, cap, lane, write);
}

// One wavefront per amide a; lanes = every other amide b of the 27 cells: ordered pairs, float32 (I:1217-1300).
struct GroupGroupArgs {
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
__device__ __forceinline__ void group_group_body(GridDesc g, const int* __restrict__ start, const int* __restrict__ perm,
                                                     int namide, const float* __restrict__ am_c, const float* __restrict__ am_n,
                                                     const uint8_t* __restrict__ am_sel, const uint8_t* __restrict__ am_plus,
                                                     const uint8_t* __restrict__ am_home, const int* __restrict__ am_gid,
                                                     long long cap, int* __restrict__ out_bgn, int* __restrict__ out_end,
                                                     float* __restrict__ out_dist, float* __restrict__ out_dih,
                                                     float* __restrict__ out_theta, uint8_t* __restrict__ out_ct,
                                                     u64* __restrict__ n_out, int vblock, int vgrid, PlaneShared* sh) {
    PlaneQueue Q{sh->q[threadIdx.x >> 6], 0};
    PairQueue P{sh->pq[threadIdx.x >> 6], 0};
    auto write = [&](long long slot, const PlaneRec& t) {   // float values travel as doubles (exact both ways)
        out_bgn[slot] = t.i0; out_end[slot] = t.i1;
        out_dist[slot] = (float)t.d0; out_dih[slot] = (float)t.d1; out_theta[slot] = (float)t.d2;
        out_ct[slot] = (uint8_t)(t.u & 255u);
    };
    const int lane = threadIdx.x & 63;
    auto eval = [&](bool live, int a, int b) {   // ordered pair of amides, float32 (I:1227-1300)
        bool emit = false;
        PlaneRec rec;
        if (live) {
            const num::f3 ca = lf3(am_c, a), na = lf3(am_n, a), cbv = lf3(am_c, b), nb = lf3(am_n, b);
            const num::f3 pab = num::sub(ca, cbv);
            const float dist = num::norm(pab);                 // I:1268
            if (!(dist > (float)6.0)) {                        // I:1270
                const float cosd = num::dot(na, nb) / (num::norm(na) * num::norm(nb));
                const float dih = num::fold_deg(acosf(cosd));  // I:1278
                const float theta = num::group_angle(na, pab); // I:1279
                emit = !(dih > 30.0f || theta > 30.0f);        // I:1282
                rec.i0 = am_gid ? am_gid[a] : a; rec.i1 = am_gid ? am_gid[b] : b;
                rec.d0 = (double)dist; rec.d1 = (double)dih; rec.d2 = (double)theta;
                rec.u = (unsigned)plane_ctype(am_sel[a], am_sel[b], true, true);
            }
        }
        if (Q.nearly_full()) Q.flush(n_out, cap, lane, write);
        Q.push(emit, rec, lane);
    };
    const int wave = (vblock * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (vgrid * blockDim.x) >> 6;
    for (int a = wave; a < namide; a += nwave) {
        if (!am_plus[a]) continue;
        if (am_home && !am_home[a]) continue;  // multi-GPU: owner of the bgn amide emits
        const num::d3 cad = num::to_d3(lf3(am_c, a));
        const Stencil st = stencil_load(g, start, cell_box(g, cad), lane);
        for (int kb = 0; kb < st.pre[9]; kb += 64) {
            const int k = kb + lane;
            bool ok = false;
            int b = 0;
            if (k < st.pre[9]) {
                b = perm[stencil_pos(st, k)];
                // I:1233, 1237; float64 squared-distance superset of the float32 test at I:1270 (settled exactly in eval)
                ok = b != a && am_plus[b] && num::dist2_kd(cad, num::to_d3(lf3(am_c, b))) <= 36.0 * (1.0 + 1e-5);
            }
            P.push(ok, a, b, lane, eval);
        }
    }
    P.drain(lane, eval);
    Q.flush_block(sh->n, &sh->base, n_out, cap, lane, write);
}

// One wavefront per amide; lanes = rings of the 27 cells of the RING grid around the amide centre (I:1302-1382).
struct GroupPlaneArgs {
    GridDesc g;
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
__device__ __forceinline__ void group_plane_body(GridDesc g, const int* __restrict__ start, const int* __restrict__ perm,
                                                     int namide, const float* __restrict__ am_c, const float* __restrict__ am_n,
                                                     const uint8_t* __restrict__ am_sel, const uint8_t* __restrict__ am_plus,
                                                     const double* __restrict__ ring_c, const double* __restrict__ ring_n,
                                                     const uint8_t* __restrict__ ring_sel, const uint8_t* __restrict__ ring_plus,
                                                     const uint8_t* __restrict__ am_home, const int* __restrict__ am_gid,
                                                     const int* __restrict__ ring_gid,
                                                     long long cap, int* __restrict__ out_amide, int* __restrict__ out_ring,
                                                     double* __restrict__ out_dist, double* __restrict__ out_dih,
                                                     double* __restrict__ out_theta, uint8_t* __restrict__ out_ct,
                                                     u64* __restrict__ n_out, int vblock, int vgrid, PlaneShared* sh) {
    PlaneQueue Q{sh->q[threadIdx.x >> 6], 0};
    PairQueue P{sh->pq[threadIdx.x >> 6], 0};
    auto write = [&](long long slot, const PlaneRec& t) {
        out_amide[slot] = t.i0; out_ring[slot] = t.i1;
        out_dist[slot] = t.d0; out_dih[slot] = t.d1; out_theta[slot] = t.d2;
        out_ct[slot] = (uint8_t)(t.u & 255u);
    };
    const int lane = threadIdx.x & 63;
    auto eval = [&](bool live, int a, int r) {   // amide a, ring r (I:1312-1382)
        bool emit = false;
        PlaneRec rec;
        if (live) {
            const num::f3 ca = lf3(am_c, a), na = lf3(am_n, a);
            const num::d3 cad = num::to_d3(ca);
            const num::d3 cr = ld3(ring_c, r), nr = ld3(ring_n, r);
            const num::d3 par = num::sub(cad, cr);
            const double dist = num::norm(par);         // I:1349
            if (!(dist > 6.0)) {                        // I:1351
                const double cosd = num::dot(num::to_d3(na), nr) / ((double)num::norm(na) * num::norm(nr));
                const double dih = num::fold_deg(acos(cosd));    // I:1359
                const double theta = num::group_angle(na, par);  // I:1360
                emit = !(dih > 30.0 || theta > 30.0);            // I:1363
                rec.i0 = am_gid ? am_gid[a] : a; rec.i1 = ring_gid ? ring_gid[r] : r;
                rec.d0 = dist; rec.d1 = dih; rec.d2 = theta;
                rec.u = (unsigned)plane_ctype(am_sel[a], ring_sel[r], true, true);
            }
        }
        if (Q.nearly_full()) Q.flush(n_out, cap, lane, write);
        Q.push(emit, rec, lane);
    };
    const int wave = (vblock * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (vgrid * blockDim.x) >> 6;
    for (int a = wave; a < namide; a += nwave) {
        if (!am_plus[a]) continue;
        if (am_home && !am_home[a]) continue;  // multi-GPU: owner of the amide emits
        const num::d3 cad = num::to_d3(lf3(am_c, a));
        const Stencil st = stencil_load(g, start, cell_box(g, cad), lane);
        for (int kb = 0; kb < st.pre[9]; kb += 64) {
            const int k = kb + lane;
            bool ok = false;
            int r = 0;
            if (k < st.pre[9]) {
                r = perm[stencil_pos(st, k)];
                ok = ring_plus[r] && num::dist2_kd(cad, ld3(ring_c, r)) <= 36.0 * (1.0 + 1e-9);   // I:1318; superset of I:1351
            }
            P.push(ok, a, r, lane, eval);
        }
    }
    P.drain(lane, eval);
    Q.flush_block(sh->n, &sh->base, n_out, cap, lane, write);
}

// ---- launchable forms -----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_atom_plane(AtomPlaneArgs a) {
    __shared__ PlaneShared s_sh;
    atom_plane_body(a.g, a.start, a.s_xyzm, a.s_aux, a.nring, a.ring_c, a.ring_n, a.ring_res, a.ring_sel, a.ring_plus, a.plus, a.ring_home, a.ring_gid, a.gid, a.cap, a.out_atom, a.out_ring, a.out_dist, a.out_theta, a.out_mask, a.out_ct, a.n_out, (int)blockIdx.x, (int)gridDim.x, &s_sh);
}
__global__ __launch_bounds__(256) void k_plane_plane(PlanePlaneArgs a) {
    __shared__ PlaneShared s_sh;
    plane_plane_body(a.g, a.start, a.perm, a.nring, a.ring_c, a.ring_n, a.ring_res, a.ring_sel, a.ring_plus, a.ring_home, a.ring_gid, a.cap, a.out_bgn, a.out_end, a.out_dist, a.out_dih, a.out_t1, a.out_t2, a.out_y1, a.out_y2, a.out_ct, a.n_out, (int)blockIdx.x, (int)gridDim.x, &s_sh);
}
__global__ __launch_bounds__(256) void k_group_group(GroupGroupArgs a) {
    __shared__ PlaneShared s_sh;
    group_group_body(a.g, a.start, a.perm, a.namide, a.am_c, a.am_n, a.am_sel, a.am_plus, a.am_home, a.am_gid, a.cap, a.out_bgn, a.out_end, a.out_dist, a.out_dih, a.out_theta, a.out_ct, a.n_out, (int)blockIdx.x, (int)gridDim.x, &s_sh);
}
__global__ __launch_bounds__(256) void k_group_plane(GroupPlaneArgs a) {
    __shared__ PlaneShared s_sh;
    group_plane_body(a.g, a.start, a.perm, a.namide, a.am_c, a.am_n, a.am_sel, a.am_plus, a.ring_c, a.ring_n, a.ring_sel, a.ring_plus, a.am_home, a.am_gid, a.ring_gid, a.cap, a.out_amide, a.out_ring, a.out_dist, a.out_dih, a.out_theta, a.out_ct, a.n_out, (int)blockIdx.x, (int)gridDim.x, &s_sh);
}

// __calculate_atom_plane_contacts (I:947-1062) on the CONTACT grid of the pass (cell edge = the interacting cut-off,
// atoms of selection_plus without hydrogens — exactly the atoms I:964-968 let through) instead of the all-atom 6 A
// grid: with a 5 A edge the 6 A query reaches two cells, so the stencil is (2R + 1)^2 rows of 2R + 1 contiguous cells,
// R = floor(6 / edge) + 1.  One wavefront per ring; lane r fetches the bounds of row r, the rows are then swept one
// after the other, 64 atoms at a time.  Lets a pass do without the second grid build.
__device__ __forceinline__ void atom_plane_cg_body(const AtomPlaneArgs& A, int vblock, int vgrid, PlaneShared* sh) {
    PlaneQueue Q{sh->q[threadIdx.x >> 6], 0};
    PairQueue P{sh->pq[threadIdx.x >> 6], 0};
    auto write = [&](long long slot, const PlaneRec& t) {
        A.out_atom[slot] = t.i0; A.out_ring[slot] = t.i1;
        A.out_dist[slot] = t.d0; A.out_theta[slot] = t.d1;
        A.out_mask[slot] = (uint8_t)(t.u & 255u); A.out_ct[slot] = (uint8_t)((t.u >> 8) & 255u);
    };
    const GridDesc g = A.g;
    const int lane = threadIdx.x & 63;
    auto eval = [&](bool live, int r, int j) {   // ring r, atom at sorted position j (inside the tree radius, I:960)
        bool emit = false;
        PlaneRec rec;
        if (live) {
            const num::d3 ctr_ = ld3(A.ring_c, r), nrm = ld3(A.ring_n, r);
            const float4 v = A.s_xyzm[j];
            const num::d3 x = {(double)v.x, (double)v.y, (double)v.z};
            const uint32_t m = __float_as_uint(v.w);
            const int lid = A.s_aux[j].x;
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
        if (Q.nearly_full()) Q.flush(A.n_out, A.cap, lane, write);
        Q.push(emit, rec, lane);
    };
    const int wave = (vblock * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (vgrid * blockDim.x) >> 6;
    const int R = (int)floor(6.0 * g.inv) + 1, W = 2 * R + 1, nrows = W * W;
    for (int r = wave; r < A.nring; r += nwave) {
        if (!A.ring_plus[r]) continue;  // I:957
        if (A.ring_home && !A.ring_home[r]) continue;  // multi-GPU: the rank owning the ring emits
        const num::d3 ctr_ = ld3(A.ring_c, r);
        const CellBox cb = cell_box(g, ctr_);
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
                        // I:960 tree membership (float64, inclusive); hydrogens (I:964) and atoms outside selection_plus
                        // (I:968) are not in this grid; I:975 aromatic atoms
                        ok = num::dist2_kd(ctr_, num::d3{(double)v.x, (double)v.y, (double)v.z}) <= 36.0 &&
                             !(__float_as_uint(v.w) & ARP_T_AROMATIC);
                    }
                    P.push(ok, r, js + k, lane, eval);
                }
            }
        }
    }
    P.drain(lane, eval);
    Q.flush_block(sh->n, &sh->base, A.n_out, A.cap, lane, write);
}

// The four ring / amide loops of a pass in ONE launch: blocks [0, nb0) work as k_atom_plane (on the contact grid),
// [nb0, nb1) as k_plane_plane, [nb1, nb2) as k_group_group, [nb2, nb3) as k_group_plane.  It is launched on the second
// stream as soon as the contact grid and the ring / amide masks exist, i.e. together with the neighbour search, whose
// long tail of retiring blocks leaves room for these ~1500 short blocks; it is over before the sift kernel is.
struct PlanesSplit { int nb0, nb1, nb2, nb3; };   // cumulative block counts of the four parts
__global__ __launch_bounds__(256) void k_planes(AtomPlaneArgs ap, PlanePlaneArgs pp, GroupGroupArgs gg, GroupPlaneArgs gp, PlanesSplit ps,
                                                PublishArgs pub) {
    __shared__ PlaneShared s_sh;
    const int b = (int)blockIdx.x;
    if (b < ps.nb0) {
        atom_plane_cg_body(ap, b, ps.nb0, &s_sh);
    } else if (b < ps.nb1) {
        const PlanePlaneArgs& a = pp;
        plane_plane_body(a.g, a.start, a.perm, a.nring, a.ring_c, a.ring_n, a.ring_res, a.ring_sel, a.ring_plus, a.ring_home, a.ring_gid, a.cap, a.out_bgn, a.out_end, a.out_dist, a.out_dih, a.out_t1, a.out_t2, a.out_y1, a.out_y2, a.out_ct, a.n_out, b - ps.nb0, ps.nb1 - ps.nb0, &s_sh);
    } else if (b < ps.nb2) {
        const GroupGroupArgs& a = gg;
        group_group_body(a.g, a.start, a.perm, a.namide, a.am_c, a.am_n, a.am_sel, a.am_plus, a.am_home, a.am_gid, a.cap, a.out_bgn, a.out_end, a.out_dist, a.out_dih, a.out_theta, a.out_ct, a.n_out, b - ps.nb1, ps.nb2 - ps.nb1, &s_sh);
    } else if (b < ps.nb3) {
        const GroupPlaneArgs& a = gp;
        group_plane_body(a.g, a.start, a.perm, a.namide, a.am_c, a.am_n, a.am_sel, a.am_plus, a.ring_c, a.ring_n, a.ring_sel, a.ring_plus, a.am_home, a.am_gid, a.ring_gid, a.cap, a.out_amide, a.out_ring, a.out_dist, a.out_dih, a.out_theta, a.out_ct, a.n_out, b - ps.nb2, ps.nb3 - ps.nb2, &s_sh);
    }
    pass_end(pub, 1);
}

// interactions.py:715-936 for every pair of the list (sift_body); ends the pass on the main stream.
__global__ __launch_bounds__(256, SIFT_MIN_WAVES) void k_sift(SiftArgs sa, PublishArgs pub) {
    __shared__ SiftShared s_sh;
    sift_body(sa, (int)blockIdx.x, (int)gridDim.x, &s_sh);
    pass_end(pub, 0);
}

// The same two kernels as ONE grid: blocks [0, np) = the ring / amide loops (np a multiple of 8: idle padding blocks),
// blocks [np, np + nsift) = the sift kernel.  One stream, no cross-stream events (those cost the host ~5 us each).
union SiftPlanesShared {
    PlaneShared planes;
    SiftShared sift;
};
__global__ __launch_bounds__(256, SIFT_MIN_WAVES) void k_sift_planes(SiftArgs sa, int nsift, AtomPlaneArgs ap, PlanePlaneArgs pp,
                                                                     GroupGroupArgs gg, GroupPlaneArgs gp, PlanesSplit ps, int np,
                                                                     PublishArgs pub) {
    __shared__ SiftPlanesShared s_sh;
    const int b = (int)blockIdx.x;
    if (b >= np) {
        sift_body(sa, b - np, nsift, &s_sh.sift);
    } else if (b < ps.nb0) {
        atom_plane_cg_body(ap, b, ps.nb0, &s_sh.planes);
    } else if (b < ps.nb1) {
        const PlanePlaneArgs& a = pp;
        plane_plane_body(a.g, a.start, a.perm, a.nring, a.ring_c, a.ring_n, a.ring_res, a.ring_sel, a.ring_plus, a.ring_home, a.ring_gid, a.cap, a.out_bgn, a.out_end, a.out_dist, a.out_dih, a.out_t1, a.out_t2, a.out_y1, a.out_y2, a.out_ct, a.n_out, b - ps.nb0, ps.nb1 - ps.nb0, &s_sh.planes);
    } else if (b < ps.nb2) {
        const GroupGroupArgs& a = gg;
        group_group_body(a.g, a.start, a.perm, a.namide, a.am_c, a.am_n, a.am_sel, a.am_plus, a.am_home, a.am_gid, a.cap, a.out_bgn, a.out_end, a.out_dist, a.out_dih, a.out_theta, a.out_ct, a.n_out, b - ps.nb1, ps.nb2 - ps.nb1, &s_sh.planes);
    } else if (b < ps.nb3) {
        const GroupPlaneArgs& a = gp;
        group_plane_body(a.g, a.start, a.perm, a.namide, a.am_c, a.am_n, a.am_sel, a.am_plus, a.ring_c, a.ring_n, a.ring_sel, a.ring_plus, a.am_home, a.am_gid, a.ring_gid, a.cap, a.out_amide, a.out_ring, a.out_dist, a.out_dih, a.out_theta, a.out_ct, a.n_out, b - ps.nb2, ps.nb3 - ps.nb2, &s_sh.planes);
    }
    pass_end(pub, 0);
}
