// arp_contacts.h — _calculate_atom_contacts (interactions.py:693-936) in ONE kernel: the neighbour search of
// NeighborSearch.search_all (I:707) over the cell-sorted atoms, the residue filters (I:712-741) and, for every pair that
// passes, the part of the loop body that needs no hydrogen coordinates (I:715-936: float32 distance, covalent / clash /
// vdw ladder, metal complex, polar / weak-polar gates, halogen bond, ionic, carbonyl, aromatic, hydrophobic, contact
// type).  The ~10 % of pairs with a hydrogen-geometry branch left (utils.is_hbond / is_weak_hbond /
// is_halogen_weak_hbond, U:73-155) leave a 16-byte task for k_tasks_planes, the launch that ends the pass.
//
// Why one kernel: the split design (k_search -> 8 B/pair list -> k_sift) wrote and re-read the list, gathered two 32-byte
// records per pair that the searching wave had held in registers a moment before, and paid the fixed cost of a launch
// (queue heads, radius table, first gathers, end-of-pass tickets) twice.  Here the wave that finds a pair evaluates it:
//   stage 1  distance tests of a home cell against its half stencil, per-lane hit bit masks (as k_search)
//   stage 2  every lane walks its own hits: residue filters, orientation, and the pair's operands — both atoms'
//            coordinates, meta words and local ids, 40 bytes — go into a per-wave LDS ring (ballot compaction)
//   stage 3  whenever the ring holds 64 pairs: one lane per pair, full wavefront, straight-line float32 code; the
//            finished 16-byte record {i, j, distance, sift | type << 16 | need << 24} is stored in one coalesced
//            1 KiB write
// Output slots: a block owns one chunk of REC_CHUNK records of its XCD's segment at a time; waves take slots from it
// with one LDS compare-and-swap, a chunk that cannot hold the next batch is retired (its fill count goes to the fill
// table) and replaced with ONE returning atomicAdd on the segment head — a few per block, against one per 64 pairs
// (same-address atomics run at ~90 per microsecond on this chip).  The contact list in HBM is therefore a sequence of
// chunks with a fill count each; k_pack_contacts (arp_api.hip, at fetch time) makes the dense columns the C ABI returns.
#pragma once
#include "arp_pairs.h"
#include "arp_planes.h"

#ifndef REC_CHUNK
#define REC_CHUNK 512
#endif
#ifndef TASK_CHUNK
#define TASK_CHUNK 128
#endif
#define RING_CAP 128

struct ContactArgs {
    int4* recs;                 // PAIR_SEGS segments of `cap` records
    unsigned int* rec_fill;     // valid leading records of each chunk: PAIR_SEGS * nfill words
    u64 cap;                    // records per segment
    unsigned int nfill;         // chunks per segment the fill table has room for
    u64* ctr_recs;              // PAIR_SEGS heads (records allocated, holes included)
    uint4* tasks;               // PAIR_SEGS segments of `tcap` tasks {record index, bgn, end, record word 3}
    unsigned int* task_fill;
    u64 tcap;
    unsigned int ntfill;
    u64* ctr_tasks;             // PAIR_SEGS heads
    u64* ctr_emit;              // STAT_SLOTS hashed slots: records written
    u64* ctr_cand;
    u64* ctr_acc;
    SiftSide sd;
    const uint16_t* rad_idx;    // by local id (atoms whose index did not fit the meta word)
    const int4* st_b4;          // first bonded neighbours by local id (k_prepare_static)
    const int* bond_idx;
    const int* gid;             // null: records carry local ids
    double comp;
    int* err;
};

// LDS word of the block's chunk allocator: (chunk index + 1) << 32 | slots used; 0 = no chunk yet
#define CHUNK_LOCKED 0xFFFFFFFFFFFFFFFFull
template <int CH>
__device__ __forceinline__ unsigned long long chunk_alloc(unsigned long long* st, int cnt, u64* head, unsigned int* fill, unsigned int nfill) {
    for (;;) {
        const unsigned long long w = __hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (w == CHUNK_LOCKED) { __builtin_amdgcn_s_sleep(1); continue; }
        const unsigned int used = (unsigned int)w, ck = (unsigned int)(w >> 32);
        if (ck != 0 && used + (unsigned)cnt <= (unsigned)CH) {
            if (atomicCAS(st, w, w + (unsigned long long)cnt) == w) return (unsigned long long)(ck - 1) * CH + used;
            continue;
        }
        if (atomicCAS(st, w, CHUNK_LOCKED) != w) continue;
        if (ck != 0 && ck - 1 < nfill) fill[ck - 1] = used;           // retire the full chunk
        const unsigned long long base = atomicAdd(head, (unsigned long long)CH);
        const unsigned long long nck = base / CH;
        __hip_atomic_store(st, ((nck + 1ull) << 32) | (unsigned long long)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return base;
    }
}
template <int CH>
__device__ __forceinline__ void chunk_retire(const unsigned long long* st, unsigned int* fill, unsigned int nfill) {
    const unsigned long long w = *st;
    const unsigned int used = (unsigned int)w, ck = (unsigned int)(w >> 32);
    if (ck != 0 && ck - 1 < nfill) fill[ck - 1] = used;
}

__device__ __forceinline__ double2 meta_rad(uint32_t m, int lid, const double2* s_tab, const ContactArgs& A) {
    unsigned ri = (m >> RAD_META_SHIFT) & 31u;
    if (ri == RAD_META_NONE) {
        ri = A.rad_idx[lid];
        if (ri == RAD_NONE) return A.sd.rad[lid];
    }
    return s_tab[ri & (RAD_TABLE - 1)];
}

struct ContactShared {
    float4 ra[SEARCH_WAVES][RING_CAP];     // ring: bgn x, y, z, meta
    float4 rb[SEARCH_WAVES][RING_CAP];     //       end x, y, z, meta
    int2 rij[SEARCH_WAVES][RING_CAP];      //       local ids
    float4 hx[SEARCH_WAVES][HOME_BLOCK];   // home atoms of the moment
    int4 ha[SEARCH_WAVES][HOME_BLOCK];
    double2 tab[RAD_TABLE];
    float4 thr[256];
    unsigned long long rec_state, task_state;
    u64 cand[SEARCH_WAVES], acc[SEARCH_WAVES], emit[SEARCH_WAVES];
};

// stage 3: one pair per lane.  Returns word 3 of the record (sift | type << 16 | need << 24) and the distance.
__device__ __forceinline__ uint32_t pair_stage_a(float4 vb, float4 ve, int b, int e, const ContactShared* sh, const ContactArgs& A,
                                                 float longest_bond, double h_slack, float& d_out) {
    const double2* s_tab = sh->tab;
    const uint32_t mb = __float_as_uint(vb.w), me = __float_as_uint(ve.w);
    const uint32_t tb = mb & M_TMASK, te = me & M_TMASK;
    const num::f3 xb = xyz_of(vb), xe = xyz_of(ve);
    const bool bw = mb & M_WATER, ew = me & M_WATER;
    const int ct = contact_type(mb & M_SEL, me & M_SEL, bw, ew);    // interactions.py:715
    const float d = num::norm(num::sub(xb, xe));                    // interactions.py:745
    d_out = d;
    // interactions.py:748-757: end among the bonded neighbours of bgn (only pairs within the longest bond can be)
    int4 nbr = make_int4(-1, -1, -1, -1);
    const bool near_bond = d <= longest_bond;
    if (near_bond) nbr = A.st_b4[b];
    float f_sum_cov, f_sum_vdw, f_vdw_comp;                         // interactions.py:717-718 and the casts of 756-773
    {
        const unsigned rib = (mb >> RAD_META_SHIFT) & 31u, rie = (me >> RAD_META_SHIFT) & 31u;
        if ((rib | rie) < 16u) {
            const float4 t = sh->thr[rib * 16u + rie];
            f_sum_cov = t.x; f_sum_vdw = t.y; f_vdw_comp = t.z;
        } else {
            const double2 rb = meta_rad(mb, b, s_tab, A), re = meta_rad(me, e, s_tab, A);   // {vdw, cov}
            const double sum_vdw = rb.x + re.x;
            f_sum_cov = (float)(rb.y + re.y); f_sum_vdw = (float)sum_vdw; f_vdw_comp = (float)(sum_vdw + A.comp);
        }
    }
    uint32_t s = 0;
    unsigned need = 0;
    // feature flags that do not depend on the ladder (the clash gate is applied below): interactions.py:786-921
    uint32_t f = 0;
    if (d <= (float)4.5) {
        if (bw && d <= f_vdw_comp) {
            if (te & (ARP_T_HBOND_ACCEPTOR | ARP_T_HBOND_DONOR)) f |= ARP_S_HBOND | ARP_S_POLAR;
        } else if (ew && d <= f_vdw_comp) {
            if (tb & (ARP_T_HBOND_ACCEPTOR | ARP_T_HBOND_DONOR)) f |= ARP_S_HBOND | ARP_S_POLAR;
        } else {
            if ((tb & ARP_T_HBOND_DONOR) && (te & ARP_T_HBOND_ACCEPTOR)) {
                need |= 1u;
                if (d <= (float)3.5) f |= ARP_S_POLAR;
            } else if ((te & ARP_T_HBOND_DONOR) && (tb & ARP_T_HBOND_ACCEPTOR)) {
                need |= 2u;
                if (d <= (float)3.5) f |= ARP_S_POLAR;
            }
        }
        if ((tb & ARP_T_HBOND_ACCEPTOR) && (te & ARP_T_WEAK_HBOND_DONOR)) need |= 4u;
        if ((tb & ARP_T_WEAK_HBOND_DONOR) && (te & ARP_T_HBOND_ACCEPTOR)) need |= 8u;
        if ((tb & ARP_T_WEAK_HBOND_ACCEPTOR) && (mb & M_HALOGEN) && (te & (ARP_T_HBOND_DONOR | ARP_T_WEAK_HBOND_DONOR))) need |= 16u;
        if ((te & ARP_T_WEAK_HBOND_ACCEPTOR) && (me & M_HALOGEN) && (tb & (ARP_T_HBOND_DONOR | ARP_T_WEAK_HBOND_DONOR))) need |= 32u;
        if ((need & 60u) && d <= (float)3.5) f |= ARP_S_WEAK_POLAR;   // each applicable weak branch sets it (I:861,869,877,885)
        // interactions.py:898-904
        if (d <= (float)4.0) {
            if ((tb & ARP_T_POS_IONISABLE) && (te & ARP_T_NEG_IONISABLE)) f |= ARP_S_IONIC;
            else if ((tb & ARP_T_NEG_IONISABLE) && (te & ARP_T_POS_IONISABLE)) f |= ARP_S_IONIC;
        }
        // interactions.py:907-913
        if (d <= (float)3.6) {
            if ((tb & ARP_T_CARBONYL_OXYGEN) && (te & ARP_T_CARBONYL_CARBON)) f |= ARP_S_CARBONYL;
            else if ((te & ARP_T_CARBONYL_OXYGEN) && (tb & ARP_T_CARBONYL_CARBON)) f |= ARP_S_CARBONYL;
        }
        // interactions.py:916-917, 920-921
        if ((tb & te & ARP_T_AROMATIC) && d <= (float)4.0) f |= ARP_S_AROMATIC;
        if ((tb & te & ARP_T_HYDROPHOBE)) f |= ARP_S_HYDROPHOBIC;   // (d <= 4.5 holds here)
    }
    // interactions.py:748-773: covalent test, then float32 distance against Python floats -> float32 compare
    bool cov = false;
    if (near_bond) {
        cov = nbr.x == e || nbr.y == e || nbr.z == e || nbr.w == e;       // (-1 / -2 never equal a local id)
        if (!cov && nbr.w == -2)                                          // more than four neighbours: the rest of the list
            for (int k = A.sd.bond_off[b] + 3, k1 = A.sd.bond_off[b + 1]; k < k1; ++k)
                if (A.bond_idx[k] == e) { cov = true; break; }
    }
    if (cov) s |= ARP_S_COVALENT;
    else if (d < f_sum_cov) s |= ARP_S_CLASH;
    else if (d < f_sum_vdw) s |= ARP_S_VDW_CLASH;
    else if (d <= f_vdw_comp) s |= ARP_S_VDW;
    else s |= ARP_S_PROXIMAL;
    // interactions.py:777-783
    if (d <= (float)2.8) {
        if ((tb & ARP_T_HBOND_ACCEPTOR) && (me & M_METAL)) s |= ARP_S_METAL_COMPLEX;
        else if ((te & ARP_T_HBOND_ACCEPTOR) && (mb & M_METAL)) s |= ARP_S_METAL_COMPLEX;
    }
    // interactions.py:786: not clash (covalent pairs do get feature flags) and d <= 4.5
    if ((s & ARP_S_CLASH) || !(d <= (float)4.5)) need = 0;
    else {
        s |= f;
        if (need) {
            // Branches that cannot succeed need no hydrogen loop: the donor has no hydrogen, the halogen no single-bond
            // neighbour (U:139-141), or the partner is further from the donor than the test's reach 1.2 + vdw + comp
            // (U:86, 109, 145) plus the longest atom - hydrogen distance of the structure.  If EVERY applicable branch is
            // such a one the pair gets no hbond / weak hbond bit — what the loops would find — and leaves no task; if one
            // is left the task runs with the full set (the last applicable weak branch decides, I:857-886).
            const double dd = (double)d;
            const bool far_e = dd > 1.2 + meta_rad(me, e, s_tab, A).x + A.comp + h_slack;   // target = end (acceptor / halogen)
            const bool far_b = dd > 1.2 + meta_rad(mb, b, s_tab, A).x + A.comp + h_slack;   // target = bgn
            unsigned dead = 0;
            if (!(mb & M_HAS_H) || far_e) dead |= 1u | 8u | 32u;               // hydrogens of bgn
            if (!(me & M_HAS_H) || far_b) dead |= 2u | 4u | 16u;               // hydrogens of end
            if (!(mb & M_HAS_SB)) dead |= 16u;
            if (!(me & M_HAS_SB)) dead |= 32u;
            if ((need & ~dead) == 0) need = 0;
        }
        // interactions.py:889-895
        if (d <= f_vdw_comp) {
            if ((tb & ARP_T_XBOND_DONOR) && (te & ARP_T_XBOND_ACCEPTOR)) {
                if (xbond(A.sd.sb[b], xb, xe, A.err)) s |= ARP_S_XBOND;
            } else if ((te & ARP_T_XBOND_DONOR) && (tb & ARP_T_XBOND_ACCEPTOR)) {
                if (xbond(A.sd.sb[e], xe, xb, A.err)) s |= ARP_S_XBOND;
            }
        }
    }
    return s | ((uint32_t)ct << 16) | (need << 24);
}

#ifndef CONTACT_MIN_WAVES
#define CONTACT_MIN_WAVES 1
#endif
__global__ __launch_bounds__(64 * SEARCH_WAVES, CONTACT_MIN_WAVES) void k_contacts(GridDesc g, const int* __restrict__ start,
                                                                                  const float4* __restrict__ s_xyzm,
                                                                                  const int4* __restrict__ s_aux, double r2,
                                                                                  int include_seq_adj, int count_owned, ContactArgs A) {
    __shared__ ContactShared sh;
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    // XCD-aware remap: blocks that land on one XCD (b % 8) walk a contiguous run of cells
    const int nb = gridDim.x;
    const int per = nb >> 3;
    int vb = blockIdx.x;
    if (per > 0 && blockIdx.x < per * 8) vb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const int cells_per_block = (g.ncell + nb - 1) / nb;
    const int c_begin = vb * cells_per_block + w;
    const int c_end = min((vb + 1) * cells_per_block, g.ncell);

    // radius table and the pairwise float32 thresholds of the ladder (I:717-718, 756-773)
    for (int k = threadIdx.x; k < RAD_TABLE; k += blockDim.x) sh.tab[k] = A.sd.rad_tab[k];
    if (threadIdx.x < 256) {
        const double2 ra = A.sd.rad_tab[threadIdx.x >> 4], rb_ = A.sd.rad_tab[threadIdx.x & 15];
        const double sv = ra.x + rb_.x;
        sh.thr[threadIdx.x] = make_float4((float)(ra.y + rb_.y), (float)sv, (float)(sv + A.comp), 0.0f);
    }
    if (threadIdx.x == 0) { sh.rec_state = 0ull; sh.task_state = 0ull; }
    const float longest_bond = A.sd.longest_bond[0];
    const double h_slack = (double)A.sd.longest_bond[1] + 1e-4;   // |H - A| >= |D - A| - h_slack for every hydrogen H of D
    __syncthreads();

    const int seg = blockIdx.x & (PAIR_SEGS - 1);
    int4* const seg_recs = A.recs + (size_t)seg * A.cap;
    uint4* const seg_tasks = A.tasks + (size_t)seg * A.tcap;
    unsigned int* const seg_fill = A.rec_fill + (size_t)seg * A.nfill;
    unsigned int* const seg_tfill = A.task_fill + (size_t)seg * A.ntfill;

    int qn = 0, qhead = 0;                  // ring: entries held, index of the oldest
#ifdef CX_NOALLOC
    int n_emit_w = 0;
#endif
    unsigned int n_cand = 0, n_acc = 0, n_emit = 0;
    const float r2_lo = (float)(r2 * (1.0 - 1e-5)), r2_hi = (float)(r2 * (1.0 + 1e-5));

    auto eval_batch = [&](int cnt) __attribute__((always_inline)) {        // stage 3 on the `cnt` oldest ring entries (wave-uniform)
        __builtin_amdgcn_wave_barrier();
        unsigned long long slot = 0;
#ifdef CX_NOALLOC
        slot = ((unsigned long long)(blockIdx.x >> 3) * SEARCH_WAVES + w) * 448 + (n_emit_w & 255);
#else
        if (lane == 0) slot = chunk_alloc<REC_CHUNK>(&sh.rec_state, cnt, A.ctr_recs + seg, seg_fill, A.nfill);
        slot = __shfl(slot, 0);
#endif
        const int k = (qhead + lane) & (RING_CAP - 1);
        const bool live = lane < cnt;
        uint32_t w3 = 0;
        if (live) {
            const float4 vb_ = sh.ra[w][k], ve_ = sh.rb[w][k];
            const int2 ij = sh.rij[w][k];
            float d;
#ifdef CX_NOEVAL
            d = vb_.x - ve_.x; w3 = (uint32_t)ij.x & 0xFFu;
#else
            w3 = pair_stage_a(vb_, ve_, ij.x, ij.y, &sh, A, longest_bond, h_slack, d);
#endif
#ifdef CX_NOTASK
            w3 &= 0x00FFFFFFu;
#endif
#ifdef CX_NOSTORE
            if (slot + lane == 0x7FFFFFFFFFull)
#else
            if (slot + lane < A.cap)
#endif
                seg_recs[slot + lane] = make_int4(A.gid ? A.gid[ij.x] : ij.x, A.gid ? A.gid[ij.y] : ij.y, (int)__float_as_uint(d), (int)w3);
        }
        const bool task = live && (w3 >> 24) != 0u;
        const unsigned long long mt = __ballot(task);
        if (mt) {
            const int tcnt = __popcll(mt);
            unsigned long long tslot = 0;
            if (lane == 0) tslot = chunk_alloc<TASK_CHUNK>(&sh.task_state, tcnt, A.ctr_tasks + seg, seg_tfill, A.ntfill);
            tslot = __shfl(tslot, 0);
            if (task) {
                const unsigned long long t = tslot + __popcll(mt & ((1ull << lane) - 1ull));
                const int2 ij = sh.rij[w][k];
                if (t < A.tcap && slot + lane < A.cap)
                    seg_tasks[t] = make_uint4((unsigned)((size_t)seg * A.cap + slot + lane), (unsigned)ij.x, (unsigned)ij.y, w3);
            }
        }
        __builtin_amdgcn_wave_barrier();
        qhead = (qhead + cnt) & (RING_CAP - 1);
        qn -= cnt;
        n_emit += (lane == 0) ? (unsigned)cnt : 0u;
#ifdef CX_NOALLOC
        n_emit_w += cnt;
#endif
    };

    // The wave's cells are taken eight at a time: lane 8 * ci + r fetches the bounds of range r of cell ci
    // (range 0 = home pencil [own cell, cx+1], ranges 1..4 = the forward pencils [cx-1, cx+1], r = 5: end of
    // the home cell), so the start table costs ONE load latency per eight cells instead of two per cell.
    for (int cg = c_begin; cg < c_end; cg += SEARCH_WAVES * 8) {
      int my_js = 0, my_len = 0;
      {
        const int mycell = cg + (lane >> 3) * SEARCH_WAVES;
        const int r = lane & 7;
        if (mycell < c_end && r < 6) {
            const int cz = mycell / (g.nx * g.ny);
            const int rem = mycell - cz * g.nx * g.ny;
            const int cy = rem / g.nx;
            const int cx = rem - cy * g.nx;
            const int dy = (r == 0 || r == 5) ? 0 : (r == 1) ? 1 : (r - 3);
            const int dz = (r <= 1 || r == 5) ? 0 : 1;
            const int y2 = cy + dy, z2 = cz + dz;
            if (r == 5) {
                my_js = start[mycell + 1];
            } else if (y2 >= 0 && y2 < g.ny && z2 < g.nz) {
                const int rowbase = (z2 * g.ny + y2) * g.nx;
                const int xlo = (r == 0) ? cx : max(cx - 1, 0);
                const int xhi = min(cx + 1, g.nx - 1);
                my_js = start[rowbase + xlo];
                my_len = start[rowbase + xhi + 1] - my_js;
            }
        }
      }
#pragma unroll 1
      for (int ci = 0; ci < 8; ++ci) {
        const int cell = cg + ci * SEARCH_WAVES;
        if (cell >= c_end) break;
        const int hs = __builtin_amdgcn_readlane(my_js, ci * 8);
        const int he = __builtin_amdgcn_readlane(my_js, ci * 8 + 5);
        if (hs == he) continue;
        const int js0 = hs, js1 = __builtin_amdgcn_readlane(my_js, ci * 8 + 1),
                  js2 = __builtin_amdgcn_readlane(my_js, ci * 8 + 2), js3 = __builtin_amdgcn_readlane(my_js, ci * 8 + 3),
                  js4 = __builtin_amdgcn_readlane(my_js, ci * 8 + 4);
        const int o1 = __builtin_amdgcn_readlane(my_len, ci * 8);      // candidates [0, o1) come from range 0
        const int o2 = o1 + __builtin_amdgcn_readlane(my_len, ci * 8 + 1);
        const int o3 = o2 + __builtin_amdgcn_readlane(my_len, ci * 8 + 2);
        const int o4 = o3 + __builtin_amdgcn_readlane(my_len, ci * 8 + 3);
        const int total = o4 + __builtin_amdgcn_readlane(my_len, ci * 8 + 4);
#pragma unroll 1
        for (int hb = hs; hb < he; hb += HOME_BLOCK) {  // home atoms, 32 at a time: one bit each in the per-lane hit masks
            const int hcount = min(HOME_BLOCK, he - hb);
            const int hpos = min(hb + lane, he - 1);   // (clamped: no branch around the loads; lanes >= hcount are never read)
            const float4 hreg = s_xyzm[hpos];
            const int4 hauxreg = s_aux[hpos];
            // the hit stage below addresses home atoms by a per-lane index: keep them in LDS as well
            __builtin_amdgcn_wave_barrier();
            if (lane < HOME_BLOCK) { sh.hx[w][lane] = hreg; sh.ha[w][lane] = hauxreg; }
            __builtin_amdgcn_wave_barrier();
#pragma unroll 1
            for (int kb = 0; kb < total; kb += 128) {  // the ~87 candidates of this cell, two per lane
                // second candidate of the lane in REVERSE order: the candidates most likely to hit come first in the list (home
                // pencil, then the pencils of the same layer); lane l pairs candidate l with 127 - l, which evens out the hits
                const int k0 = kb + lane, k1 = kb + 127 - lane;
                const bool valid0 = k0 < total, valid1 = k1 < total;
                const int j0 = valid0 ? cand_pos(k0, o1, o2, o3, o4, js0, js1 - o1, js2 - o2, js3 - o3, js4 - o4) : hs;
                const int j1 = valid1 ? cand_pos(k1, o1, o2, o3, o4, js0, js1 - o1, js2 - o2, js3 - o3, js4 - o4) : hs;
                const float4 x0 = s_xyzm[j0];
                const float4 x1 = s_xyzm[j1];
                const int4 a0 = s_aux[j0];
                const int4 a1 = s_aux[j1];
                // Inside the home pencil only later entries of the sorted array (j > h) pair up: candidate k of
                // range 0 is position hs + k, so it is tested against home atom h iff k > h - hs.
                const int kk0 = valid0 ? ((k0 < o1) ? k0 : INT_MAX) : -1;
                const int kk1 = valid1 ? ((k1 < o1) ? k1 : INT_MAX) : -1;
                const int t0 = hb - hs;
                if (!count_owned) {   // tests of this chunk, in closed form
                    const int c0 = kk0 < 0 ? 0 : (kk0 == INT_MAX ? hcount : min(max(kk0 - t0, 0), hcount));
                    const int c1 = kk1 < 0 ? 0 : (kk1 == INT_MAX ? hcount : min(max(kk1 - t0, 0), hcount));
                    n_cand += (unsigned)(c0 + c1);
                }
                // ---- stage 1: distance tests only.  Bit hh of lo/hi = candidate within r2_lo / r2_hi of home atom hh.
                // float32 pre-filter: |d2f - d2| <= 4e-7 * d2, so outside the +-1e-5 band the float32 answer IS the float64 answer.
                uint32_t lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0;
                const v2f cx = {x0.x, x1.x}, cy = {x0.y, x1.y}, cz = {x0.z, x1.z};
#pragma unroll 1
                for (int hh = hcount - 1; hh >= 0; --hh) {
                    const float hx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(hreg.x), hh));
                    const float hy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(hreg.y), hh));
                    const float hz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(hreg.z), hh));
                    const v2f dx = hx - cx, dy = hy - cy, dz = hz - cz;
                    const v2f dd = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
                    const float d0 = dd.x, d1 = dd.y;
                    lo0 = shl1_or_le(lo0, d0, r2_lo);
                    hi0 = shl1_or_le(hi0, d0, r2_hi);
                    lo1 = shl1_or_le(lo1, d1, r2_lo);
                    hi1 = shl1_or_le(hi1, d1, r2_hi);
                    if (count_owned) {
                        // sharded run: a boundary pair is tested on two ranks; count it for the owner of its bgn atom only
                        const int t = t0 + hh;
                        const bool te0 = kk0 > t, te1 = kk1 > t;
                        const int lh = __builtin_amdgcn_readlane(hauxreg.x, hh);
                        const uint32_t mh0 = __builtin_amdgcn_readlane(__float_as_uint(hreg.w), hh);
                        n_cand += (unsigned)(te0 && (((lh < a0.x) ? mh0 : __float_as_uint(x0.w)) & M_HOME));
                        n_cand += (unsigned)(te1 && (((lh < a1.x) ? mh0 : __float_as_uint(x1.w)) & M_HOME));
                    }
                }
                {   // pairs that are tested: candidate k of the home pencil meets home atom t iff k > t (bit hh <-> t = t0 + hh)
                    auto tested = [&](int kk) -> uint32_t {
                        const int c = kk - t0;
                        return kk < 0 ? 0u : (c >= 32 ? 0xFFFFFFFFu : (c <= 0 ? 0u : ((1u << c) - 1u)));
                    };
                    const uint32_t te0 = tested(kk0), te1 = tested(kk1);
                    lo0 &= te0; hi0 &= te0; lo1 &= te1; hi1 &= te1;
                }
                // inside the band (rare) the exact Bio.PDB.kdtrees float64 test decides
                uint32_t band0 = hi0 & ~lo0, band1 = hi1 & ~lo1;
                if (__any((band0 | band1) != 0)) {
                    const num::d3 p0 = {(double)x0.x, (double)x0.y, (double)x0.z};
                    const num::d3 p1 = {(double)x1.x, (double)x1.y, (double)x1.z};
                    while (band0) {
                        const int hh = __ffs(band0) - 1;
                        band0 &= band0 - 1;
                        const float4 hv = sh.hx[w][hh];
                        if (num::dist2_kd(num::d3{(double)hv.x, (double)hv.y, (double)hv.z}, p0) <= r2) lo0 |= 1u << hh;
                    }
                    while (band1) {
                        const int hh = __ffs(band1) - 1;
                        band1 &= band1 - 1;
                        const float4 hv = sh.hx[w][hh];
                        if (num::dist2_kd(num::d3{(double)hv.x, (double)hv.y, (double)hv.z}, p1) <= r2) lo1 |= 1u << hh;
                    }
                }
                n_acc += __popc(lo0) + __popc(lo1);
                // ---- stage 2: every lane walks its own hits (both masks of the lane as one 64-bit word)
                unsigned long long lo = (unsigned long long)lo0 | ((unsigned long long)lo1 << 32);
                while (__any(lo != 0)) {
                    const bool has = lo != 0;
                    const int bit = __ffsll((long long)lo) - 1;     // (-1 for a lane without hits: it reads home atom 31 and is masked by `has`)
                    lo &= lo - 1ull;
                    const int hh = bit & 31;
                    const bool use1 = bit >= 32;
                    const int4 aj = make_int4(use1 ? a1.x : a0.x, use1 ? a1.y : a0.y, use1 ? a1.z : a0.z, use1 ? a1.w : a0.w);
                    const float4 xj = make_float4(use1 ? x1.x : x0.x, use1 ? x1.y : x0.y, use1 ? x1.z : x0.z, use1 ? x1.w : x0.w);
                    const float4 xh = sh.hx[w][hh];
                    const int4 ah = sh.ha[w][hh];
                    const uint32_t mh = __float_as_uint(xh.w), mj = __float_as_uint(xj.w);
                    // canonical orientation: bgn = lower packed index.  Only three things depend on it — which residue's
                    // polypeptide flag is read (I:734 tests res_end twice), whose HOME bit decides ownership, and the order
                    // of the two atoms in the record; the same-residue and sequence-neighbour tests are symmetric.
                    const bool h_first = ah.x < aj.x;
                    const uint32_t m_bgn = h_first ? mh : mj;
                    const uint32_t m_end = h_first ? mj : mh;
                    // interactions.py:729 same residue; 733-741 sequence-adjacent residues — one of the four links equal
                    // <=> the smallest of the four XORs is zero —; ownership: the rank owning the bgn atom emits the pair
                    const unsigned adj = min(min((unsigned)(ah.w ^ aj.y), (unsigned)(ah.z ^ aj.y)), min((unsigned)(aj.w ^ ah.y), (unsigned)(aj.z ^ ah.y)));
                    const unsigned gate = (include_seq_adj ? 0u : 1u) & ((m_end & M_RES_POLY) ? 1u : 0u) & ((mh & mj & M_RES_HASSEQ) ? 1u : 0u);
                    const unsigned drop = (ah.y == aj.y ? 1u : 0u) | (gate & (adj == 0u ? 1u : 0u)) | ((m_bgn & M_HOME) ? 0u : 1u);
                    const bool pass = has & (drop == 0u);
                    const unsigned long long mp = __ballot(pass);
                    if (mp) {
                        if (pass) {
                            const int k = (qhead + qn + __popcll(mp & ((1ull << lane) - 1ull))) & (RING_CAP - 1);
                            // (component-wise selects: a select between two float4 objects goes through scratch memory)
                            sh.ra[w][k] = make_float4(h_first ? xh.x : xj.x, h_first ? xh.y : xj.y, h_first ? xh.z : xj.z, __uint_as_float(m_bgn));
                            sh.rb[w][k] = make_float4(h_first ? xj.x : xh.x, h_first ? xj.y : xh.y, h_first ? xj.z : xh.z, __uint_as_float(m_end));
                            sh.rij[w][k] = make_int2(min(ah.x, aj.x), max(ah.x, aj.x));
                        }
                        qn += __popcll(mp);
#ifdef CX_NORING
                        qn = 0;
#endif
                        if (qn >= 64) eval_batch(64);
                    }
                }
            }
        }
      }
    }
    if (qn > 0) eval_batch(qn);
    // End of block: statistics with one atomic each per block; the last chunks' fill counts
    const u64 w_cand = wave_sum_u32(n_cand), w_acc = wave_sum_u32(n_acc), w_emit = wave_sum_u32(n_emit);
    if (lane == 0) { sh.cand[w] = w_cand; sh.acc[w] = w_acc; sh.emit[w] = w_emit; }
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 tc = 0, ta = 0, te = 0;
        for (int k = 0; k < SEARCH_WAVES; ++k) { tc += sh.cand[k]; ta += sh.acc[k]; te += sh.emit[k]; }
        const int slot = blockIdx.x & (STAT_SLOTS - 1);
        atomicAdd(A.ctr_cand + slot, tc);
        atomicAdd(A.ctr_acc + slot, ta);
        if (te) atomicAdd(A.ctr_emit + slot, te);
        chunk_retire<REC_CHUNK>(&sh.rec_state, seg_fill, A.nfill);
        chunk_retire<TASK_CHUNK>(&sh.task_state, seg_tfill, A.ntfill);
    }
}

// ---- the launch that ends a pass: ring / amide loops from the candidate lists + the hydrogen-geometry tasks ----------
struct TaskArgs {
    const uint4* tasks;
    const unsigned int* task_fill;
    u64 tcap;
    unsigned int ntfill;
    const u64* ctr_tasks;
    int4* recs;
    const float4* st_xyzm;     // static record columns by local id (k_prepare_static)
    const int4* st_q1;
    SiftSide sd;
    const double* h_xyz;
    double comp;
};
__device__ __forceinline__ void tasks_body(const TaskArgs& T, int vblock, int vgrid, double2* s_tab) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int seg = vblock & (PAIR_SEGS - 1);
    const u64 head = T.ctr_tasks[seg];
    s_tab[threadIdx.x] = T.sd.rad_tab[threadIdx.x];   // (blockDim.x == RAD_TABLE)
    __syncthreads();
    const long long nchunk = (long long)min((min(head, T.tcap) + TASK_CHUNK - 1) / TASK_CHUNK, (u64)T.ntfill);
    const uint4* __restrict__ seg_tasks = T.tasks + (size_t)seg * T.tcap;
    const unsigned int* __restrict__ seg_fill = T.task_fill + (size_t)seg * T.ntfill;
    // a wave takes half a chunk (64 tasks) at a time
    const long long nhalf = nchunk * (TASK_CHUNK / 64);
    const long long stride = (long long)(vgrid / PAIR_SEGS) * 4;
    for (long long hc = (long long)(vblock / PAIR_SEGS) * 4 + w; hc < nhalf; hc += stride) {
        const long long ck = hc / (TASK_CHUNK / 64);
        const int first = (int)(hc % (TASK_CHUNK / 64)) * 64;
        const int fill = (int)seg_fill[ck];
        if (first + lane < fill) {
            const uint4 t = seg_tasks[ck * TASK_CHUNK + first + lane];
            SiftRec qb, qe;
            qb.xyzm = T.st_xyzm[t.y]; qb.q1 = T.st_q1[t.y];
            qe.xyzm = T.st_xyzm[t.z]; qe.q1 = T.st_q1[t.z];
            const uint32_t add = sift_geometry(qb, qe, t.w >> 24, T.h_xyz, s_tab, T.sd, T.comp);
            reinterpret_cast<uint32_t*>(T.recs + t.x)[3] = (t.w & 0x00FFFFFFu) | add;
        }
    }
}

// Blocks [0, np) evaluate the ring / amide candidate lists (np a multiple of 8, so that vblock % 8 of the task blocks is
// still the XCD the dispatcher puts them on), blocks [np, np + ntask) run the hydrogen-geometry tasks; the last block to
// finish publishes the counters of the pass (pass_end).
union TasksPlanesShared {
    PlaneShared planes;
    double2 tab[RAD_TABLE];
};
__global__ __launch_bounds__(256, SIFT_MIN_WAVES) void k_tasks_planes(TaskArgs ta, int ntask, AtomPlaneArgs ap, PlanePlaneArgs pp,
                                                                      GroupGroupArgs gg, GroupPlaneArgs gp, PlaneLists L,
                                                                      u64* publish_counts, int np, PublishArgs pub) {
    __shared__ TasksPlanesShared s_sh;
    const int b = (int)blockIdx.x;
    if (b >= np) tasks_body(ta, b - np, ntask, s_sh.tab);
    else planes_from_lists(ap, pp, gg, gp, L, publish_counts, b, np, &s_sh.planes);
    pass_end(pub, 0);
}

// ---- dense columns from the chunked record list (fetch time / device-side consumers) ---------------------------------
// k_chunk_offsets: exclusive prefix of the fill counts of all chunks in use (segment after segment), one block.
// k_pack_contacts: one wavefront per chunk copies its valid records to the dense i / j / distance / sift / type columns.
struct PackArgs {
    const int4* recs;
    const unsigned int* rec_fill;
    u64 cap;
    unsigned int nfill;
    unsigned int nchunk[PAIR_SEGS];   // chunks in use per segment
    unsigned int* offsets;            // exclusive prefix per chunk (sum(nchunk) words) + total
    int* out_i;
    int* out_j;
    float* out_d;
    uint16_t* out_s;
    uint8_t* out_ct;
};
__global__ __launch_bounds__(1024) void k_chunk_offsets(PackArgs P) {
    __shared__ unsigned int s_w[16];
    __shared__ unsigned int s_run;
    unsigned int total_chunks = 0;
    for (int s = 0; s < PAIR_SEGS; ++s) total_chunks += P.nchunk[s];
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (unsigned int base = 0; base < total_chunks; base += 1024) {
        const unsigned int c = base + threadIdx.x;
        unsigned int v = 0;
        if (c < total_chunks) {
            unsigned int k = c, s = 0;
            while (k >= P.nchunk[s]) { k -= P.nchunk[s]; ++s; }
            v = min(P.rec_fill[(size_t)s * P.nfill + k], (unsigned int)REC_CHUNK);
        }
        unsigned int incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        unsigned int before = s_run;
        for (int k = 0; k < wv; ++k) before += s_w[k];
        if (c < total_chunks) P.offsets[c] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_run = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) P.offsets[total_chunks] = s_run;
}
__global__ __launch_bounds__(256) void k_pack_contacts(PackArgs P) {
    unsigned int total_chunks = 0;
    for (int s = 0; s < PAIR_SEGS; ++s) total_chunks += P.nchunk[s];
    const int lane = threadIdx.x & 63;
    for (unsigned int c = blockIdx.x * 4 + (threadIdx.x >> 6); c < total_chunks; c += gridDim.x * 4) {
        unsigned int k = c, s = 0;
        while (k >= P.nchunk[s]) { k -= P.nchunk[s]; ++s; }
        const unsigned int fill = min(P.rec_fill[(size_t)s * P.nfill + k], (unsigned int)REC_CHUNK);
        const unsigned int off = P.offsets[c];
        const int4* __restrict__ src = P.recs + (size_t)s * P.cap + (size_t)k * REC_CHUNK;
        for (unsigned int e = lane; e < fill; e += 64) {
            const int4 r = src[e];
            P.out_i[off + e] = r.x;
            P.out_j[off + e] = r.y;
            P.out_d[off + e] = __int_as_float(r.z);
            P.out_s[off + e] = (uint16_t)((uint32_t)r.w & 0xFFFFu);
            P.out_ct[off + e] = (uint8_t)(((uint32_t)r.w >> 16) & 0xFFu);
        }
    }
}
