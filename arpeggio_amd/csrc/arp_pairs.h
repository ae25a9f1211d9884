// arp_pairs.h — neighbour search over the cell-sorted atoms and the fused per-pair
// SIFt kernel.
//
//   k_search  = NeighborSearch.search_all (interactions.py:707, 1420) + the residue
//               filters of interactions.py:712-741, wave-ballot compaction.
//   k_sift    = the body of _calculate_atom_contacts' loop (interactions.py:715-936)
//               with utils.is_hbond / is_weak_hbond / is_halogen_weak_hbond / is_xbond
//               (utils.py:73-179) and __get_contact_type (interactions.py:643-691).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/arpeggio_hip.h"
#include "arp_grid.h"
#include "arp_numerics.h"

// ---- meta word (float4.w of an atom record) ---------------------------------------
#define M_TMASK 0xFFFu
#define M_FLAG_SHIFT 12
#define M_METAL (1u << 12)
#define M_HALOGEN (1u << 13)
#define M_WATER (1u << 14)
#define M_HYDROGEN (1u << 15)
#define M_ELEM_C (1u << 16)
#define M_ELEM_S (1u << 17)
#define M_RES_MET (1u << 18)
#define M_SEL (1u << 19)
#define M_PLUS (1u << 20)
#define M_RES_POLY (1u << 21)
#define M_RES_HASSEQ (1u << 22)
#define M_HOME (1u << 24)
// what the per-pair kernel needs of an atom's radii and hydrogens rides in the meta word as well (k_prepare_static): the index of
// its {vdw, cov} pair in the radius table if that is below 15 (15: look it up by local id), and whether it has hydrogens at all
#define M_RAD4_SHIFT 25
#define M_RAD4_ESC 15u
#define M_HAS_H (1u << 29)
// ... and how many: bits 30-31 = hydrogens - 1 for one to three of them, 3 = four or more (then the CSR offsets say)
#define M_HCNT_SHIFT 30
#define M_HCNT_ESC 3u

// Counters of a pass (one u64 each).  The enum is the LOGICAL layout — the order of the page-locked host mirror and of every
// h_ctr[] index.  On the device a counter sits at word ctr_dev(logical) of a block of 128-byte lines, because returning
// atomics of different workgroups serialise per cache LINE, not per word (tools/micro/atomic_lines.hip: 768 blocks, one
// atomic each — one word +8 us, eight neighbouring words +8 us, eight words on eight lines +0.9 us): everything that many
// blocks hit at about the same moment — the pair-list heads, the statistics slots, the end-of-pass tickets, the result bags —
// has a line of its own.  Kernels receive pointers to the first of their slots and step by CTR_LINE.
#define STAT_SLOTS 16
#define CTR_LINE 16   // u64 words per 128-byte line
enum {
    C_PAIRS = 0,      // contact pairs enqueued (after the residue filters)
    C_CAND = 1,       // distance tests of the contact search
    C_ACC = 2,        // pairs with d^2 <= cutoff^2
    C_MARK_CAND = 3,  // distance tests of the selection-expansion search
    C_MARK_ACC = 4,
    C_AP = 5, C_PP = 6, C_GG = 7, C_GP = 8,  // records emitted by the ring / amide kernels
    C_SEARCH_PAIRS = 9,  // arp_search_all
    C_SCRATCH0 = 10, C_SCRATCH1 = 11,
    C_BINNED = 12,    // atoms in the contact grid (low 32 bits)
    C_ERR = 15,       // ARP_E_* raised on the device (low 32 bits)
    // statistics counters are spread over STAT_SLOTS lines (hashed by block): slot s holds {cand, acc, mark cand, mark acc}
    C_STAT_CAND = 16, C_STAT_ACC = 16 + STAT_SLOTS, C_STAT_MCAND = 16 + 2 * STAT_SLOTS, C_STAT_MACC = 16 + 3 * STAT_SLOTS,
    C_TAIL = 16 + 4 * STAT_SLOTS,
    // the pair list is written in PAIR_SEGS segments, one queue head per XCD (blockIdx % 8)
    C_SEG_PAIRS = C_TAIL,
    // end-of-pass tickets (pass_end): blocks of k_sift / k_planes that have finished, kernels that have finished
    // (two sets of 8 group tickets + 1 kernel ticket: the sift kernel and the ring / amide kernel end a pass together)
    C_TICKET_GROUP = C_TAIL + 16, C_TICKET_KERNEL = C_TAIL + 24, C_TICKET_SET = 16, C_KERNELS_DONE = C_TAIL + 15,
    C_PLIST = C_TAIL + 8,   // 4 slots: entries of the static ring / amide candidate lists (copied from their own counters each pass)
    // 128 logical words: the last block of a pass hands them to the host with ONE round of returning atomics per thread (pass_end)
    C_COUNT = C_TAIL + 48,
    // device lines: 0 scalars | 1-4 the four bags | 5-20 statistics slots | 21-28 pair-list heads | 29 list entries |
    // 30 kernels done (+ parking for the unused logical words) | 31-39, 40-48 the two ticket sets
    C_DEV_LINES = 49, C_DEV_WORDS = C_DEV_LINES * CTR_LINE
};
__host__ __device__ constexpr int ctr_dev(int i) {
    if (i >= C_AP && i <= C_GP) return (1 + (i - C_AP)) * CTR_LINE;
    if (i < 16) return i;
    if (i < C_TAIL) return (5 + (i - 16) % STAT_SLOTS) * CTR_LINE + (i - 16) / STAT_SLOTS;
    if (i < C_TAIL + 8) return (21 + (i - C_TAIL)) * CTR_LINE;
    if (i >= C_PLIST && i < C_PLIST + 4) return 29 * CTR_LINE + (i - C_PLIST);
    if (i == C_KERNELS_DONE) return 30 * CTR_LINE;
    if (i >= C_TICKET_GROUP && i < C_COUNT && (i - C_TICKET_GROUP) % C_TICKET_SET <= 8)
        return (31 + 9 * ((i - C_TICKET_GROUP) / C_TICKET_SET) + (i - C_TICKET_GROUP) % C_TICKET_SET) * CTR_LINE;
    return 30 * CTR_LINE + 1 + i % 15;   // logical words nothing uses
}
#define PAIR_SEGS 8
typedef unsigned long long u64;

#define M_HAS_SB (1u << 23)

// raw per-atom inputs as uploaded through the C ABI (pointers may be null where noted)
struct RawAtoms {
    const float4* xyz;          // w unused
    const uint16_t* tmask;
    const uint16_t* flags;
    const int* res_id;
    const uint8_t* res_flags;   // null: no residue table
    const int* res_prev;
    const int* res_next;
    const uint8_t* home;        // null: every atom owned by this rank
    const double2* rad;         // {vdw, cov}
    const uint16_t* rad_idx;    // index of the atom's {vdw, cov} in the radius table, RAD_NONE: not in the table
    const int* h_off;
    const int* bond_off;
    const int* bond_idx;
    const float4* sb;           // single-bond neighbour xyz, w = present
};

// What the per-pair kernel holds of each atom of a pair: two 16-byte quads, kept as two columns of the cell-sorted grid and
// copied to LDS group by group —
//   s_xyzm = x, y, z, meta      (the search record; meta carries the radius index and the has-hydrogens bit as well)
//   s_qa   = local atom id, and the first three bonded neighbours (local ids) IN OTHER RESIDUES: -1 = none, .w = -2 = more than
//            three (then the atom's whole CSR list is walked).  Bonded neighbours of the atom's own residue are left out because
//            a pair of one residue never reaches the covalent test (I:729 comes before I:748).
// and a third, 4-byte column for the pairs that need hydrogen geometry (stage B: a tenth of them, gathered by position again):
//   s_h    = h_off, the index of the atom's first hydrogen
// Everything else — float64 radii outside the table's first entries, the halogen's neighbour, a fifth hydrogen — is read by
// local id from the uploaded arrays by the very few pairs that need it.
#define RAD_TABLE 256
#define RAD_NONE 0xFFFFu
#define CNT_SAT 255

// arp_set_single_bond_neighbours: coordinates of every atom's single-bond heavy neighbour (w = 1) or zeros (w = 0)
__global__ __launch_bounds__(256) void k_gather_neighbours(int n, const int* __restrict__ nbr, const float4* __restrict__ xyz,
                                                           float4* __restrict__ sb) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int k = nbr[i];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k >= 0 && k < n) {
            v = xyz[k];
            v.w = 1.0f;
        }
        sb[i] = v;
    }
}

// Everything of an atom record that does not depend on the selection, composed once per structure and
// kept as 16-byte columns so that the per-pass grid builds read them coalesced.  The columns are held in a SPATIAL order
// fixed once per structure (6 A cells, x fastest; k_static_bin / k_static_permute): row k is some atom whose local id is
// aux[k].x, and the atoms of one wavefront lie in a handful of neighbouring cells whatever the cell size of the pass —
// so the binning kernel can combine the histogram atomics of a wave (one per distinct cell instead of one per atom: the
// memory-side atomic rate was its bound) and the scatter writes land next to each other.  sel / plus are by local id.
struct StaticAtoms {
    const float4* xyzm;         // x, y, z, static meta
    const int4* aux;            // local id, residue, previous residue, next residue
    const int4* qa;             // local id + bonded neighbours in other residues (see above)
    const int* hoff;            // index of the first hydrogen
    const uint8_t* sel;         // null: nothing selected
    const uint8_t* plus;        // null: everything in selection_plus
    int all;                    // the selection is the whole structure (then selection_plus is, too): sel / plus not read
};

// Longest bond of the structure (float32 distance, rounded up a little), once per uploaded structure: k_sift runs its
// covalent test — a walk over the bonded neighbours of bgn, dependent loads — only for pairs that are at most this far
// apart (a pair further apart than every bond of the structure is not bonded).  out = float bits, atomicMax on unsigned
// (positive floats order like their bit patterns).
// out[1] = the longest distance between an atom and one of its hydrogens (same rounding up): a hydrogen of D is no nearer to A
// than |D - A| minus this, which lets k_sift leave the hydrogen loops of far pairs alone.
// (computed by k_prepare_static, below, on its one walk over the bonds)

// What the classic setters check on the host (arp_set_atoms ...), for structures that arrive as one blob: done on the
// device (the arrays are already there), result in *err (0 / ARP_E_ARG) which arp_set_blob waits for.
struct BlobCheck {
    int n, nres, nbond, nh, nring, namide, nrad;
    float lo[3], hi[3];           // bounding box the grids are sized from
    const float4* xyz;
    const double2* rad;
    const uint16_t* rad_idx;
    const int* res_id;
    const int* res_prev;
    const int* res_next;
    const int* bond_off;
    const int* bond_idx;
    const int* h_off;
    const double* h_xyz;
    const int* sb_nbr;
    const double* ring_c;
    const int* ring_res;
    const float* am_c;
    const int* am_res;
    int* err;
    float4* sb_out;               // not null: also the coordinates of every atom's single-bond heavy neighbour (k_gather_neighbours' work)
    // two fills the first pass over the structure would otherwise launch (neither reads the structure): the default selection
    // (all ones, n bytes) and the cleared words of the static order's histogram
    uint32_t* fill_ones;          // may be null; fill_ones_n 32-bit words
    int fill_ones_n;
    int4* fill_zero;              // may be null; fill_zero_n 16-byte words
    int fill_zero_n;
    // not null: the per-atom radii did not travel (every atom's pair is in the table: arp_set_blob skips that sixth of the
    // upload) and are written here from the table
    double2* rad_out;
    const double2* rad_tab;
    // not null: a failed check also stores `seq` (the number of this upload) there, where it stays (see arp_set_blob)
    int* bad_seq;
    int seq;
};
// (k_validate_blob itself: behind pass_end, which it ends with)

// Once per uploaded structure, ONE launch: the static record columns, the 6 A cell of every atom for their spatial order
// (histogram + rank in cell: what k_static_bin did as a launch of its own) and the longest bond / atom - hydrogen distance.
__global__ __launch_bounds__(256) void k_prepare_static(RawAtoms r, int n, float4* __restrict__ st_xyzm, int4* __restrict__ st_aux,
                                                        int4* __restrict__ st_qa, int* __restrict__ st_h, GridDesc g6, int* __restrict__ cnt6,
                                                        int2* __restrict__ cr6, const double* __restrict__ h_xyz, unsigned int* __restrict__ longest,
                                                        const int* __restrict__ bad_seq, int seq) {
    // (enqueued ahead of the verdict of the upload's device-side check: a structure that failed it is not to be indexed)
    if (bad_seq && __hip_atomic_load(bad_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == seq) return;
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gstride = gridDim.x * blockDim.x;
    // One walk over an atom's bonds serves both the longest bond (float32 distance, rounded up a little) and the neighbours in other
    // residues, and it goes FOUR bonds at a time: index -> {coordinates, residue} is a chain of dependent loads per bond, and with
    // 1.5 waves per SIMD at 100 k atoms nothing hides it (21 us when the two loops walked the list one bond after the other).
    float lm = 0.0f, lmh = 0.0f;
    for (int i = gtid; i < n; i += gstride) {
        float4 v = r.xyz[i];
        const int res = r.res_id[i];
        const int b0 = r.bond_off[i], b1 = r.bond_off[i + 1];
        const int h0 = r.h_off[i], h1 = r.h_off[i + 1];
        const float4 sb = r.sb[i];
        uint32_t m = (uint32_t)(r.tmask[i] & M_TMASK) | ((uint32_t)(r.flags[i] & 0x7F) << M_FLAG_SHIFT);
        if (!r.home || r.home[i]) m |= M_HOME;
        const uint8_t rf = r.res_flags ? r.res_flags[res] : 0;
        if (rf & ARP_R_POLYPEPTIDE) m |= M_RES_POLY;
        if (rf & ARP_R_HAS_SEQ) m |= M_RES_HASSEQ;
        if (sb.w != 0.0f) m |= M_HAS_SB;
        m |= min((uint32_t)r.rad_idx[i], M_RAD4_ESC) << M_RAD4_SHIFT;
        const int nh = h1 - h0;
        if (nh > 0) m |= M_HAS_H | (min((uint32_t)(nh - 1), M_HCNT_ESC) << M_HCNT_SHIFT);
        const int c6 = cell_index(g6, num::d3{(double)v.x, (double)v.y, (double)v.z}, g6.place ? g6.sid_atom[i] : 0);
        const int rank = atomicAdd(&cnt6[c6], 1);      // (asked for early: the answer travels while the bonds are walked)
        // the bonded neighbours in OTHER residues beside the record (-1: none; w = -2: more than three, walk the CSR list)
        int4 qa = make_int4(i, -1, -1, -1);
        int k = 0;
        for (int b = b0; b < b1; b += 4) {
            int nb[4];
            float4 w[4];
            int rr[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) nb[q] = r.bond_idx[min(b + q, b1 - 1)];
#pragma unroll
            for (int q = 0; q < 4; ++q) { w[q] = r.xyz[nb[q]]; rr[q] = r.res_id[nb[q]]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (b + q >= b1) continue;
                const double dx = (double)v.x - w[q].x, dy = (double)v.y - w[q].y, dz = (double)v.z - w[q].z;
                lm = fmaxf(lm, (float)(sqrt(dx * dx + dy * dy + dz * dz) * (1.0 + 1e-6)));
                if (rr[q] == res) continue;
                if (k == 0) qa.y = nb[q]; else if (k == 1) qa.z = nb[q]; else if (k == 2) qa.w = nb[q]; else qa.w = -2;
                ++k;
            }
        }
        for (int h = h0; h < h1; ++h) {
            const double dx = (double)v.x - h_xyz[3 * (size_t)h], dy = (double)v.y - h_xyz[3 * (size_t)h + 1], dz = (double)v.z - h_xyz[3 * (size_t)h + 2];
            lmh = fmaxf(lmh, (float)(sqrt(dx * dx + dy * dy + dz * dz) * (1.0 + 1e-6)));
        }
        v.w = __uint_as_float(m);
        st_xyzm[i] = v;
        st_qa[i] = qa;
        st_h[i] = h0;
        st_aux[i] = make_int4(i, res, r.res_prev ? r.res_prev[res] : -1, r.res_next ? r.res_next[res] : -1);
        cr6[i] = make_int2(c6, rank);
    }
    // longest bond / atom - hydrogen distance of the structure: one atomicMax per block
    for (int o = 32; o > 0; o >>= 1) { lm = fmaxf(lm, __shfl_xor(lm, o)); lmh = fmaxf(lmh, __shfl_xor(lmh, o)); }
    __shared__ float s_m[4], s_mh[4];
    if ((threadIdx.x & 63) == 0) { s_m[threadIdx.x >> 6] = lm; s_mh[threadIdx.x >> 6] = lmh; }
    __syncthreads();
    if (threadIdx.x == 0) {
        lm = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
        lmh = fmaxf(fmaxf(s_mh[0], s_mh[1]), fmaxf(s_mh[2], s_mh[3]));
        if (lm > 0.0f) atomicMax(longest, __float_as_uint(lm));
        if (lmh > 0.0f) atomicMax(longest + 1, __float_as_uint(lmh));
    }
}

// search record of row i (local id lid): static part + the selection bits of the moment
__device__ __forceinline__ float4 compose_xyzm(const StaticAtoms& r, int i, int lid) {
    float4 v = r.xyzm[i];
    uint32_t m = __float_as_uint(v.w);
    if (r.all) m |= M_SEL | M_PLUS;
    else {
        if (r.sel && r.sel[lid]) m |= M_SEL;
        if (!r.plus || r.plus[lid]) m |= M_PLUS;
    }
    v.w = __uint_as_float(m);
    return v;
}

// ---- the spatial order of the static columns (once per structure) ----
// the same binning alone: a pass with another cell edge re-orders the columns of a structure that is already resident
__global__ __launch_bounds__(256) void k_static_bin(int n, const float4* __restrict__ st_xyzm, GridDesc g, int* __restrict__ cnt,
                                                    int2* __restrict__ cr) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 v = st_xyzm[i];
        const int c = cell_index(g, num::d3{(double)v.x, (double)v.y, (double)v.z}, g.place ? g.sid_atom[i] : 0);
        cr[i] = make_int2(c, atomicAdd(&cnt[c], 1));
    }
}
__global__ __launch_bounds__(256) void k_static_permute(int n, const int2* __restrict__ cr, const int* __restrict__ start,
                                                        const float4* __restrict__ st_xyzm, const int4* __restrict__ st_aux,
                                                        const int4* __restrict__ st_qa, const int* __restrict__ st_h,
                                                        float4* __restrict__ sp_xyzm, int4* __restrict__ sp_aux, int4* __restrict__ sp_qa,
                                                        int* __restrict__ sp_h, int* __restrict__ sp_cell,
                                                        const int* __restrict__ bad_seq, int seq) {
    if (bad_seq && __hip_atomic_load(bad_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == seq) return;      // (see k_prepare_static)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int2 c = cr[i];
        const int pos = start[c.x] + c.y;
        sp_xyzm[pos] = st_xyzm[i];
        sp_aux[pos] = st_aux[i];
        sp_qa[pos] = st_qa[i];
        sp_h[pos] = st_h[i];
        sp_cell[pos] = c.x;
    }
}

// Residue / ring / amide sets of _make_selection (I:1413-1437) ride along with the contact grid build: the binning
// kernel, which reads every atom's selection bits anyway, tags the residues (ResMarks; tag = pass number mod 255 + 1, so
// the arrays never need clearing), and the scatter kernel — the next launch, i.e. after every tag is in place — turns
// them into the ring / amide masks (GroupMasks).  Null pointers / zero counts switch either part off.
struct ResMarks {
    uint8_t* res_sel;
    uint8_t* res_plus;
    uint8_t tag;
};
struct GroupMasks {
    int nring, namide;          // 0, 0: nothing to do
    const int* ring_res;
    const int* amide_res;
    const uint8_t* res_sel;
    const uint8_t* res_plus;
    uint8_t tag;
    int all;                    // every residue counts as selected (arp_set_whole_structure)
    uint8_t* ring_sel;
    uint8_t* ring_plus;
    uint8_t* amide_sel;
    uint8_t* amide_plus;
};
__device__ __forceinline__ void group_masks(const GroupMasks& gm, int first, int stride) {
    for (int i = first; i < gm.nring + gm.namide; i += stride) {
        const bool ring = i < gm.nring;
        const int k = ring ? i : i - gm.nring;
        const int r = ring ? gm.ring_res[k] : gm.amide_res[k];
        const uint8_t s_ = (r >= 0 && (gm.all || gm.res_sel[r] == gm.tag)) ? 1 : 0;    // a ring whose residue is None never qualifies
        const uint8_t p_ = (r >= 0 && (gm.all || gm.res_plus[r] == gm.tag)) ? 1 : 0;
        if (ring) { gm.ring_sel[k] = s_; gm.ring_plus[k] = p_; }
        else { gm.amide_sel[k] = s_; gm.amide_plus[k] = p_; }
    }
}

__global__ __launch_bounds__(256) void k_group_masks(GroupMasks gm) {   // structures without atoms: no scatter launch to ride on
    group_masks(gm, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// cell id + histogram of the atoms passing the filter:
// FILTER 1: active[i] != 0;  FILTER 2: (meta & req) == req && !(meta & forb)
// The returning atomicAdd hands every atom its rank inside its cell, so the scatter needs no atomics of its own
// (device-scope atomics execute at the memory side: ~10 k per microsecond whatever the kernel around them does, and a
// second round of them was a third of the grid build).  The histogram is double-buffered: this build counts in `cell_cnt`
// and clears the buffer of the previous build (`zero_other`), which nothing reads any more.
template <int FILTER>
__global__ __launch_bounds__(256) void k_bin_atoms(StaticAtoms r, int n, GridDesc g, const uint8_t* __restrict__ active,
                                                   uint32_t req, uint32_t forb, int2* __restrict__ cell_rank,
                                                   int* __restrict__ cell_cnt, uint8_t* __restrict__ plus_init,
                                                   int* __restrict__ zero_other, int nzero, ResMarks rm) {
    const int stride = gridDim.x * blockDim.x;
    for (int k = (blockIdx.x * blockDim.x + threadIdx.x) * 4; k < nzero; k += stride * 4)
        *reinterpret_cast<int4*>(zero_other + k) = make_int4(0, 0, 0, 0);          // (buffers are padded to a multiple of 4)
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    const bool need_aux = !r.all || rm.res_sel || plus_init || FILTER == 1 || g.place;
    for (int base = blockIdx.x * blockDim.x + threadIdx.x - lane; base < n; base += stride) {   // (wave-uniform trip count)
        const int i = base + lane;
        const bool valid = i < n;
        const int ii = valid ? i : n - 1;
        const int4 aux = need_aux ? r.aux[ii] : make_int4(0, 0, 0, 0);
        const float4 xyzm = compose_xyzm(r, ii, aux.x);
        const uint32_t m = __float_as_uint(xyzm.w);
        if (valid) {
            if (plus_init) plus_init[aux.x] = r.all ? (uint8_t)1 : r.sel[aux.x];   // I:1407: selection_plus starts as the selection
            if (rm.res_sel) {   // I:1413, 1431: residues of the selection / of selection_plus (hydrogens included), tagged with the pass
                if (m & M_SEL) rm.res_sel[aux.y] = rm.tag;
                if (m & M_PLUS) rm.res_plus[aux.y] = rm.tag;
            }
        }
        const bool on = valid && ((FILTER == 1) ? (active[aux.x] != 0) : (((m & req) == req) && !(m & forb)));
        const int c = on ? cell_index(g, num::d3{(double)xyzm.x, (double)xyzm.y, (double)xyzm.z}, g.place ? g.sid_atom[aux.x] : 0) : -1;
        // one atomic per distinct cell of the wave: the lanes of a cell elect the lowest one, which asks for the whole group
        int leader = 0, before = 0, group = 0;
        unsigned long long todo = __ballot(on);
        while (todo) {
            const int l = __ffsll((long long)todo) - 1;
            const int cl = __shfl(c, l);
            const unsigned long long same = __ballot(on && c == cl);
            if (on && c == cl) { leader = l; before = __popcll(same & below); group = __popcll(same); }
            todo &= ~same;
        }
        int first = 0;
        if (on && lane == leader) first = atomicAdd(&cell_cnt[c], group);
        first = __shfl(first, leader);
        if (valid) cell_rank[i] = make_int2(c, first + before);
    }
}

// one atom's cell-sorted records (search record 32 B; the contact grid adds the second quad of the sift record)
__device__ __forceinline__ void scatter_one(const StaticAtoms& r, int i, int pos, float4* __restrict__ s_xyzm,
                                            int4* __restrict__ s_aux, int4* __restrict__ s_qa, int* __restrict__ s_h) {
    const int4 aux = r.aux[i];
    const float4 xyzm = compose_xyzm(r, i, aux.x);
    s_xyzm[pos] = xyzm;
    s_aux[pos] = aux;
    if (s_qa) { s_qa[pos] = r.qa[i]; s_h[pos] = r.hoff[i]; }
}

// counting-sort scatter fused with the record build, start table from a separate scan (large grids)
__global__ __launch_bounds__(256) void k_scatter_atoms(StaticAtoms r, int n, const int2* __restrict__ cell_rank,
                                                       const int* __restrict__ start, float4* __restrict__ s_xyzm,
                                                       int4* __restrict__ s_aux, int4* __restrict__ s_qa, int* __restrict__ s_h, GroupMasks gm) {
    group_masks(gm, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int2 cr = cell_rank[i];
        if (cr.x < 0) continue;
        scatter_one(r, i, start[cr.x] + cr.y, s_xyzm, s_aux, s_qa, s_h);
    }
}

// Grids whose start table fits in LDS (<= SCAN_LDS_CELLS cells: every structure below ~250 k atoms at 5 A): scan and
// scatter in ONE launch.  Every 1024-thread block scans the whole histogram itself — 70 KB out of L2, all blocks at
// once, against ~11 us for a single-block scan kernel that one CU's load bandwidth bounds, plus a launch — keeps the
// exclusive prefix in LDS, writes its slice of the global start table (the search reads it) and scatters its
// SCAT_ATOMS atoms.  Wave w scans cells [w * chunk, (w + 1) * chunk) in steps of 256 (one coalesced int4 load per
// lane and step, all STEPS loads in flight together), shuffles inside the wave, no barrier until the 16 wave totals
// meet; the LDS table holds prefixes relative to the wave's first cell, the wave offsets sit beside it.
#define SCAT_ATOMS 1024
#define SCAN_LDS_CELLS (16 * 9 * 256)
template <int STEPS>
__global__ __launch_bounds__(1024) void k_scan_scatter_atoms(StaticAtoms r, int n, int ncell, const int2* __restrict__ cell_rank,
                                                             const int* __restrict__ cell_cnt, int* __restrict__ start,
                                                             unsigned long long* __restrict__ total_out,
                                                             float4* __restrict__ s_xyzm, int4* __restrict__ s_aux,
                                                             int4* __restrict__ s_qa, int* __restrict__ s_h, GroupMasks gm) {
    extern __shared__ __attribute__((aligned(16))) int s_start[];   // 16 * STEPS * 256 ints
    group_masks(gm, blockIdx.x * 1024 + threadIdx.x, gridDim.x * 1024);
    __shared__ int s_wtot[16], s_woff[17];
    constexpr int CHUNK = STEPS * 256;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * SCAT_ATOMS + threadIdx.x;
    const int2 cr = (i < n) ? cell_rank[i] : make_int2(-1, 0);      // in flight beside the histogram loads
    // ... and so are the atom's record columns (they do not depend on where the atom goes)
    const int ii = (i < n) ? i : 0;
    const int4 my_aux = r.aux[ii];
    const float4 my_xyzm = compose_xyzm(r, ii, my_aux.x);
    const int4 my_qa = s_qa ? r.qa[ii] : make_int4(0, 0, 0, 0);
    const int my_h = s_qa ? r.hoff[ii] : 0;
    int4 v[STEPS];
#pragma unroll
    for (int k = 0; k < STEPS; ++k) {
        const int c0 = wv * CHUNK + k * 256 + lane * 4;             // (the histogram is padded: reads past ncell are in bounds)
        const int4 q = *reinterpret_cast<const int4*>(cell_cnt + c0);
        v[k] = make_int4(c0 < ncell ? q.x : 0, c0 + 1 < ncell ? q.y : 0, c0 + 2 < ncell ? q.z : 0, c0 + 3 < ncell ? q.w : 0);
    }
    int run = 0;
#pragma unroll
    for (int k = 0; k < STEPS; ++k) {
        const int t = v[k].x + v[k].y + v[k].z + v[k].w;
        int incl = t;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int u = __shfl_up(incl, off);
            if (lane >= off) incl += u;
        }
        const int e = run + incl - t;
        *reinterpret_cast<int4*>(s_start + wv * CHUNK + k * 256 + lane * 4) = make_int4(e, e + v[k].x, e + v[k].x + v[k].y, e + v[k].x + v[k].y + v[k].z);
        run += __shfl(incl, 63);
    }
    if (lane == 0) s_wtot[wv] = run;
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int k = 0; k < 16; ++k) { s_woff[k] = acc; acc += s_wtot[k]; }
        s_woff[16] = acc;
    }
    __syncthreads();
    const int total = s_woff[16];
    {   // global start table: block b writes slice b (start[ncell] = the grand total)
        const int per = ((ncell + 1 + gridDim.x - 1) / gridDim.x + 3) & ~3;
        const int lo = blockIdx.x * per, hi = min(lo + per, ncell + 1);
        for (int k = lo + threadIdx.x; k < hi; k += 1024) start[k] = (k < ncell) ? s_start[k] + s_woff[k / CHUNK] : total;
        if (blockIdx.x == 0 && threadIdx.x == 0 && total_out) *total_out = (unsigned long long)total;
    }
    if (cr.x >= 0) {
        const int pos = s_start[cr.x] + s_woff[cr.x / CHUNK] + cr.y;
        s_xyzm[pos] = my_xyzm;
        s_aux[pos] = my_aux;
        if (s_qa) { s_qa[pos] = my_qa; s_h[pos] = my_h; }
    }
}

#if defined(ARP_COMPACT_TRACE) && !defined(ARP_SEARCH_TRACE)
#define ARP_SEARCH_TRACE      // (shares the search trace's buffer and entry points)
#endif
#if defined(ARP_SIFT_TRACE) && !defined(ARP_SEARCH_TRACE)
#define ARP_SEARCH_TRACE
#endif
#ifdef ARP_SEARCH_TRACE
__device__ unsigned long long* g_search_trace = nullptr;      // developer builds only (tools/*_trace.py)
#endif
// ---- the contact grid of a pass in ONE launch ---------------------------------------------------------------------
// The static columns of a structure are kept in the order of the pass's own cells (ensure_spatial: counting sort by cell,
// once per structure and cell edge).  The atoms a pass lets into its grid — selection_plus without hydrogens, I:707-712 —
// are then an ORDERED subset of that array: the grid build is a stream compaction.  Every 1024-thread block takes 1024
// consecutive rows, counts what it keeps (ballot / popcount), publishes the count and finds its base with a decoupled
// look-back over the blocks before it (one 64-bit word per block: launch number, state, value — nothing to clear between
// launches), writes the kept records at base + rank, and the first row of every cell writes the cell's new start.  No
// histogram, no atomics on cells, no second launch: 8.7 + 9.6 us for k_bin_atoms + k_scan_scatter_atoms become one kernel.
struct CompactArgs {
    StaticAtoms r;             // columns in cell order + the selection of the moment
    const int* sp_cell;        // cell of every row (ascending)
    int n, ncell;
    uint32_t req, forb;        // kept: (meta & req) == req && !(meta & forb)
    float4* s_xyzm;
    int4* s_aux;
    int4* s_qa;                // second quad of the sift record
    int* s_h;                  // first hydrogen
    int* start;                // out: ncell + 1
    int* s_cell;               // out: cell of every kept row (k_search splits its blocks by atoms, not by cells, when the grid is sparse)
    unsigned long long* chain; // one word per block
    unsigned int epoch;        // launch number (30 bits)
    unsigned long long* total_out;
    uint8_t* plus_init;        // not null: selection_plus = selection (I:1407), by local id
    ResMarks rm;
    int* err;
};
#define CHAIN_AGG 1ull
#define CHAIN_PFX 2ull
__device__ __forceinline__ unsigned long long chain_word(unsigned int epoch, unsigned long long state, unsigned int value) {
    return ((unsigned long long)epoch << 34) | (state << 32) | (unsigned long long)value;
}
// (rows per block: 512 up to 150 000 rows — more blocks reading at once: 13.0 against 14.0 us at 100 k atoms; 250 k: 19.9 against 17.4 —,
// 1024 beyond, where the shorter look-back wins: 61 against 68 us at 1 M)
template <int COMPACT_THREADS>
__global__ __launch_bounds__(COMPACT_THREADS) void k_compact_atoms(CompactArgs A) {
    __shared__ int s_wtot[16], s_woff[16];
    __shared__ int s_base;
    constexpr int NW = COMPACT_THREADS / 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#ifdef ARP_COMPACT_TRACE
    unsigned long long tc[4] = {__builtin_amdgcn_s_memrealtime(), 0, 0, 0};
#define COMPACT_T(k) tc[k] = __builtin_amdgcn_s_memrealtime()
#else
#define COMPACT_T(k)
#endif
    const int i = blockIdx.x * COMPACT_THREADS + threadIdx.x;
    const bool valid = i < A.n;
    const int ii = valid ? i : A.n - 1;
    const bool need_aux = !A.r.all || A.rm.res_sel || A.plus_init;
    const int4 aux = A.r.aux[ii];
    const float4 xyzm = compose_xyzm(A.r, ii, aux.x);
    const uint32_t m = __float_as_uint(xyzm.w);
    (void)need_aux;
    if (valid) {
        if (A.plus_init) A.plus_init[aux.x] = A.r.all ? (uint8_t)1 : A.r.sel[aux.x];   // I:1407
        if (A.rm.res_sel) {   // I:1413, 1431: residues of the selection / of selection_plus (hydrogens included), tagged with the pass
            if (m & M_SEL) A.rm.res_sel[aux.y] = A.rm.tag;
            if (m & M_PLUS) A.rm.res_plus[aux.y] = A.rm.tag;
        }
    }
    const bool keep = valid && ((m & A.req) == A.req) && !(m & A.forb);
    // the columns of a kept row travel while the counts meet
    const int4 qa = A.r.qa[ii];
    const int hoff = A.r.hoff[ii];
    const int my_cell = A.sp_cell[ii];
    const int prev_cell = (i > 0 && valid) ? A.sp_cell[i - 1] : -1;
    const unsigned long long mk = __ballot(keep);
    const int rank_w = __popcll(mk & ((1ull << lane) - 1ull));
    if (lane == 0) s_wtot[wv] = __popcll(mk);
    COMPACT_T(1);
    __syncthreads();
    if (wv == 0) {
        // block total, wave offsets; publish the aggregate; look back for the base
        int t = (lane < NW) ? s_wtot[lane] : 0;
        int incl = t;
#pragma unroll
        for (int off = 1; off < NW; off <<= 1) {
            const int u = __shfl_up(incl, off);
            if (lane >= off) incl += u;
        }
        if (lane < NW) s_woff[lane] = incl - t;
        const int total = __shfl(incl, NW - 1);
        const int b = (int)blockIdx.x;
        // (relaxed device-scope atomics: a word carries everything its readers need — no other memory is published through it,
        // so none of the L2 write-back / invalidate of a release / acquire pair is wanted: that pair cost 44 us in round 1)
        if (lane == 0)
            __hip_atomic_store(A.chain + b, chain_word(A.epoch, b == 0 ? CHAIN_PFX : CHAIN_AGG, (unsigned)total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int base = 0;
        // (all four windows of a 100 k-atom launch asked for at once — lane l looking at blocks hi - l, hi - 64 - l, ... with four loads
        // in flight — changed nothing: base known at 5.6 us against 5.5 in the per-wave trace; what the look-back waits for is the
        // slowest predecessor's aggregate, not its own round trips)
        for (int hi = b - 1; hi >= 0; hi -= 64) {       // 64 predecessors at a time, nearest first: lane l looks at block hi - l
            const int k = hi - lane;
            unsigned long long wd = chain_word(A.epoch, CHAIN_PFX, 0);      // (lanes before block 0: a zero prefix)
            if (k >= 0) {
                int spins = 0;
                for (;;) {
                    wd = __hip_atomic_load(A.chain + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned int)(wd >> 34) == A.epoch) break;
                    if (++spins > (1 << 20)) { atomicExch(A.err, -2 /* ARP_E_HIP */); wd = chain_word(A.epoch, CHAIN_PFX, 0); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            const bool is_pfx = ((wd >> 32) & 3ull) == CHAIN_PFX;
            const unsigned long long mp = __ballot(is_pfx);
            const int stop = mp ? (__ffsll((long long)mp) - 1) : 64;      // nearest block that knows its whole prefix
            int v = (lane <= stop) ? (int)(unsigned int)wd : 0;
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            base += v;
            if (mp) break;
        }
        if (lane == 0) {
            s_base = base;
            if (b > 0) __hip_atomic_store(A.chain + b, chain_word(A.epoch, CHAIN_PFX, (unsigned)(base + total)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (b == (int)gridDim.x - 1 && A.total_out) *A.total_out = (unsigned long long)(base + total);
        }
    }
    __syncthreads();
    COMPACT_T(2);
    const int kp = s_base + s_woff[wv] + rank_w;        // kept rows before this one
    if (keep) {
        A.s_xyzm[kp] = xyzm;
        A.s_aux[kp] = aux;
        if (A.s_qa) { A.s_qa[kp] = qa; A.s_h[kp] = hoff; }
        if (A.s_cell) A.s_cell[kp] = my_cell;
    }
    // The first row of a cell knows where the cell's kept rows begin; so do the empty cells before it (a protein in its
    // bounding box, the gaps between the structures of a batch: runs of thousands), which the wave fills together.
    if (valid && my_cell != prev_cell) A.start[my_cell] = kp;
    const bool last = valid && i == A.n - 1;
    unsigned long long mg = __ballot((valid && my_cell - prev_cell > 1) || last);
    while (mg) {
        const int l = __ffsll((long long)mg) - 1;
        mg &= mg - 1ull;
        const int lo = __shfl(prev_cell, l) + 1, hi = __shfl(my_cell, l), v = __shfl(kp, l);
        for (int c = lo + lane; c < hi; c += 64) A.start[c] = v;
        if (__shfl(last ? 1 : 0, l)) {       // cells after the last row: the total
            const int tot = v + __shfl(keep ? 1 : 0, l);
            for (int c = hi + 1 + lane; c <= A.ncell; c += 64) A.start[c] = tot;
        }
    }
#ifdef ARP_COMPACT_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    COMPACT_T(3);
    if (g_search_trace && lane == 0) {
        unsigned long long* t = g_search_trace + ((size_t)blockIdx.x * NW + wv) * 4;
        for (int k = 0; k < 4; ++k) t[k] = tc[k];
    }
#endif
}

// ---- end of a pass, without a launch of its own ------------------------------------------------------------
// The last kernels of a pass (k_sift on the main stream, k_planes beside it on the second one) call pass_end() as their
// final statement: every block takes a ticket once its own atomics have been performed, the last block of a kernel
// bumps the kernel count, and the last block of the last kernel publishes the counter block to the pinned host copy,
// returns it to zero for the next pass and stores the pass number the host is polling — what k_publish_counters does
// as a separate 5 us launch behind a cross-stream join.  expected = 0 switches it off (callers that publish themselves).
struct PublishArgs {
    u64* ctr;        // C_COUNT device counters
    u64* host;       // pinned mirror, C_COUNT + 1 words (the last one = completion word)
    int expected;    // kernels that end this pass (0: off)
    u64 seq;         // value of the completion word for this pass
};
// Tickets are hierarchical — one counter per (blockIdx % 8), i.e. per XCD as the dispatcher places blocks, then one for
// the eight groups — because atomics on one LINE run at ~90 per microsecond: 2500 blocks on one word were a 15 us tail, and
// 1024 blocks on eight words of one line still 7 us (config 3); each ticket has a line of its own (ctr_dev).
// The ordering below is spelled for gfx9 (gfx942 / gfx950), not for the language's memory model: on these chips vmcnt counts
// stores as well as loads and returns only when the store has left the CU (on gfx10+ stores have a counter of their own), so
// "s_waitcnt vmcnt(0)" of every wave + the workgroup barrier is what a release of the counter stores amounts to, without the
// L2 write-back a release FENCE of that scope performs.  The asm statements carry a "memory" clobber: the compiler keeps
// the relaxed flag store behind them.  tests/test_gpu_publication.py (and tools/pub_stress.py) are the check on hardware.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "this library is built for gfx950 only: pass_end and the per-pair kernel rely on the gfx9 meaning of s_waitcnt vmcnt for stores, the sort kernels on 160 KB of LDS per CU, the per-pair kernel on global_load_lds_dwordx4"
#endif
__device__ __forceinline__ void pass_end(const PublishArgs& pa, int set) {
    if (!pa.expected) return;
    const int tset = set * C_TICKET_SET;
    __shared__ int s_publisher;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's counter atomics have been performed (memory side)
    __syncthreads();
    if (threadIdx.x == 0) {
        int pub = 0;
        const unsigned grp = blockIdx.x & 7u;
        const u64 members = ((u64)gridDim.x + 7ull - grp) >> 3;                  // blocks b < gridDim.x with b % 8 == grp
        if (atomicAdd(pa.ctr + ctr_dev(C_TICKET_GROUP + tset + (int)grp), 1ull) == members - 1ull) {
            const u64 groups = gridDim.x < 8u ? (u64)gridDim.x : 8ull;
            if (atomicAdd(pa.ctr + ctr_dev(C_TICKET_KERNEL + tset), 1ull) == groups - 1ull)     // last block of this kernel
                pub = pa.expected == 1 || atomicAdd(pa.ctr + ctr_dev(C_KERNELS_DONE), 1ull) == (u64)pa.expected - 1ull;
        }
        s_publisher = pub;
    }
    __syncthreads();
    if (!s_publisher) return;
    // returning atomics read the memory-side value whatever this XCD's L2 holds, and leave the slot zero
    // The mirror is page-locked host memory: system-scope stores go straight to the fabric.  No release fence here — a fence
    // of that scope writes back every dirty line of this XCD's L2 first, and at the end of a pass those are the contact records
    // the kernel has just written (config 3: 19 MB across the eight L2s; the fence was 4 - 5 us of the pass).  What the flag
    // needs is that the counter stores have left this CU (vmcnt(0) of every wave, then the barrier); writes to the host
    // arrive in the order they were sent.
    for (int i = threadIdx.x; i < (int)C_COUNT; i += blockDim.x)
        __hip_atomic_store(pa.host + i, atomicExch(pa.ctr + ctr_dev(i), 0ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(pa.host + C_COUNT, pa.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- validation of an uploaded structure (BlobCheck above) ------------------------------------------
// pub.expected = 1: the last block to finish stores the verdict (the counter block with the error word) in the pinned mirror,
// which the host polls — no copy launch, no stream synchronisation (arp_set_blob); 0: the caller reads bc.err itself.
__global__ __launch_bounds__(256) void k_validate_blob(BlobCheck bc, PublishArgs pub) {
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gstride = gridDim.x * blockDim.x;
    bool bad = false;
    for (int k = gtid; k < bc.nbond; k += gstride) bad |= (unsigned)bc.bond_idx[k] >= (unsigned)bc.n;
    for (int k = gtid; k < 3 * bc.nh; k += gstride) bad |= !isfinite(bc.h_xyz[k]);
    for (int k = gtid; k < bc.nring; k += gstride) {
        bad |= bc.ring_res[k] < -1 || bc.ring_res[k] >= bc.nres;
        bad |= !(isfinite(bc.ring_c[3 * k]) && isfinite(bc.ring_c[3 * k + 1]) && isfinite(bc.ring_c[3 * k + 2]));
    }
    for (int k = gtid; k < bc.namide; k += gstride) {
        bad |= bc.am_res[k] < -1 || bc.am_res[k] >= bc.nres;
        bad |= !(isfinite(bc.am_c[3 * k]) && isfinite(bc.am_c[3 * k + 1]) && isfinite(bc.am_c[3 * k + 2]));
    }
    for (int k = gtid; k < bc.nres; k += gstride)
        bad |= bc.res_prev[k] < -1 || bc.res_prev[k] >= bc.nres || bc.res_next[k] < -1 || bc.res_next[k] >= bc.nres;
    for (int i = gtid; i < bc.n; i += gstride) {
        const float4 v = bc.xyz[i];
        bad |= !(v.x >= bc.lo[0] && v.x <= bc.hi[0] && v.y >= bc.lo[1] && v.y <= bc.hi[1] && v.z >= bc.lo[2] && v.z <= bc.hi[2]);   // (NaN fails)
        bad |= (unsigned)bc.res_id[i] >= (unsigned)bc.nres;
        const int h0 = bc.h_off[i], h1 = bc.h_off[i + 1], b0 = bc.bond_off[i], b1 = bc.bond_off[i + 1];
        bad |= h0 < 0 || h1 < h0 || h1 > bc.nh || b0 < 0 || b1 < b0 || b1 > bc.nbond;
        bad |= (i == 0 && (h0 != 0 || b0 != 0)) || (i == bc.n - 1 && (h1 != bc.nh || b1 != bc.nbond));
        const unsigned ri = bc.rad_idx[i];
        bad |= ri != RAD_NONE && ri >= (unsigned)bc.nrad;
        double2 rd;
        if (bc.rad_out) {
            rd = bc.rad_tab[min(ri, (unsigned)RAD_TABLE - 1u)];
            bc.rad_out[i] = rd;
            bad |= ri == RAD_NONE;      // (the host looked: cannot happen)
        } else {
            rd = bc.rad[i];
        }
        bad |= !(isfinite(rd.x) && isfinite(rd.y));
        const int nb = bc.sb_nbr[i];
        bad |= nb < -1 || nb >= bc.n;
        if (bc.sb_out) {
            float4 s_ = make_float4(0.f, 0.f, 0.f, 0.f);
            if (nb >= 0 && nb < bc.n) { s_ = bc.xyz[nb]; s_.w = 1.0f; }
            bc.sb_out[i] = s_;
        }
    }
    for (int k = gtid; k < bc.fill_ones_n; k += gstride) bc.fill_ones[k] = 0x01010101u;
    for (int k = gtid; k < bc.fill_zero_n; k += gstride) bc.fill_zero[k] = make_int4(0, 0, 0, 0);
    if (bad) {
        atomicExch(bc.err, ARP_E_ARG);
        if (bc.bad_seq) atomicExch(bc.bad_seq, bc.seq);
    }
    pass_end(pub, 0);
}

// ---- neighbour search ---------------------------------------------------------------
#ifndef SEARCH_WAVES
#define SEARCH_WAVES 8
#endif
#ifndef QCAP
#define QCAP 512
#endif
#ifndef SEARCH_MIN_WAVES
#define SEARCH_MIN_WAVES 6
#endif
#define HOME_BLOCK 32
#define DESC_CAP 256

enum { MODE_CONTACTS = 0, MODE_PAIRS = 1, MODE_MARK = 2 };

// One wavefront per home cell.  Half stencil: own cell (later entries) + 13 forward cells,
// expressed as 5 contiguous ranges of the cell-sorted array (cells are x-fastest, so 3
// x-neighbours are contiguous).  The ~87 candidate atoms of the 5 ranges are flattened over
// the lanes and held in registers; the home atoms are broadcast from registers with
// v_readlane, so each record is loaded once per home cell and the inner loop touches no
// memory.
// Accepted pairs are compacted with __ballot into a per-wave LDS queue that is flushed
// to global memory with one atomicAdd per ~QCAP pairs.
// position in the cell-sorted arrays of candidate k of a home cell (five contiguous ranges, see k_search);
// dN = start of range N minus the number of candidates before it.  Straight-line selects, no branches.
__device__ __forceinline__ int cand_pos(int k, int o1, int o2, int o3, int o4, int d0, int d1, int d2, int d3, int d4) {
    int d = d0;
    d = (k >= o1) ? d1 : d;
    d = (k >= o2) ? d2 : d;
    d = (k >= o3) ? d3 : d;
    d = (k >= o4) ? d4 : d;
    return d + k;
}
// (v << 1) | (d <= thr) in TWO instructions: the compare writes its lane mask to an SGPR pair, which is the carry-in of
// v + v + carry — the select / shift / or the compiler makes of the C expression were a third of the distance loop's VALU work
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t shl1_or_le(uint32_t v, float d, float thr) {   // (v << 1) | (d <= thr)
    uint32_t r;
    unsigned long long m;
    asm("v_cmp_le_f32_e64 %1, %2, %3\n\tv_addc_co_u32_e64 %0, %1, %4, %4, %1" : "=v"(r), "=&s"(m) : "v"(d), "v"(thr), "v"(v));
    return r;
}
__device__ __forceinline__ unsigned long long wave_sum_u32(unsigned int v) {
    unsigned long long s = v;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    return s;
}

// the XCD a wave runs on (HW_REG_XCC_ID[3:0]).  blockIdx % 8 is that only up to a rotation that differs from launch to launch
// (tools/micro/xcc_queue.hip: block b of a 768-block launch ran on XCD (b + 7) % 8)
__device__ __forceinline__ int xcc_id() { return (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u); }

__global__ void k_xcc_probe(int* __restrict__ out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

// (developer builds, tools/search_trace.py: per block of k_search<MODE_CONTACTS> {start, end of the cell loops, end} in
// s_memrealtime ticks (100 MHz), the XCD and the hardware id of the block's first wave — g_search_trace, declared above)

// Runs of tiles of equal WEIGHT for the nb blocks of k_search (see there).  A cell weighs its atoms + cell_w16 / 16 (its set-up:
// range bounds, claim, the loads of its records — a block of a face of the box, where cells are a third emptier, must not get
// that many more cells for its atoms): blk_tile[b] = the tile of the cell where the running weight passes b / nb of the total
// (binary search in the start table), blk_tile[nb] = ntile.
__global__ __launch_bounds__(256) void k_balance_blocks(GridDesc g, const int* __restrict__ start, int nb, int tx, int cell_w16, int* __restrict__ blk_tile) {
    const int ntx = (g.nx + tx - 1) / tx, ntile = ntx * g.ny * g.nz;
    const long long T = 16ll * start[g.ncell] + (long long)cell_w16 * g.ncell;
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b <= nb; b += gridDim.x * blockDim.x) {
        if (b == nb || start[g.ncell] <= 0) { blk_tile[b] = (b == 0) ? 0 : ntile; continue; }
        const long long p = (long long)b * T / nb;
        int lo = 0, hi = g.ncell;                 // the last cell c whose running weight (before it) is <= p
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (16ll * start[mid] + (long long)cell_w16 * mid <= p) lo = mid; else hi = mid;
        }
        const int row = lo / g.nx;
        blk_tile[b] = (b == 0) ? 0 : row * ntx + (lo - row * g.nx) / tx;
    }
}

// ... and for blocks that take runs of cell-sorted ATOMS (sparse or clumped grids): equal runs of atoms are not equal work when the
// density varies — a 96 k-atom chain folded onto itself gave its CUs 4 k ... 102 k distance tests each (median 49 k), and the
// kernel lasts as long as the fullest one.  k_cell_weights prices every cell (its units, their chunks, its distance tests: the
// candidates of a cell are the five ranges of the half stencil), a scan turns that into running weights, k_balance_atoms finds for every block the atom position where the running weight passes b / nb of the total
// (binary search over the cells, then in proportion inside the cell).  A hint like the runs of tiles: any partition is correct.
__global__ __launch_bounds__(256) void k_cell_weights(GridDesc g, const int* __restrict__ start, int w_unit, int w_chunk, int w_test8, int* __restrict__ cw) {
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < g.ncell; c += gridDim.x * blockDim.x) {
        const int n = start[c + 1] - start[c];
        int w = 0;
        if (n > 0) {
            const int row = c / g.nx, cx = c - row * g.nx;
            const int cz = row / g.ny, cy = row - cz * g.ny;
            long long cand = start[row * g.nx + min(cx + 2, g.nx)] - start[c];      // home pencil: own cell (later entries) and cx + 1
#pragma unroll
            for (int r = 1; r < 5; ++r) {
                const int dy = (r == 1) ? 1 : (r - 3), dz = (r == 1) ? 0 : 1;
                const int y2 = cy + dy, z2 = cz + dz;
                if (y2 < 0 || y2 >= g.ny || z2 >= g.nz) continue;
                const int rb = (z2 * g.ny + y2) * g.nx;
                cand += start[rb + min(cx + 2, g.nx)] - start[rb + max(cx - 1, 0)];
            }
            // in wave instructions, as the ISA of k_search has them: a unit (a home block of 32 against a span of candidates) costs
            // w_unit + w_chunk per chunk of 128 candidates, a distance test w_test8 / 8 (17 per home atom and chunk, the hits' share)
            const long long nhb = (n + HOME_BLOCK - 1) / HOME_BLOCK, nch = (cand + 127) / 128;
            // (a test in a crowded cell costs more than one in a sparse cell: more of them hit, and hits are what stage 2 walks —
            // the test's price grows with the candidates of the cell, + 1 per 256 of them)
            const long long tests = (long long)n * cand;
            const long long v = (nhb * (w_unit + w_chunk * nch) + ((tests * w_test8 * (256 + cand)) >> 11) + 3) >> 2;
            // (the running sum of the weights is a 32-bit scan: a cell's weight is capped so that ncell of them cannot pass 2^31 - 1 —
            // the table k_balance_atoms makes of it must be non-decreasing, or the blocks' runs of home atoms would overlap or leave
            // gaps; typical cells weigh a few thousand, only clumps meet the cap, and a capped clump is still a heavy cell)
            const long long wcap = min((long long)(1 << 24), (long long)INT_MAX / (long long)max(g.ncell, 1));
            w = (int)min(v, max(wcap, 1ll));
        }
        cw[c] = w;
    }
}
__global__ __launch_bounds__(256) void k_balance_atoms(GridDesc g, const int* __restrict__ start, const int* __restrict__ cwp, int nb, int* __restrict__ blk_pos) {
    const int T = start[g.ncell];
    const long long W = cwp[g.ncell];
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b <= nb; b += gridDim.x * blockDim.x) {
        if (b == 0 || b == nb || W <= 0) { blk_pos[b] = (b == 0) ? 0 : (W <= 0 ? (int)((long long)b * T / nb) : T); continue; }
        const long long target = (long long)b * W / nb;
        int lo = 0, hi = g.ncell;                 // the last cell whose running weight (before it) is <= target
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if ((long long)cwp[mid] <= target) lo = mid; else hi = mid;
        }
        const int n = start[lo + 1] - start[lo];
        const long long w = (long long)cwp[lo + 1] - cwp[lo];
        const int inside = (w > 0) ? (int)min((long long)n, (target - cwp[lo]) * n / w) : 0;
        blk_pos[b] = max(min(start[lo] + max(inside, 0), T), 0);
    }
}

// TX: x-adjacent home cells per tile (1 or 2; a template parameter: with one cell per tile the column rules below vanish at
// compile time — as a run-time value they cost the small grids 3 us)
template <int MODE, int TX = 1>
__global__ __launch_bounds__(64 * SEARCH_WAVES, SEARCH_MIN_WAVES) void k_search(GridDesc g, const int* __restrict__ start,
                                                               const float4* __restrict__ s_xyzm,
                                                               const int4* __restrict__ s_aux, double r2,
                                                               int include_seq_adj, int count_owned, int2* __restrict__ pairs,
                                                               unsigned long long cap, u64* __restrict__ ctr_pairs,
                                                               u64* __restrict__ ctr_cand, u64* __restrict__ ctr_acc,
                                                               uint8_t* __restrict__ plus, GroupMasks gm, const int* __restrict__ cell_of_pos,
                                                               const int* __restrict__ blk_tile) {
    // ring / amide sets of _make_selection (I:1433-1437) from the residue tags the grid build of this pass left: every thread
    // of the launch takes at most a few (nothing to do when gm is empty)
#if defined(ARP_SEARCH_TRACE) && !defined(ARP_SIFT_TRACE) && !defined(ARP_COMPACT_TRACE)
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
    unsigned long long t_loops = 0;
#endif
    group_masks(gm, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
    __shared__ int2 q[MODE == MODE_MARK ? 1 : SEARCH_WAVES][QCAP];   // (the expansion search queues nothing)
    __shared__ float4 s_hx[SEARCH_WAVES][HOME_BLOCK];   // home atoms of the moment: x, y, z, meta
    __shared__ int4 s_ha[SEARCH_WAVES][HOME_BLOCK];     //                         local id, residue, prev, next
    __shared__ uint16_t s_desc[MODE == MODE_MARK ? 1 : SEARCH_WAVES][DESC_CAP]; // hits of the chunk: candidate slot << 5 | home atom
    __shared__ int s_dn[SEARCH_WAVES];
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    // XCD-aware remap: blocks that land on one XCD (b % 8) walk a contiguous run of cells,
    // so the 26 neighbour cells of a home cell are mostly served by that XCD's L2.
    const int nb = gridDim.x;
    const int per = nb >> 3;
    int vb = blockIdx.x;
    if (per > 0 && blockIdx.x < per * 8) vb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    // The unit of the walk is a TILE of one or two x-adjacent home cells (two: the home atoms of both cells share one
    // candidate set — the pencils [cx - 1, cx + 2] instead of twice [cx - 1, cx + 1] — so the 32-slot home block and the
    // 128-slot candidate chunk run fuller, and the set-up of a unit is paid once for two cells).  tile t = row * ntx + cx / 2.
    const int ntx = (g.nx + TX - 1) / TX;
    const int ntile = ntx * g.ny * g.nz;
    const int tiles_per_block = (ntile + nb - 1) / nb;
    int blk_begin = vb * tiles_per_block;
    int c_end = min((vb + 1) * tiles_per_block, ntile);
    auto tile_of_cell = [&](int c) -> int { const int row = c / g.nx; return row * ntx + (c - row * g.nx) / TX; };
    int h_lo = 0, h_hi = INT_MAX;      // positions of the cell-sorted array whose atoms this block takes as HOME atoms
    if (!cell_of_pos && blk_tile) {
        // Dense grids: equal runs of CELLS give the blocks of the box's faces half the work of the others (a 26^3 grid whose last
        // layers are two thirds full: 10 k ... 41 k tests per CU).  blk_tile holds runs of tiles with equal numbers of ATOMS, worked
        // out once per grid from its start table (k_balance_blocks) — a hint: any partition of the tiles is a correct one.
        blk_begin = min(max(blk_tile[vb], 0), ntile);
        c_end = min(max(blk_tile[vb + 1], blk_begin), ntile);
    }
    if (cell_of_pos) {
        // Sparse or clumped grids — a protein in its bounding box, the selection_plus of a ligand inside a large structure, a
        // batch with its gaps, a chain folded onto itself — put their atoms into a fraction of the cells, and equal runs of
        // CELLS gave a few blocks all the work (a 96 k-atom chain: 280 us where uniform atoms of the same number take 28).
        // Here every block takes an equal run [p0, p1) of the cell-sorted ATOMS as its home atoms, wherever the cell
        // boundaries are: a cell that straddles p0 or p1 is shared with the neighbouring block, each taking its own home atoms
        // against the cell's whole candidate set (the cell of an atom position is written by the grid build: two dependent
        // load rounds before the first cell).
        const long long T = start[g.ncell];
        int p0 = (int)((long long)vb * T / nb), p1 = (int)((long long)(vb + 1) * T / nb);
        if (blk_tile) {      // runs of atoms of equal WEIGHT (k_balance_atoms): a hint from an earlier pass over this grid
            // (the table is non-decreasing by construction; the first run begins at 0 and the last one ends at T whatever it says, so
            // that the runs tile [0, T) for any such table)
            p0 = (vb == 0) ? 0 : min(max(blk_tile[vb], 0), (int)T);
            p1 = (vb == nb - 1) ? (int)T : min(max(blk_tile[vb + 1], p0), (int)T);
        }
        if (p1 > p0) {
            blk_begin = tile_of_cell(cell_of_pos[p0]);
            c_end = tile_of_cell(cell_of_pos[p1 - 1]) + 1;
            h_lo = p0; h_hi = p1;
        } else {
            blk_begin = c_end = 0;
        }
    }

    int qn = 0;
    unsigned int n_cand = 0, n_acc = 0;   // per lane; reduced over the wave at the end
    const float r2_lo = (float)(r2 * (1.0 - 1e-5)), r2_hi = (float)(r2 * (1.0 + 1e-5));

    // output segment of this block (cap = capacity of ONE segment)
    // (the XCD the block runs on: the sift blocks of that XCD consume the segment, and the records its pairs point at are in that L2)
    const int seg = (MODE == MODE_CONTACTS) ? xcc_id() : 0;
    u64* const seg_ctr = ctr_pairs + seg * CTR_LINE;
    int2* const seg_pairs = pairs + (size_t)seg * cap;
    auto flush = [&]() {
        __builtin_amdgcn_wave_barrier();
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(seg_ctr, (unsigned long long)qn);
        base = __shfl(base, 0);
        for (int k = lane; k < qn; k += 64)
            if (base + k < cap) seg_pairs[base + k] = q[w][k];
        __builtin_amdgcn_wave_barrier();
        qn = 0;
    };

    // The waves of a block CLAIM its work one HOME BLOCK at a time (a cell's home atoms in blocks of 32: nearly always the
    // whole cell).  A cell's work goes with the square of its atom count, and with a fixed three cells per wave the slowest wave
    // of a block had twice the mean; a cell with hundreds of atoms (a clump: the 96 k-atom chain of tools/small_bench.py) was a
    // serial chain on one wave, whatever the other waves did.  What made claiming dear is one start-table round trip per
    // claim; here the block looks at its cells together:
    //   A  runs of more than 64 cells only, 512 cells per step, one per thread: which hold atoms (ONE load round) ->
    //      compacted list in LDS.  A protein in its bounding box, or a batch of structures with the gaps between them
    //      (arp_set_batch), leaves most cells empty;
    //   B  64 cells at a time, thread 8 * i + r: the bounds of range r of cell i (range 0 = home pencil [own cell, cx + 1],
    //      ranges 1..4 = the forward pencils [cx - 1, cx + 1], r = 5: end of the home cell) -> LDS; every wave then
    //      numbers the home blocks of the 64 cells (lane = cell, a prefix sum over the lanes);
    //   C  every wave takes the next home block with an LDS atomic until none is left.
    __shared__ int s_occ[64 * SEARCH_WAVES];
    // per tile: [0..4] start of ranges 0..4, [5] end of the home atoms, [6] end of the FIRST home cell, [7..11] length of ranges
    // 0..4, [12..15] where column cx begins in forward rows 1..4 (candidates before it are column cx - 1: neighbours of the first
    // home cell only), [16..19] where column cx + TX begins there (candidates from it on: neighbours of the last home cell only),
    // [20] start of the LAST home cell
    __shared__ __attribute__((aligned(16))) int s_info[8 * SEARCH_WAVES][24];
    __shared__ int s_nocc, s_next;
    __shared__ int s_upre[65];
    constexpr int CLAIM_CELLS = 8 * SEARCH_WAVES;
    constexpr int CAND_SPAN = 1024;      // candidates of a tile one unit tests its home block against (a multiple of 128)
    auto tile_cell0 = [&](int t, int& cx, int& cy, int& cz) -> int {      // first cell of tile t
        const int row = t / ntx;
        cx = (t - row * ntx) * TX;
        cz = row / g.ny;
        cy = row - cz * g.ny;
        return row * g.nx + cx;
    };
    for (int win = blk_begin; win < c_end; win += 64 * SEARCH_WAVES) {
     const bool listed = c_end - win > CLAIM_CELLS;      // (a short run: every tile of it is looked at, stage A would be a round trip for nothing)
     int nocc = min(c_end - win, CLAIM_CELLS);
     if (listed) {
        __syncthreads();
        if (threadIdx.x == 0) s_nocc = 0;
        __syncthreads();
        const int tile = win + (int)threadIdx.x;
        bool oc = false;
        if (tile < c_end) {
            int cx, cy, cz;
            const int c0 = tile_cell0(tile, cx, cy, cz);
            oc = start[c0 + min(TX, g.nx - cx)] != start[c0];
        }
        const unsigned long long m = __ballot(oc);
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(&s_nocc, __popcll(m));
        base = __shfl(base, 0);
        if (oc) s_occ[base + __popcll(m & ((1ull << lane) - 1ull))] = tile;
        __syncthreads();
        nocc = s_nocc;
     }
     for (int k0 = 0; k0 < nocc; k0 += CLAIM_CELLS) {
      const int nk = min(CLAIM_CELLS, nocc - k0);
      {
        const int i = (int)threadIdx.x >> 3, r = (int)threadIdx.x & 7;
        if (i < nk && r < 5) {
            int cx, cy, cz;
            const int c0 = tile_cell0(listed ? s_occ[k0 + i] : win + i, cx, cy, cz);
            const int ncol = min(TX, g.nx - cx);                      // home cells of the tile (1 at the end of a row of odd length)
            if (r == 0) {
                const int rowbase = c0 - cx;
                const int hs_ = start[c0], mid = start[c0 + 1], he_ = start[c0 + ncol], end0 = start[rowbase + min(cx + TX + 1, g.nx)];
                s_info[i][0] = hs_;
                s_info[i][5] = he_;
                s_info[i][6] = (ncol > 1) ? mid : he_;               // end of the first home cell
                s_info[i][20] = (ncol > 1) ? mid : hs_;              // start of the last home cell
                s_info[i][7] = end0 - hs_;
            } else {
                const int dy = (r == 1) ? 1 : (r - 3);
                const int dz = (r == 1) ? 0 : 1;
                const int y2 = cy + dy, z2 = cz + dz;
                int js = 0, len = 0, ca = 0, cb = 0;
                if (y2 >= 0 && y2 < g.ny && z2 < g.nz) {
                    const int rowbase = (z2 * g.ny + y2) * g.nx;
                    js = start[rowbase + max(cx - 1, 0)];
                    len = start[rowbase + min(cx + TX + 1, g.nx)] - js;
                    ca = start[rowbase + cx];
                    cb = start[rowbase + min(cx + TX, g.nx)];
                }
                s_info[i][r] = js;
                s_info[i][7 + r] = len;
                s_info[i][11 + r] = ca;
                s_info[i][15 + r] = cb;
            }
        }
        if (threadIdx.x == 0) s_next = 0;
      }
      __syncthreads();
      // home blocks of cell `lane`, and how many there are in the cells before it (first wave; kept in LDS: the kernel has no
      // vector register to spare for it)
      if (w == 0) {
        // (a unit = one home block against at most CAND_SPAN candidates of its cell: a clump of hundreds of atoms with thousands of
        // candidates is many units, for many waves)
        int nhb = (lane < nk) ? max(min(s_info[lane][5], h_hi) - max(s_info[lane][0], h_lo) + HOME_BLOCK - 1, 0) / HOME_BLOCK : 0;
        if (lane < nk) nhb *= max((s_info[lane][7] + s_info[lane][8] + s_info[lane][9] + s_info[lane][10] + s_info[lane][11] + CAND_SPAN - 1) / CAND_SPAN, 1);
        int incl = nhb;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        s_upre[lane] = incl - nhb;
        if (lane == 63) s_upre[64] = incl;
      }
      __syncthreads();
      const int n_units = __builtin_amdgcn_readfirstlane(s_upre[64]);
#pragma unroll 1
      for (;;) {
        int u = 0;
        if (lane == 0) u = atomicAdd(&s_next, 1);
        u = __builtin_amdgcn_readfirstlane(u);
        if (u >= n_units) break;
        const int ci = __popcll(__ballot(lane < nk && s_upre[lane] <= u)) - 1;      // the last cell whose first home block is not after u
        const int uu = u - __builtin_amdgcn_readfirstlane(s_upre[ci]);
        const int4 ia = *reinterpret_cast<const int4*>(&s_info[ci][0]), ib = *reinterpret_cast<const int4*>(&s_info[ci][4]),
                   ic = *reinterpret_cast<const int4*>(&s_info[ci][8]);
        int4 id = make_int4(0, 0, 0, 0), ie = make_int4(0, 0, 0, 0);
        if (TX > 1) { id = *reinterpret_cast<const int4*>(&s_info[ci][12]); ie = *reinterpret_cast<const int4*>(&s_info[ci][16]); }
        const int hs = __builtin_amdgcn_readfirstlane(ia.x);
        const int he_all = __builtin_amdgcn_readfirstlane(ib.y);            // end of the tile's home atoms = where column cx + TX begins in the home row
        const int he = min(he_all, h_hi);                                   // (home atoms of this block only)
        const int n_first = (TX > 1) ? __builtin_amdgcn_readfirstlane(ib.z) - hs : INT_MAX;      // home atoms [0, n_first) are in the first home cell,
        const int n_last0 = (TX > 1) ? __builtin_amdgcn_readfirstlane(s_info[ci][20]) - hs : 0;  // [n_last0, ...) in the last one (the same cell when the tile has one)
        const int js0 = hs, js1 = __builtin_amdgcn_readfirstlane(ia.y), js2 = __builtin_amdgcn_readfirstlane(ia.z),
                  js3 = __builtin_amdgcn_readfirstlane(ia.w), js4 = __builtin_amdgcn_readfirstlane(ib.x);
        const int o1 = __builtin_amdgcn_readfirstlane(ib.w);      // candidates [0, o1) come from range 0
        const int o2 = o1 + __builtin_amdgcn_readfirstlane(ic.x);
        const int o3 = o2 + __builtin_amdgcn_readfirstlane(ic.y);
        const int o4 = o3 + __builtin_amdgcn_readfirstlane(ic.z);
        const int total = o4 + __builtin_amdgcn_readfirstlane(ic.w);
        // column boundaries of the five ranges (range 0 has no column cx - 1; its column cx + TX begins at he_all)
        const int ca1 = __builtin_amdgcn_readfirstlane(id.x), ca2 = __builtin_amdgcn_readfirstlane(id.y),
                  ca3 = __builtin_amdgcn_readfirstlane(id.z), ca4 = __builtin_amdgcn_readfirstlane(id.w);
        const int cb1 = __builtin_amdgcn_readfirstlane(ie.x), cb2 = __builtin_amdgcn_readfirstlane(ie.y),
                  cb3 = __builtin_amdgcn_readfirstlane(ie.z), cb4 = __builtin_amdgcn_readfirstlane(ie.w);
        const int nks = max((total + CAND_SPAN - 1) / CAND_SPAN, 1);      // candidate spans of this cell (1 unless it is a clump)
        const int hbi = (nks == 1) ? uu : uu / nks;
        const int kb_begin = (uu - hbi * nks) * CAND_SPAN, kb_end = min(total, kb_begin + CAND_SPAN);
        {   // home atoms [hb, hb + 32) of the cell: one bit each in the per-lane hit masks
            const int hb = max(hs, h_lo) + HOME_BLOCK * hbi;
            const int hcount = min(HOME_BLOCK, he - hb);
            const bool hvalid = lane < hcount;
            const int hpos = min(hb + lane, he - 1);   // (clamped: no branch around the loads; lanes >= hcount are never read)
            const float4 hreg = s_xyzm[hpos];
            int4 hauxreg;
            if (MODE != MODE_PAIRS) hauxreg = s_aux[hpos];
            else hauxreg = make_int4(s_aux[hpos].x, 0, 0, 0);
            // the hit stage below addresses home atoms by a per-lane index: keep them in LDS as well
            __builtin_amdgcn_wave_barrier();
            if (lane < HOME_BLOCK) { s_hx[w][lane] = hreg; s_ha[w][lane] = hauxreg; }
            __builtin_amdgcn_wave_barrier();
            // expansion search: which home atoms are selected (lane = home atom)
            const uint32_t m_hvalid = (uint32_t)__ballot(hvalid);
            const uint32_t m_selh = (MODE == MODE_MARK) ? (uint32_t)__ballot(hvalid && (__float_as_uint(hreg.w) & M_SEL)) : 0u;
#pragma unroll 1
            for (int kb = kb_begin; kb < kb_end; kb += 128) {  // the ~87 candidates of this cell, two per lane
                // second candidate of the lane in REVERSE order: the candidates most likely to hit come first in the list (home
                // pencil, then the pencils of the same layer), and a lane holding two of those walks twice as many hits in
                // stage 2 as the rest — whose iteration count is the fullest lane's.  Lane l pairs candidate l with 127 - l.
                const int k0 = kb + lane, k1 = kb + 127 - lane;
                const bool valid0 = k0 < total, valid1 = k1 < total;
                // invalid lanes read the first home atom instead of branching around the loads: the four loads of a
                // chunk leave back to back (their lanes never test, kk = -1 below)
                const int j0 = valid0 ? cand_pos(k0, o1, o2, o3, o4, js0, js1 - o1, js2 - o2, js3 - o3, js4 - o4) : hs;
                const int j1 = valid1 ? cand_pos(k1, o1, o2, o3, o4, js0, js1 - o1, js2 - o2, js3 - o3, js4 - o4) : hs;
                const float4 x0 = s_xyzm[j0];
                const float4 x1 = s_xyzm[j1];
                int4 a0, a1;
                if (MODE != MODE_PAIRS) {
                    a0 = s_aux[j0];
                    a1 = s_aux[j1];
                } else {
                    a0 = make_int4(s_aux[j0].x, 0, 0, 0);
                    a1 = make_int4(s_aux[j1].x, 0, 0, 0);
                }
                const uint32_t mj0 = __float_as_uint(x0.w), mj1 = __float_as_uint(x1.w);
                // Inside the home pencil only later entries of the sorted array (j > h) pair up: candidate k of
                // range 0 is position hs + k, so it is tested against home atom h iff k > h - hs.  Other valid
                // candidates are always tested, invalid lanes never.
                const int kk0 = valid0 ? ((k0 < o1) ? k0 : INT_MAX) : -1;
                const int kk1 = valid1 ? ((k1 < o1) ? k1 : INT_MAX) : -1;
                const bool selj0 = mj0 & M_SEL, selj1 = mj1 & M_SEL;
                if (MODE == MODE_MARK) {
                    // Expansion search (I:1420-1424): a pair changes selection_plus only if exactly ONE of its atoms is
                    // selected (both selected: already members; neither: not concerned), so only those pairs are tested.
                    const unsigned long long mv0 = __ballot(valid0), mv1 = __ballot(valid1);
                    const unsigned long long ms0 = __ballot(valid0 && selj0), ms1 = __ballot(valid1 && selj1);
                    if ((m_selh == m_hvalid && ms0 == mv0 && ms1 == mv1) || (m_selh == 0 && ms0 == 0 && ms1 == 0)) continue;
                }
                const int t0 = hb - hs;
                // Which (home atom, candidate) pairs are tested — bit hh = home atom t0 + hh: a candidate k of the home pencil meets
                // home atom t iff k > t; a candidate of column cx - 1 meets the atoms of the FIRST home cell only, one of column
                // cx + TX those of the LAST one (the cells of a pair must be neighbours: SURVEY 8d's candidate pair); every other
                // candidate meets every home atom.  The masks do not depend on the distances: the tests are counted from them.
                uint32_t te0, te1;
                {
                    auto lowmask = [](int c) -> uint32_t { return c >= 32 ? 0xFFFFFFFFu : (c <= 0 ? 0u : ((1u << c) - 1u)); };
                    const uint32_t m_first = lowmask(n_first - t0), m_last = ~lowmask(n_last0 - t0);
                    auto tested = [&](int k, int kk, int j) -> uint32_t {
                        if (kk < 0) return 0u;
                        uint32_t m = (kk == INT_MAX) ? 0xFFFFFFFFu : lowmask(kk - t0);
                        if (TX > 1) {
                            int ca = INT_MIN, cb = he_all;                  // range 0
                            ca = (k >= o1) ? ca1 : ca; cb = (k >= o1) ? cb1 : cb;
                            ca = (k >= o2) ? ca2 : ca; cb = (k >= o2) ? cb2 : cb;
                            ca = (k >= o3) ? ca3 : ca; cb = (k >= o3) ? cb3 : cb;
                            ca = (k >= o4) ? ca4 : ca; cb = (k >= o4) ? cb4 : cb;
                            m &= (j < ca) ? m_first : 0xFFFFFFFFu;
                            m &= (j >= cb) ? m_last : 0xFFFFFFFFu;
                        }
                        return m;
                    };
                    te0 = tested(k0, kk0, j0);
                    te1 = tested(k1, kk1, j1);
                }
                if (MODE != MODE_MARK && !count_owned) n_cand += __popc(te0 & m_hvalid) + __popc(te1 & m_hvalid);
                // ---- stage 1: distance tests only.  Bit hh of lo/hi = candidate within r2_lo / r2_hi of home atom hh.
                // float32 pre-filter: |d2f - d2| <= 4e-7 * d2 (three rounded differences, three rounded squares, two
                // rounded sums), so outside the +-1e-5 band the float32 answer IS the float64 answer.
                uint32_t lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0;
                const v2f cx = {x0.x, x1.x}, cy = {x0.y, x1.y}, cz = {x0.z, x1.z};
#pragma unroll 1
                for (int hh = hcount - 1; hh >= 0; --hh) {
                    // broadcast the home atom from lane hh (v_readlane, no memory traffic)
                    const float hx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(hreg.x), hh));
                    const float hy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(hreg.y), hh));
                    const float hz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(hreg.z), hh));
                    const v2f dx = hx - cx, dy = hy - cy, dz = hz - cz;          // both candidates of the lane at once (v_pk_*)
                    const v2f dd = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));   // any rounding order fits the 4e-7 bound
                    const float d0 = dd.x, d1 = dd.y;
                    // bit hh of the masks (the loop runs downwards and shifts left); which pairs are to be tested at all does
                    // not depend on the distance and is applied to the finished masks below
                    lo0 = shl1_or_le(lo0, d0, r2_lo);
                    hi0 = shl1_or_le(hi0, d0, r2_hi);
                    lo1 = shl1_or_le(lo1, d1, r2_lo);
                    hi1 = shl1_or_le(hi1, d1, r2_hi);
                    if (count_owned) {
                        // sharded run: a boundary pair is tested on two ranks; count it for the owner of its bgn atom only
                        const bool tb0 = (te0 >> hh) & 1u, tb1 = (te1 >> hh) & 1u;
                        const int lh = __builtin_amdgcn_readlane(hauxreg.x, hh);
                        const uint32_t mh0 = __builtin_amdgcn_readlane(__float_as_uint(hreg.w), hh);
                        n_cand += (unsigned)(tb0 && (((lh < a0.x) ? mh0 : mj0) & M_HOME));
                        n_cand += (unsigned)(tb1 && (((lh < a1.x) ? mh0 : mj1) & M_HOME));
                    }
                }
                {   // only the pairs that are to be tested count as hits
                    if (MODE == MODE_MARK) {   // ... and, for the expansion, only pairs with exactly one selected atom
                        te0 &= selj0 ? ~m_selh : m_selh;
                        te1 &= selj1 ? ~m_selh : m_selh;
                        n_cand += __popc(te0 & m_hvalid) + __popc(te1 & m_hvalid);
                    }
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
                        const float4 hv = s_hx[w][hh];
                        if (num::dist2_kd(num::d3{(double)hv.x, (double)hv.y, (double)hv.z}, p0) <= r2) lo0 |= 1u << hh;
                    }
                    while (band1) {
                        const int hh = __ffs(band1) - 1;
                        band1 &= band1 - 1;
                        const float4 hv = s_hx[w][hh];
                        if (num::dist2_kd(num::d3{(double)hv.x, (double)hv.y, (double)hv.z}, p1) <= r2) lo1 |= 1u << hh;
                    }
                }
                n_acc += __popc(lo0) + __popc(lo1);
                if (MODE == MODE_MARK) {
                    // interactions.py:1420-1424: either atom selected -> both join selection_plus (every lane walks its own hits)
                    unsigned long long lo = (unsigned long long)lo0 | ((unsigned long long)lo1 << 32);
                    while (lo != 0) {
                        const int bit = __ffsll((long long)lo) - 1;
                        lo &= lo - 1ull;
                        const int hh = bit & 31;
                        const bool use1 = bit >= 32;
                        const uint32_t mh = __float_as_uint(s_hx[w][hh].w);
                        if ((mh | (use1 ? mj1 : mj0)) & M_SEL) {
                            plus[use1 ? a1.x : a0.x] = 1;
                            plus[s_ha[w][hh].x] = 1;
                        }
                    }
                    continue;
                }
                // ---- stage 2: residue filters, orientation and queueing of the hits.
                // A lane has 0-5 hits, ~1.1 on average: walking them lane by lane kept a quarter of the wave busy in the
                // ~70 instructions a hit costs.  Instead the hits are first COMPACTED: every lane drops a 16-bit descriptor
                // {candidate slot, home atom} per hit into a per-wave LDS list (positions from one LDS atomicAdd per lane), and
                // the filters then run on 64 descriptors at a time with full lanes; a lane fetches its candidate's operands
                // from the registers of the lane that holds them (ds_bpermute: no LDS storage, the kernel keeps 3 blocks per CU).
                __builtin_amdgcn_wave_barrier();
                uint32_t l0 = lo0, l1 = lo1;
                for (;;) {
                    const int c = __popc(l0) + __popc(l1);
                    if (!__any(c != 0)) break;
                    if (lane == 0) s_dn[w] = 0;
                    __builtin_amdgcn_wave_barrier();
                    int pos = DESC_CAP;
                    if (c != 0) pos = atomicAdd(&s_dn[w], c);                   // (LDS; the order of the descriptors does not matter)
                    int room = min(max(DESC_CAP - pos, 0), c);                  // hits beyond the list wait for the next round
                    while (room > 0) {
                        const bool from0 = l0 != 0;
                        const uint32_t m = from0 ? l0 : l1;
                        const int bit = __ffs((int)m) - 1;
                        const uint32_t m2 = m & (m - 1u);
                        l0 = from0 ? m2 : l0;
                        l1 = from0 ? l1 : m2;
                        s_desc[w][pos] = (uint16_t)((((from0 ? 0 : 64) + lane) << 5) | bit);
                        ++pos;
                        --room;
                    }
                    __builtin_amdgcn_wave_barrier();
                    const int T = min(s_dn[w], DESC_CAP);
                    for (int r0 = 0; r0 < T; r0 += 64) {
                        const bool has = r0 + lane < T;
                        const unsigned dsc = s_desc[w][min(r0 + lane, DESC_CAP - 1)];
                        const int hh = dsc & 31, cidx = (dsc >> 5) & 127;
                        // (slot l < 64 = first candidate of lane l, 64 + l its second one)
                        const int src = (cidx & 63) << 2;
                        const bool second = cidx >= 64;
                        auto fetch = [&](int v0, int v1) -> int {
                            const int f0 = __builtin_amdgcn_ds_bpermute(src, v0), f1 = __builtin_amdgcn_ds_bpermute(src, v1);
                            return second ? f1 : f0;
                        };
                        const uint32_t mj = (uint32_t)fetch((int)mj0, (int)mj1);
                        const int j = fetch(j0, j1);
                        const int h = hb + hh;
                        bool pass = has;
                        int pb, pe;
                        if (MODE == MODE_CONTACTS) {
                            const int4 aj = make_int4(fetch(a0.x, a1.x), fetch(a0.y, a1.y), fetch(a0.z, a1.z), fetch(a0.w, a1.w));
                            const int4 ah = s_ha[w][hh];
                            const uint32_t mh = __float_as_uint(s_hx[w][hh].w);
                            // canonical orientation: bgn = lower packed index.  Only three things depend on it — which
                            // residue's polypeptide flag is read (I:734 tests res_end twice), whose HOME bit decides
                            // ownership, and the order of the stored positions; the same-residue and sequence-neighbour
                            // tests are symmetric in the two atoms.
                            const bool h_first = ah.x < aj.x;
                            const uint32_t m_bgn = h_first ? mh : mj;
                            const uint32_t m_end = h_first ? mj : mh;
                            pb = h_first ? h : j;
                            pe = h_first ? j : h;
                            // Straight-line filters: interactions.py:729 same residue; 733-741 sequence-adjacent residues — one
                            // of the four links equal <=> the smallest of the four XORs is zero —; ownership: the rank owning
                            // the bgn atom emits the pair
                            const unsigned adj = min(min((unsigned)(ah.w ^ aj.y), (unsigned)(ah.z ^ aj.y)), min((unsigned)(aj.w ^ ah.y), (unsigned)(aj.z ^ ah.y)));
                            const unsigned gate = (include_seq_adj ? 0u : 1u) & ((m_end & M_RES_POLY) ? 1u : 0u) & ((mh & mj & M_RES_HASSEQ) ? 1u : 0u);
                            const unsigned drop = (ah.y == aj.y ? 1u : 0u) | (gate & (adj == 0u ? 1u : 0u)) | ((m_bgn & M_HOME) ? 0u : 1u);
                            pass = has & (drop == 0u);
                        } else {  // MODE_PAIRS: raw search_all, report packed ids (i < j)
                            const int lj = fetch(a0.x, a1.x), lh = s_ha[w][hh].x;
                            pb = min(lh, lj);
                            pe = max(lh, lj);
                        }
                        const unsigned long long mp = __ballot(pass);
                        if (mp) {
                            if (pass) q[w][qn + __popcll(mp & ((1ull << lane) - 1ull))] = make_int2(pb, pe);
                            qn += __popcll(mp);
                            if (qn > QCAP - 64) flush();
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
      }
      __syncthreads();      // (before the cell bounds and the claim counter are written again)
     }
    }
#if defined(ARP_SEARCH_TRACE) && !defined(ARP_SIFT_TRACE) && !defined(ARP_COMPACT_TRACE)
    t_loops = __builtin_amdgcn_s_memrealtime();
#endif
    // End of block: the per-wave queues of the block leave with ONE atomicAdd (single-address atomics
    // run at ~90 per microsecond on this chip, so one per wave would dominate the kernel).
    __shared__ int s_qn[SEARCH_WAVES];
    __shared__ u64 s_base, s_cand[SEARCH_WAVES], s_acc[SEARCH_WAVES];
    const u64 w_cand = wave_sum_u32(n_cand), w_acc = wave_sum_u32(n_acc);
    if (lane == 0) { s_qn[w] = qn; s_cand[w] = w_cand; s_acc[w] = w_acc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        u64 tc = 0, ta = 0;
        for (int k = 0; k < SEARCH_WAVES; ++k) { tot += s_qn[k]; tc += s_cand[k]; ta += s_acc[k]; }
        s_base = (MODE != MODE_MARK && tot > 0) ? atomicAdd(seg_ctr, (u64)tot) : 0;
        const int slot = blockIdx.x & (STAT_SLOTS - 1);
        atomicAdd(ctr_cand + slot * CTR_LINE, tc);
        atomicAdd(ctr_acc + slot * CTR_LINE, ta);
    }
    __syncthreads();
    if (MODE != MODE_MARK && qn > 0) {
        u64 base = s_base;
        for (int k = 0; k < w; ++k) base += (u64)s_qn[k];
        for (int k = lane; k < qn; k += 64)
            if (base + k < cap) seg_pairs[base + k] = q[w][k];
    }
#if defined(ARP_SEARCH_TRACE) && !defined(ARP_SIFT_TRACE) && !defined(ARP_COMPACT_TRACE)
    if (MODE == MODE_CONTACTS && g_search_trace && lane == 0) {
        unsigned long long* t = g_search_trace + ((size_t)blockIdx.x * SEARCH_WAVES + w) * 4;
        t[0] = t_begin; t[1] = t_loops; t[2] = __builtin_amdgcn_s_memrealtime();
        t[3] = (unsigned long long)xcc_id() | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 8) | ((unsigned long long)w_cand << 40);   // HW_REG_HW_ID
    }
#endif
}

// ---- per-pair SIFt --------------------------------------------------------------------
__device__ __forceinline__ num::f3 xyz_of(float4 v) { return {v.x, v.y, v.z}; }

// utils.is_hbond (utils.py:73-93, angle_min 1.57) / is_weak_hbond (utils.py:96-116, 2.27).
// cos_min = cos(angle_min).  Decisions away from a threshold use the squared-quantity shortcuts of
// arp_numerics.h; inside the safety margins the reference's exact operation sequence decides.
__device__ __forceinline__ bool hbond_like(num::f3 donor, const double* __restrict__ hx, int h0, int h1, num::f3 acc,
                                           double acc_vdw, double comp, double angle_min, double cos_min) {
    const num::d3 d = num::to_d3(donor), a = num::to_d3(acc);
    const double thr = 1.2 + acc_vdw + comp;  // config.VDW_RADII['H'] + vdw + comp
    const double thr2 = thr * thr;
    for (int k = h0; k < h1; ++k) {
        const num::d3 h = {hx[3 * (size_t)k], hx[3 * (size_t)k + 1], hx[3 * (size_t)k + 2]};
        const num::d3 v = num::sub(h, a);
        const double s = num::dot(v, v);                       // np.linalg.norm's sum of squares (FMA chain)
        int near = num::dist_le_fast(s, thr2);
        if (near < 0) near = (sqrt(s) <= thr) ? 1 : 0;         // h_dist <= thr, exact
        if (!near) continue;
        int ok = num::angle_ge_fast(d, h, a, cos_min, 1e-12);
        if (ok < 0) ok = (num::get_angle(d, h, a) >= angle_min) ? 1 : 0;
        if (ok) return true;
    }
    return false;
}

// utils.is_halogen_weak_hbond (utils.py:119-155); sbh = single-bond neighbour of the halogen
__device__ __forceinline__ bool halogen_weak(num::f3 hal, float4 sbh, double hal_vdw, const double* __restrict__ hx,
                                             int h0, int h1, double comp) {
    if (sbh.w == 0.0f) return false;  // utils.py:139-141
    const num::d3 hd = num::to_d3(hal);
    const num::f3 nbr = {sbh.x, sbh.y, sbh.z};
    const double thr = 1.2 + hal_vdw + comp;
    const double thr2 = thr * thr;
    for (int k = h0; k < h1; ++k) {
        const num::d3 h = {hx[3 * (size_t)k], hx[3 * (size_t)k + 1], hx[3 * (size_t)k + 2]};
        const num::d3 v = num::sub(hd, h);
        const double s = num::dot(v, v);
        int near = num::dist_le_fast(s, thr2);
        if (near < 0) near = (sqrt(s) <= thr) ? 1 : 0;
        if (!near) continue;
        // the reference normalises (nbr - hal) in float32 (utils.py:151): 1e-5 covers that rounding
        int ok = num::angle_in_fast(num::to_d3(nbr), hd, h, ARP_COS_0_52, ARP_COS_2_62, 1e-5);
        if (ok < 0) {
            const double ang = num::get_angle_mixed(nbr, hal, h);
            ok = (0.52 <= ang && ang <= 2.62) ? 1 : 0;
        }
        if (ok) return true;
    }
    return false;
}

// utils.is_xbond (utils.py:158-179); float32 end to end
__device__ __forceinline__ bool xbond(float4 sbd, num::f3 donor, num::f3 acc, int* err) {
    if (sbd.w == 0.0f) {  // utils.py:173 would dereference None
        atomicExch(err, ARP_E_XBOND_NBR);
        return false;
    }
    bool nan_pi;
    const float theta = num::get_angle(num::f3{sbd.x, sbd.y, sbd.z}, donor, acc, nan_pi);
    if (nan_pi) return true;  // np.pi >= 2.09
    return theta >= (float)2.09;
}

// interactions.py:643-691
__device__ __forceinline__ int contact_type(bool bs, bool es, bool bw, bool ew) {
    int ct = 0;
    if (!bs && !es) ct = ARP_CT_INTRA_NON_SELECTION;
    if (bs && es) ct = ARP_CT_INTRA_SELECTION;
    if ((bs && !es) || (es && !bs)) ct = ARP_CT_INTER;
    if ((bs && ew) || (es && bw)) ct = ARP_CT_SELECTION_WATER;
    if ((!bs && ew) || (!es && bw)) ct = ARP_CT_NON_SELECTION_WATER;
    if (bw && ew) ct = ARP_CT_WATER_WATER;
    return ct;
}

// interactions.py:643-691 as a table: entry k (4 bits) = contact_type(k & 1, k & 2, k & 4, k & 8)
__host__ __device__ constexpr int contact_type_c(bool bs, bool es, bool bw, bool ew) {
    int ct = 0;
    if (!bs && !es) ct = ARP_CT_INTRA_NON_SELECTION;
    if (bs && es) ct = ARP_CT_INTRA_SELECTION;
    if ((bs && !es) || (es && !bs)) ct = ARP_CT_INTER;
    if ((bs && ew) || (es && bw)) ct = ARP_CT_SELECTION_WATER;
    if ((!bs && ew) || (!es && bw)) ct = ARP_CT_NON_SELECTION_WATER;
    if (bw && ew) ct = ARP_CT_WATER_WATER;
    return ct;
}
__host__ __device__ constexpr unsigned long long contact_type_table() {
    unsigned long long t = 0;
    for (int k = 0; k < 16; ++k) t |= (unsigned long long)contact_type_c(k & 1, k & 2, k & 4, k & 8) << (4 * k);
    return t;
}
// float32 upper bound of the reach of a hydrogen test against an atom of radius vdw (U:86, 109, 145): 1.2 + vdw + comp plus
// the longest atom - hydrogen distance of the structure; rounded up, so `d > reach` implies the same in float64
__device__ __forceinline__ float reach_float(double vdw, double comp, double h_slack) {
    return (float)((1.2 + vdw + comp + h_slack) * (1.0 + 1e-6));
}

// One thread per accepted pair (full 64-lane occupancy for the divergent chemistry).
#ifndef SIFT_MIN_WAVES
#define SIFT_MIN_WAVES 4   // waves per SIMD the register allocator must leave room for (sweep in profiles/README.md)
#endif
// Arrays the per-pair kernel reads by LOCAL ATOM ID beside the staged records (stage B and the rare branches of stage A)
struct SiftSide {
    const double2* rad_tab;   // the structure's distinct {vdw, cov} pairs: the first 16 make the threshold table
    const double2* rad;       // {vdw, cov} of every atom
    const float4* xyz;        // uploaded coordinates (w unused)
    const int* h_off;         // uploaded CSR offsets
    const int* bond_off;
    const float4* sb;         // single-bond heavy neighbour: x, y, z, present
    const float* longest_bond;   // k_longest_bond: [0] longest bond, [1] longest atom - hydrogen distance
};

// One atom of a pair as stage B needs it, gathered by sorted position
struct GeoAtom {
    num::f3 x;
    double vdw;
    int lid, h0, h1;
};
__device__ __forceinline__ GeoAtom geo_atom(float4 xyzm, int lid, int hoff, const double* s_vdw16, const SiftSide& sd) {
    const uint32_t m = __float_as_uint(xyzm.w);
    const unsigned ri = (m >> M_RAD4_SHIFT) & 15u, hc = (m >> M_HCNT_SHIFT) & 3u;
    GeoAtom a;
    a.x = xyz_of(xyzm);
    a.lid = lid;
    a.vdw = s_vdw16[ri];
    if (ri == M_RAD4_ESC) a.vdw = sd.rad[lid].x;
    a.h0 = hoff;
    a.h1 = (m & M_HAS_H) ? hoff + 1 + (int)hc : hoff;
    if ((m & M_HAS_H) && hc == M_HCNT_ESC) a.h1 = sd.h_off[lid + 1];
    return a;
}
// The hydrogen geometry of one pair: the branches in `need` (bit k = branch k of the list in k_sift), run on a
// lane of the task stage.  Returns the SIFt bits they add.
__device__ __forceinline__ uint32_t sift_geometry(const GeoAtom& B, const GeoAtom& E, unsigned need, const double* __restrict__ h_xyz, const SiftSide& sd, double comp) {
    const num::f3 xb = B.x, xe = E.x;
    const double vb = B.vdw, ve = E.vdw;
    const int hb0 = B.h0, hb1 = B.h1, he0 = E.h0, he1 = E.h1;
    unsigned todo = need, res = 0;
    while (todo) {                     // almost always one branch per pair
        const int kind = __ffs(todo) - 1;
        todo &= todo - 1;
        bool r;
        if (kind < 4) {
            const bool donor_b = (kind == 0) || (kind == 3);
            const double amin = (kind < 2) ? 1.57 : 2.27;
            const double cmin = (kind < 2) ? ARP_COS_1_57 : ARP_COS_2_27;
            r = hbond_like(donor_b ? xb : xe, h_xyz, donor_b ? hb0 : he0, donor_b ? hb1 : he1, donor_b ? xe : xb,
                           donor_b ? ve : vb, comp, amin, cmin);
        } else {
            const bool hal_b = kind == 4;
            const float4 sbh = sd.sb[hal_b ? B.lid : E.lid];   // w = 1 when the halogen has a single-bond neighbour
            r = halogen_weak(hal_b ? xb : xe, sbh, hal_b ? vb : ve, h_xyz, hal_b ? he0 : hb0,
                             hal_b ? he1 : hb1, comp);
        }
        res |= (r ? 1u : 0u) << kind;
    }
    uint32_t s = 0;
    if (res & 3u) s |= ARP_S_HBOND;
    // the LAST applicable weak branch decides SIFt[6] (every branch overwrites it)
    if (need & 60u) {
        const int last = 31 - __clz((int)(need & 60u));
        if ((res >> last) & 1u) s |= ARP_S_WEAK_HBOND;
    }
    return s;
}

#ifndef SIFT_TASKQ
#define SIFT_TASKQ 128
#endif
#ifndef SIFT_COUNTED_WAIT
#define SIFT_COUNTED_WAIT 1
#endif
struct SiftArgs {
    const int2* pairs;      // the pair list of the pass: sorted positions (bgn, end), PAIR_SEGS segments of `cap` entries
    const u64* npairs_ptr;  // pairs per segment
    u64 cap;
    const float4* s_xyzm;   // cell-sorted records, two 16-byte columns: x, y, z, meta
    const int4* s_qa;       //   local id, bonded neighbours in other residues
    const int* s_h;         // ... and, for stage B, the index of the atom's first hydrogen
    int nblk[PAIR_SEGS];    // blocks per segment (sum = the sift blocks of the launch, each >= 1): a segment's share goes with its
                            // size in the pass before (all equal when that is not known)
    SiftSide sd;
    const int* bond_idx;
    const double* h_xyz;
    const int* gid;
    double comp;
    int* out_i;
    int* out_j;
    float* out_d;
    uint16_t* out_s;
    uint8_t* out_ct;
    int* err;
    int seg_by_block;       // 1: the segment of a sift block is vblock % 8 instead of the XCD it runs on (a device whose dispatcher does
                            // not deal consecutive blocks round-robin over eight XCDs — a partitioned mode, CU masking: arp_create looks)
};
// Contact records are not read again on the device in the pass that writes them.  When there are more of them than L2 and the
// Infinity Cache keep (1 M atoms: 190 MB), streaming stores let them leave while the kernel runs instead of in the write-back at
// its end (sift 303 -> 254 us); for a structure whose records fit (100 k atoms: 19 MB) they only cost write transactions
// (WRITE_SIZE 20.8 -> 26.7 MB: partial lines are not merged), so the host picks the variant by size.  (A template parameter:
// behind a run-time flag the compiler merges the two stores into a plain one.)
template <int STREAM, typename T>
__device__ __forceinline__ void put_record(T v, T* p) {
    if (STREAM) __builtin_nontemporal_store(v, p);
    else *p = v;
}
// The five stores of a batch of records as ONE statement: exactly five vector-memory instructions, in this order, nothing
// of the compiler's between them — the batch loop counts on it (s_waitcnt vmcnt(5), see sift_body).
template <int STREAM>
__device__ __forceinline__ void put_five(int* pi, int vi, int* pj, int vj, float* pd, float vd, uint8_t* pc, unsigned vc, uint16_t* psf, unsigned vs) {
    if (STREAM)
        asm volatile("global_store_dword %0, %1, off nt\n\tglobal_store_dword %2, %3, off nt\n\tglobal_store_dword %4, %5, off nt\n\t"
                     "global_store_byte %6, %7, off nt\n\tglobal_store_short %8, %9, off nt"
                     :: "v"(pi), "v"(vi), "v"(pj), "v"(vj), "v"(pd), "v"(vd), "v"(pc), "v"(vc), "v"(psf), "v"(vs) : "memory");
    else
        asm volatile("global_store_dword %0, %1, off\n\tglobal_store_dword %2, %3, off\n\tglobal_store_dword %4, %5, off\n\t"
                     "global_store_byte %6, %7, off\n\tglobal_store_short %8, %9, off"
                     :: "v"(pi), "v"(vi), "v"(pj), "v"(vj), "v"(pd), "v"(vd), "v"(pc), "v"(vc), "v"(psf), "v"(vs) : "memory");
}
typedef __attribute__((address_space(3))) void* lds_void_p;
struct SiftShared {
    uint4 tq[4][SIFT_TASKQ];     // {output index, bgn position, end position, sift | need << 16}
    double vdw16[16];            // van der Waals radii of the first radius pairs (stage B)
    float4 thr[256];             // for the first 16 radius pairs, pair by pair: {(float)(cov + cov'), (float)(vdw + vdw'), (float)(vdw + vdw' + comp), reach of the second}
    // per wave: the records of ONE batch of 64 pairs, lane l's in slot l — gathered by global_load_lds while the batch before is evaluated
    float4 xb[4][64], xe[4][64];
    int4 qb[4][64], qe[4][64];
    int2 pr[4][64];              // per wave: the pairs of the batch after that (one global_load_lds of 16 bytes by 32 lanes)
};
// vblock / vgrid: this block's index among the sift blocks of the launch (a multiple of 8 blocks precedes them, so
// vblock % 8 is still the XCD the dispatcher put the block on)
// GID: the records carry global ids (a shard): a template parameter, because a run-time `gid ? gid[b] : b` leaves a full
// s_waitcnt vmcnt(0) behind its branch — in front of the stores, i.e. a wait for the gather that was to travel meanwhile
template <int STREAM, int GID>
__device__ __forceinline__ void sift_body(const SiftArgs& A, int vblock, int vgrid, SiftShared* sh) {
    const int2* __restrict__ pairs = A.pairs;
    const u64* __restrict__ npairs_ptr = A.npairs_ptr;
    const u64 cap = A.cap;
    const float4* __restrict__ s_xyzm = A.s_xyzm;
    const int4* __restrict__ s_qa = A.s_qa;
    const int* __restrict__ s_h = A.s_h;
    const SiftSide sd = A.sd;
    const int* __restrict__ bond_idx = A.bond_idx;
    const double* __restrict__ h_xyz = A.h_xyz;
    const int* __restrict__ gid = A.gid;
    const double comp = A.comp;
    int* __restrict__ out_i = A.out_i;
    int* __restrict__ out_j = A.out_j;
    float* __restrict__ out_d = A.out_d;
    uint16_t* __restrict__ out_s = A.out_s;
    uint8_t* __restrict__ out_ct = A.out_ct;
    int* __restrict__ err = A.err;
    uint4 (*tq)[SIFT_TASKQ] = sh->tq;
    // Two stages per wavefront.  Stage A, one lane per pair: everything of I:715-936 that needs no hydrogen — distance
    // ladder, metal, type-pair flags, halogen bond — and the list of hydrogen-geometry branches the pair needs:
    //   0 is_hbond(bgn, end)   1 is_hbond(end, bgn)          (if / elif, I:804-819)
    //   2 is_weak_hbond(end, bgn)   3 is_weak_hbond(bgn, end)   4 / 5 is_halogen_weak_hbond  (I:857-886)
    // Only ~15 % of the pairs need one, so those lanes are compacted (ballot) into a per-wave LDS task queue and
    // stage B runs the float64 hydrogen loops on 64 queued pairs at a time — full lanes instead of ~10 of 64.
    //
    // The two 32-byte records of a pair are a GATHER (four 16-byte loads per lane at unrelated addresses), and waiting for it in
    // every batch was more than half of this kernel's wave cycles.  Now the gather of batch k + 1 is issued while batch k is
    // evaluated, as global_load_lds: asynchronous, no registers — lane l's four quads land in slot l of four LDS planes of the
    // wave, which the batch reads back as soon as its turn comes (consecutive slots: no bank conflicts).  One buffer is enough:
    // a batch moves its records to registers first, and the next gather is issued behind that.
    // Stage A issues no other global load: the vector-memory operations of an iteration are, in this order, the pairs of the
    // batch after the next one (one load), the gather of the next batch (four copies), the five stores of this batch's records,
    // then whatever stage B does.  Memory operations retire in order on gfx9 (one vmcnt for loads and stores), so "at most
    // five outstanding" at the top of the next iteration means pairs and records have landed while the stores may still be on
    // their way: s_waitcnt vmcnt(5), not (0) — the wave never waits for its own stores.
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#ifdef ARP_SIFT_TRACE
    unsigned long long tr[8] = {__builtin_amdgcn_s_memrealtime(), 0, 0, 0, 0, 0, 0, 0};
#define SIFT_T(k) tr[k] = __builtin_amdgcn_s_memrealtime()
#else
#define SIFT_T(k)
#endif
    // The segment fill counts are read on the device: no host round trip between search and sift.
    // Block b works on the segment of the XCD it runs on — the pairs written by the search blocks that ran on the same XCD,
    // whose atom records are still in that XCD's L2 — and writes its results at the segment's offset.
    // The segments differ in size by a fifth (the search blocks of an XCD cover different parts of the box), the blocks of an
    // XCD are as many as every other's: a segment gets blocks in proportion to its size (nblk) — its own XCD's first, then
    // the ones other XCDs can spare, in a fixed order every block works out for itself.
    // (blocks b with the same b % 8 share an XCD, whichever it is: vblock / 8 numbers the blocks of an XCD — checked once per
    // context by arp_create (k_xcc_probe); where it does not hold the blocks are dealt out by their index alone)
    int sgm = A.seg_by_block ? (vblock & (PAIR_SEGS - 1)) : xcc_id();
    int blk_in_seg = vblock / PAIR_SEGS, blks_of_seg = vgrid / PAIR_SEGS;
    {
        const int B = vgrid / PAIR_SEGS;
        int spare_before = 0;       // spare blocks of the XCDs before this one
#pragma unroll
        for (int q_ = 0; q_ < PAIR_SEGS; ++q_)
            if (q_ < sgm) spare_before += max(B - A.nblk[q_], 0);
        const int mine = min(A.nblk[sgm], B);
        if (blk_in_seg >= mine) {      // a spare block: the k-th spare one of the launch serves the k-th wanted one
            int k = spare_before + (blk_in_seg - mine);
            int found = sgm, idx = blk_in_seg;
#pragma unroll
            for (int q_ = PAIR_SEGS - 1; q_ >= 0; --q_) {
                int want_before = 0;
#pragma unroll
                for (int r_ = 0; r_ < PAIR_SEGS; ++r_)
                    if (r_ < q_) want_before += max(A.nblk[r_] - B, 0);
                const int want = max(A.nblk[q_] - B, 0);
                if (k >= want_before && k < want_before + want) { found = q_; idx = B + (k - want_before); }
            }
            sgm = found; blk_in_seg = idx;
        }
        blks_of_seg = max(A.nblk[sgm], 1);
        if (blk_in_seg >= blks_of_seg) blk_in_seg = blks_of_seg - 1;      // (cannot happen when the shares add up to the launch: a harmless repeat otherwise)
    }
    u64 heads[PAIR_SEGS];
#pragma unroll
    for (int q_ = 0; q_ < PAIR_SEGS; ++q_) heads[q_] = npairs_ptr[q_ * CTR_LINE];
    const float longest_bond = sd.longest_bond[0];
    const double h_slack = (double)sd.longest_bond[1] + 1e-4;   // |H - A| >= |D - A| - h_slack for every hydrogen H of D (margin: float32 distance, roundings)
    const int2* __restrict__ seg_pairs = pairs + (size_t)sgm * cap;
    const long long stride = (long long)blks_of_seg * blockDim.x;
    const long long first = (long long)blk_in_seg * blockDim.x + (threadIdx.x - lane);
    // (a pair beyond the end of the segment is read and ignored: the list is padded — see enqueue_contacts — and the count
    // that says so is still on its way)
    const int2 pr0 = (first + lane < (long long)cap) ? seg_pairs[first + lane] : make_int2(0, 0);
    {   // the three float32 thresholds of the ladder (I:717-718, 756-773: float64 sums, compared as float32) depend on the two
        // radius pairs only: for the common case — both atoms among the first 15 table entries — they are looked up, not computed
        const double2 ra = sd.rad_tab[threadIdx.x >> 4], rb_ = sd.rad_tab[threadIdx.x & 15];
        const double sv = ra.x + rb_.x;
        sh->thr[threadIdx.x] = make_float4((float)(ra.y + rb_.y), (float)sv, (float)(sv + comp), reach_float(rb_.x, comp, h_slack));
        if (threadIdx.x < 16) sh->vdw16[threadIdx.x] = rb_.x;
    }
    float4* const lxb = sh->xb[w];
    float4* const lxe = sh->xe[w];
    int4* const lqb = sh->qb[w];
    int4* const lqe = sh->qe[w];
    int2* const lpr = sh->pr[w];
    // the gather of one batch: lane l's records -> slot l of the wave's four planes
    auto gather = [&](int2 pr) {
        __builtin_amdgcn_global_load_lds(s_xyzm + pr.x, (lds_void_p)lxb, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(s_qa + pr.x, (lds_void_p)lqb, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(s_xyzm + pr.y, (lds_void_p)lxe, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(s_qa + pr.y, (lds_void_p)lqe, 16, 0, 0);
    };
    // the 64 pairs that begin at `at` -> the wave's pair slots (lanes 0 .. 31 bring two pairs each).  The compiler must not see
    // a load of its own here: it would wait for it with vmcnt(0) at the loop's edge — for this batch's stores, that is
    auto fetch_pairs = [&](long long at) {
        if (lane < 32) __builtin_amdgcn_global_load_lds(reinterpret_cast<const int4*>(seg_pairs + at) + lane, (lds_void_p)lpr, 16, 0, 0);
    };
    long long out_base = 0;
#pragma unroll
    for (int q_ = 0; q_ < PAIR_SEGS; ++q_)
        if (q_ < sgm) out_base += (long long)min(heads[q_], cap);
    const long long nseg = (long long)min(heads[sgm], cap);
    // (the first batch's records, before the barrier: they travel while the table is made.  Only pairs of THIS pass are followed:
    // what lies beyond the end of the list are positions of another pass, of another structure perhaps)
    int2 pr_cur = pr0;        // the pairs of the batch at hand (stage B follows them again)
    if (first + lane < nseg) gather(pr0);
    if (first + stride < nseg) fetch_pairs(first + stride);
    __syncthreads();
    SIFT_T(1);
    int tn = 0;
    auto run_tasks = [&](int first_, int count) {   // stage B on tq[w][first_ .. first_ + count)
        if (lane < count) {
            const uint4 t = tq[w][first_ + lane];
            // (by sorted position, like the gather of stage A: the lines are close by in L1 / L2)
            const float4 vb = s_xyzm[t.y], ve = s_xyzm[t.z];
            const int lb = s_qa[t.y].x, le = s_qa[t.z].x, hb = s_h[t.y], he = s_h[t.z];
            const uint32_t add = sift_geometry(geo_atom(vb, lb, hb, sh->vdw16, sd), geo_atom(ve, le, he, sh->vdw16, sd), t.w >> 16, h_xyz, sd, comp);
            put_record<STREAM>((uint16_t)((t.w & 0xFFFFu) | add), out_s + t.x);
        }
        // (the compiler's own count of pending loads ends here: left open, it puts s_waitcnt vmcnt(0) at the edge of the batch
        // loop — where the copies for the next batch and this batch's stores are in flight by design)
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the first batch: everything asked for so far)
    SIFT_T(3);
    for (long long base = first; base < nseg; base += stride) {
        const long long ps = base + lane;   // (wave-uniform trip count: the queue below is a wave-wide affair)
        const bool live = ps < nseg;
        __builtin_amdgcn_wave_barrier();
        // this batch's records: LDS -> registers
        const float4 vb = lxb[lane], ve = lxe[lane];
        const int4 nbr = lqb[lane];       // .x = bgn's local id, .y .z .w = its bonded neighbours in other residues (I:748-757)
        const int e = lqe[lane].x;
        const int2 pr = lpr[lane];        // the next batch's pairs
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // ... and the buffers are free: the pairs of the batch after the next one, the gather of the next batch
        if (base + stride < nseg) {
            if (base + 2 * stride < nseg) fetch_pairs(base + 2 * stride);      // (up to 63 entries beyond the segment's end are read and ignored: pair_segcap)
            if (ps + stride < nseg) gather(pr);
        }
        bool queued = false;
        uint4 task = make_uint4(0u, 0u, 0u, 0u);
        const long long p = out_base + ps;
        int ct = 0;
        const int b_ = nbr.x;
        uint32_t s = 0;
        float d = 0.0f;
        if (live) {
            const num::f3 xb = xyz_of(vb), xe = xyz_of(ve);
            d = num::norm(num::sub(xb, xe));                    // interactions.py:745
            const uint32_t mb = __float_as_uint(vb.w), me = __float_as_uint(ve.w);
            const uint32_t tb = mb & M_TMASK, te = me & M_TMASK;
            // Straight-line integer / mask code from here on: the reference's chains of `if` over the two type masks cost this
            // kernel an exec-mask branch each (~350 VALU + 220 SALU instructions per batch of 64 pairs, and VALU issue is what
            // the SIMDs run out of).  Only what is rare keeps a branch: radii outside the threshold table, a bonded-neighbour
            // list longer than four, the halogen-bond angle.
            const uint32_t bw = (mb / M_WATER) & 1u, ew = (me / M_WATER) & 1u;
            // interactions.py:643-691 (__get_contact_type): the six overriding assignments as a 16-entry table of 4-bit codes,
            // indexed by bgn selected | end selected << 1 | bgn water << 2 | end water << 3
            const uint32_t ct_idx = ((mb / M_SEL) & 1u) | (((me / M_SEL) & 1u) << 1) | (bw << 2) | (ew << 3);
            ct = (int)((contact_type_table() >> (4u * ct_idx)) & 15ull);
            float f_sum_cov, f_sum_vdw, f_vdw_comp;                         // interactions.py:717-718 and the casts of 756-773
            float reach_e, reach_b;    // >= 1.2 + vdw + comp + the longest atom - hydrogen distance: beyond it no hydrogen of the partner reaches (U:86, 109, 145)
            {
                const unsigned rib = (mb >> M_RAD4_SHIFT) & 15u, rie = (me >> M_RAD4_SHIFT) & 15u;
                if (rib != M_RAD4_ESC && rie != M_RAD4_ESC) {
                    const float4 t = sh->thr[rib * 16u + rie];
                    f_sum_cov = t.x; f_sum_vdw = t.y; f_vdw_comp = t.z; reach_e = t.w;
                    reach_b = sh->thr[rie * 16u + rib].w;
                } else {
                    const double2 rb = sd.rad[b_], re = sd.rad[e];   // {vdw, cov}
                    const double sum_vdw = rb.x + re.x;
                    f_sum_cov = (float)(rb.y + re.y); f_sum_vdw = (float)sum_vdw; f_vdw_comp = (float)(sum_vdw + comp);
                    reach_e = reach_float(re.x, comp, h_slack); reach_b = reach_float(rb.x, comp, h_slack);
                }
            }
            // interactions.py:756-773: float32 distance against Python floats -> float32 compare; an exclusive ladder
            s = ARP_S_PROXIMAL;                                   // (selects from the bottom up: no branch per rung)
            s = (d <= f_vdw_comp) ? ARP_S_VDW : s;
            s = (d < f_sum_vdw) ? ARP_S_VDW_CLASH : s;
            s = (d < f_sum_cov) ? ARP_S_CLASH : s;
            // interactions.py:777-783: an hbond acceptor beside a metal
            const uint32_t metal = ((tb & (me >> 12)) | (te & (mb >> 12))) & 1u;        // ARP_T_HBOND_ACCEPTOR = bit 0, M_METAL = bit 12
            const uint32_t s_metal = (d <= (float)2.8) ? metal * ARP_S_METAL_COMPLEX : 0u;
            // The type tests of I:791-921 pair up neighbouring bits of the two masks — (acceptor 0, donor 1), (xbond acceptor 2,
            // donor 3), (weak acceptor 4, weak donor 5), (positive 6, negative 7), (carbonyl O 9, C 10):
            // X bit k = bgn has k + 1 and end has k, Y the same with the two atoms exchanged.
            const uint32_t X = (tb >> 1) & te, Y = (te >> 1) & tb, XY = X | Y;
            const bool in_vc = d <= f_vdw_comp, d35 = d <= (float)3.5;
            // interactions.py:791-819: water rule, else donor / acceptor (if / elif); the water branches set POLAR whatever the distance
            const uint32_t wb = bw & (in_vc ? 1u : 0u), we = ew & (in_vc ? 1u : 0u) & ~wb;
            const uint32_t nowat = (wb | we) ^ 1u;
            const uint32_t hb_w = (wb & (((te & 3u) != 0u) ? 1u : 0u)) | (we & (((tb & 3u) != 0u) ? 1u : 0u));
            const uint32_t c1 = nowat & X & 1u, c2 = nowat & ~X & Y & 1u;
            // interactions.py:857-886: the four weak branches
            const uint32_t n4 = tb & (te >> 5) & 1u, n8 = (tb >> 5) & te & 1u;
            const uint32_t n16 = (tb >> 4) & (mb >> 13) & (((te & 0x22u) != 0u) ? 1u : 0u) & 1u;    // weak acceptor 4, M_HALOGEN 13, donor 1 | weak donor 5
            const uint32_t n32 = (te >> 4) & (me >> 13) & (((tb & 0x22u) != 0u) ? 1u : 0u) & 1u;
            unsigned need_ = c1 | (c2 << 1) | (n4 << 2) | (n8 << 3) | (n16 << 4) | (n32 << 5);
            uint32_t f = hb_w * (ARP_S_HBOND | ARP_S_POLAR);
            f |= (d35 & ((c1 | c2) != 0u)) ? ARP_S_POLAR : 0u;                       // I:806, 814
            f |= (d35 & ((need_ & 60u) != 0u)) ? ARP_S_WEAK_POLAR : 0u;               // I:861, 869, 877, 885
            // interactions.py:898-921: ionic (bit 6 -> 8), carbonyl (9 -> 12), aromatic (11 -> 10), hydrophobic (8 -> 11), each behind its distance
            const uint32_t near4 = ((XY & 0x40u) << 2) | ((tb & te & ARP_T_AROMATIC) >> 1);
            f |= (d <= (float)4.0) ? near4 : 0u;
            f |= (d <= (float)3.6) ? ((XY & 0x200u) << 3) : 0u;
            f |= (tb & te & ARP_T_HYDROPHOBE) << 3;
            {
                // Branches that cannot succeed need no hydrogen loop: the donor has no hydrogen, the halogen no single-bond
                // neighbour (U:139-141), or the partner is beyond the test's reach.  If EVERY applicable branch is such a one the
                // pair gets no hbond / weak hbond bit — what the loops would find — and is not queued; if one is left the task
                // runs with the full set (the last applicable weak branch decides, I:857-886).
                const bool no_hb = !(mb & M_HAS_H), no_he = !(me & M_HAS_H);
                unsigned dead = 0;
                dead |= (no_hb | (d > reach_e)) ? (1u | 8u | 32u) : 0u;               // hydrogens of bgn, target = end
                dead |= (no_he | (d > reach_b)) ? (2u | 4u | 16u) : 0u;               // hydrogens of end, target = bgn
                dead |= (mb & M_HAS_SB) ? 0u : 16u;
                dead |= (me & M_HAS_SB) ? 0u : 32u;
                need_ = ((need_ & ~dead) == 0u) ? 0u : need_;
            }
            // interactions.py:748-757: end among the bonded neighbours of bgn (only pairs within the longest bond can be)
            // (one of the four equal <=> the smallest of the four XORs is zero; -1 / -2 never equal a local id)
            const bool near_bond = d <= longest_bond;
            const unsigned nx_ = min(min((unsigned)(nbr.y ^ e), (unsigned)(nbr.z ^ e)), (unsigned)(nbr.w ^ e));
            bool cov = near_bond & (nx_ == 0u);
            if (near_bond & !cov & (nbr.w == -2)) {                                   // more than three neighbours in other residues: the whole list
                for (int k = sd.bond_off[b_], k1 = sd.bond_off[b_ + 1]; k < k1; ++k)
                    if (bond_idx[k] == e) { cov = true; break; }
            }
            s = cov ? ARP_S_COVALENT : s;
            s |= s_metal;
            // interactions.py:786: feature flags only for pairs that do not clash (covalent ones do get them) within 4.5 A
            const bool feat = !(s & ARP_S_CLASH) & (d <= (float)4.5);
            s |= feat ? f : 0u;
#ifdef EXP_NOTASKS      // (timing experiments only: no hydrogen geometry)
            const unsigned need = 0u;
#else
            const unsigned need = feat ? need_ : 0u;
#endif
            // interactions.py:889-895 (halogen bond: float32 angle at the donor), rare
            if (feat & in_vc & ((XY & 4u) != 0u)) {
                if (X & 4u) { if (xbond(sd.sb[b_], xb, xe, err)) s |= ARP_S_XBOND; }
                else if (xbond(sd.sb[e], xe, xb, err)) s |= ARP_S_XBOND;
            }
            if (need) {       // (stage B stores the finished mask over the one stored below)
                queued = true;
                task = make_uint4((unsigned)p, (unsigned)pr_cur.x, (unsigned)pr_cur.y, s | (need << 16));
            }
            }
        // the records of the batch: five vector-memory operations behind the gather (a queued pair's mask is stored again by
        // stage B, later in program order)
        if (live) put_five<STREAM>(out_i + p, GID ? gid[b_] : b_, out_j + p, GID ? gid[e] : e, out_d + p, d, out_ct + p, (unsigned)ct, out_s + p, s);
        // stage B bookkeeping (whole wave)
        const unsigned long long mq = __ballot(queued);
        if (mq) {
            if (queued) tq[w][tn + __popcll(mq & ((1ull << lane) - 1ull))] = task;
            tn += __popcll(mq);
            if (tn >= 64) {
                tn -= 64;
                run_tasks(tn, 64);
            }
        }
        pr_cur = pr;
#ifdef ARP_SIFT_TRACE
        const unsigned long long tw0 = __builtin_amdgcn_s_memrealtime();
#endif
        if (SIFT_COUNTED_WAIT) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");     // the next batch's records have landed (this one's stores may be on their way)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef ARP_SIFT_TRACE
        tr[6] += __builtin_amdgcn_s_memrealtime() - tw0;
        tr[7] += 1ull | ((unsigned long long)__popcll(__ballot(live)) << 32);
#endif
    }
    SIFT_T(4);
    if (tn > 0) run_tasks(0, tn);
    SIFT_T(5);
#ifdef ARP_SIFT_TRACE
    if (g_search_trace && lane == 0) {
        unsigned long long* t = g_search_trace + ((size_t)vblock * 4 + w) * 8;
        for (int k = 0; k < 8; ++k) t[k] = tr[k];
    }
#endif
}

// Per-atom accumulators of the contact loop (interactions.py:821-852, 923-934; utils.py:182-221) from the
// resident contact list: OR of the pair SIFt into sift / sift_inter_only (type == 'INTER') / sift_intra_only
// ('INTRA' in type) / sift_water_only ('WATER' in type), and the hbond / polar counters with the
// INTRA -> INTER -> WATER elif chain.  The feature SIFt (actual_fsift*) is bits 5..14 of the same masks.
// acc_sift: u32[2n] = {all | inter << 16, intra | water << 16}; acc_cnt: i32[8n] =
// {hbonds, hbonds_intra, hbonds_inter, hbonds_water, polars, polars_intra, polars_inter, polars_water}.
__global__ __launch_bounds__(256) void k_accumulate(long long np, const int* __restrict__ ci, const int* __restrict__ cj,
                                                    const uint16_t* __restrict__ cs, const uint8_t* __restrict__ cct,
                                                    unsigned int* __restrict__ acc_sift, int* __restrict__ acc_cnt) {
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < np; p += (long long)gridDim.x * blockDim.x) {
        const unsigned int s = cs[p];
        const int ct = cct[p];
        const bool inter = ct == ARP_CT_INTER;
        const bool intra = ct == ARP_CT_INTRA_NON_SELECTION || ct == ARP_CT_INTRA_SELECTION;
        const bool water = ct == ARP_CT_SELECTION_WATER || ct == ARP_CT_NON_SELECTION_WATER || ct == ARP_CT_WATER_WATER;
        const unsigned int w0 = s | (inter ? s << 16 : 0u);
        const unsigned int w1 = (intra ? s : 0u) | (water ? s << 16 : 0u);
        const int slot = intra ? 1 : (inter ? 2 : (water ? 3 : -1));   // 'INTRA' / elif 'INTER' / elif 'WATER'
        const int atoms[2] = {ci[p], cj[p]};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int a = atoms[k];
            atomicOr(&acc_sift[2 * (size_t)a], w0);
            if (w1) atomicOr(&acc_sift[2 * (size_t)a + 1], w1);
            if (s & ARP_S_HBOND) {
                atomicAdd(&acc_cnt[8 * (size_t)a], 1);
                if (slot > 0) atomicAdd(&acc_cnt[8 * (size_t)a + slot], 1);
            }
            if (s & ARP_S_POLAR) {
                atomicAdd(&acc_cnt[8 * (size_t)a + 4], 1);
                if (slot > 0) atomicAdd(&acc_cnt[8 * (size_t)a + 4 + slot], 1);
            }
        }
    }
}

// utils.update_atom_integer_sift (U:224-242) under the canonical pair order (contacts sorted by (i, j), i < j): for an
// atom a the last pair of a class is the one with the largest j among its pairs as bgn, or — if it is never bgn — the
// largest i among its pairs as end.  rank = (a is bgn) << 32 | partner + 1.  Pass 1 takes the maximum rank per
// (atom, class); pass 2 ORs every other pair's SIFt into `before` and stores the last pair's SIFt; pass 3 writes
// before + last per bit.  Classes: 0 every pair, 1 type == 'INTER', 2 'INTRA' in type, 3 'WATER' in type.
__device__ __forceinline__ unsigned int isift_classes(int ct) {
    const bool inter = ct == ARP_CT_INTER;
    const bool intra = ct == ARP_CT_INTRA_NON_SELECTION || ct == ARP_CT_INTRA_SELECTION;
    const bool water = ct == ARP_CT_SELECTION_WATER || ct == ARP_CT_NON_SELECTION_WATER || ct == ARP_CT_WATER_WATER;
    return 1u | (inter ? 2u : 0u) | (intra ? 4u : 0u) | (water ? 8u : 0u);
}
__global__ __launch_bounds__(256) void k_isift_last(long long np, const int* __restrict__ ci, const int* __restrict__ cj,
                                                    const uint8_t* __restrict__ cct, u64* __restrict__ last_rank) {
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < np; p += (long long)gridDim.x * blockDim.x) {
        const unsigned int cls = isift_classes(cct[p]);
        const int i = ci[p], j = cj[p];
        const u64 rank_i = (1ull << 32) | (u64)(unsigned int)(j + 1), rank_j = (u64)(unsigned int)(i + 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (!((cls >> s) & 1u)) continue;
            atomicMax(&last_rank[4 * (size_t)i + s], rank_i);
            atomicMax(&last_rank[4 * (size_t)j + s], rank_j);
        }
    }
}
__global__ __launch_bounds__(256) void k_isift_fill(long long np, const int* __restrict__ ci, const int* __restrict__ cj,
                                                    const uint16_t* __restrict__ cs, const uint8_t* __restrict__ cct,
                                                    const u64* __restrict__ last_rank, unsigned int* __restrict__ before,
                                                    unsigned int* __restrict__ last_sift) {
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < np; p += (long long)gridDim.x * blockDim.x) {
        const unsigned int cls = isift_classes(cct[p]), sft = cs[p];
        const int i = ci[p], j = cj[p];
        const u64 rank_i = (1ull << 32) | (u64)(unsigned int)(j + 1), rank_j = (u64)(unsigned int)(i + 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (!((cls >> s) & 1u)) continue;
            const size_t ki = 4 * (size_t)i + s, kj = 4 * (size_t)j + s;
            if (last_rank[ki] == rank_i) last_sift[ki] = sft; else atomicOr(&before[ki], sft);
            if (last_rank[kj] == rank_j) last_sift[kj] = sft; else atomicOr(&before[kj], sft);
        }
    }
}
__global__ __launch_bounds__(256) void k_isift_compose(long long n4, const unsigned int* __restrict__ before,
                                                       const unsigned int* __restrict__ last_sift, uint8_t* __restrict__ out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n4) return;
    const unsigned int b = before[t], l = last_sift[t];
#pragma unroll
    for (int k = 0; k < 15; ++k) out[15 * t + k] = (uint8_t)(((b >> k) & 1u) + ((l >> k) & 1u));
}

// End of a pass: the counter block goes to the pinned host copy (a kernel store to mapped host memory costs one
// launch; a D2H hipMemcpyAsync of 3 KB costs an SDMA round trip) and, when asked, returns to zero for the next pass.
__global__ __launch_bounds__(256) void k_publish_counters(u64* __restrict__ ctr, u64* __restrict__ host, int n, int zero, u64 seq) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        host[i] = ctr[ctr_dev(i)];
        if (zero) ctr[ctr_dev(i)] = 0;
    }
    // host[n] = sequence number of this pass, stored after everything else is visible to the host: the host may
    // poll it instead of blocking in hipStreamSynchronize (this kernel is the last one of the pass)
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(host + n, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}


// _make_selection for SMALL selections (a ligand, a handful of residues: the reference's `-s /A/508/` use): selection_plus
// = selection + every atom within `radius` of a selected atom (I:1420-1424; all atoms, hydrogens included) computed
// directly — every atom against the list of selected atoms, staged through LDS — instead of a grid search over the
// whole structure: no grid to build on the critical path and N x S exact float64 tests (Bio.PDB.kdtrees' inclusive
// d^2 <= r^2) where S is a few dozen.  stats[0] += tests, stats[1] += atoms added.
#define SMALL_SEL_MAX 1024
__global__ __launch_bounds__(256) void k_expand_small(int n, const float4* __restrict__ xyz, const int* __restrict__ sel_list,
                                                      int nsel, const uint8_t* __restrict__ sel, double r2,
                                                      uint8_t* __restrict__ plus, u64* __restrict__ stats) {
    __shared__ float4 s_sel[256];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    const float4 v = live ? xyz[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const num::d3 x = {(double)v.x, (double)v.y, (double)v.z};
    bool in = live && sel[i] != 0;
    const bool was = in;
    unsigned tests = 0;
    for (int s0 = 0; s0 < nsel; s0 += 256) {
        __syncthreads();
        if (s0 + (int)threadIdx.x < nsel) s_sel[threadIdx.x] = xyz[sel_list[s0 + threadIdx.x]];
        __syncthreads();
        const int m = min(256, nsel - s0);
        if (live && !in) {
            for (int k = 0; k < m; ++k) {
                const float4 q = s_sel[k];
                ++tests;
                if (num::dist2_kd(x, num::d3{(double)q.x, (double)q.y, (double)q.z}) <= r2) { in = true; break; }
            }
        }
    }
    if (live) plus[i] = in ? 1 : 0;
    // statistics: one pair of atomics per block
    __shared__ unsigned s_t[4], s_a[4];
    unsigned t = tests, a = (in && !was) ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) { t += __shfl_xor(t, o); a += __shfl_xor(a, o); }
    if ((threadIdx.x & 63) == 0) { s_t[threadIdx.x >> 6] = t; s_a[threadIdx.x >> 6] = a; }
    __syncthreads();
    if (threadIdx.x == 0 && stats) {
        atomicAdd(stats + (blockIdx.x & (STAT_SLOTS - 1)) * CTR_LINE, (u64)(s_t[0] + s_t[1] + s_t[2] + s_t[3]));       // (mark cand, mark acc: neighbours in the slot's line)
        atomicAdd(stats + (blockIdx.x & (STAT_SLOTS - 1)) * CTR_LINE + 1, (u64)(s_a[0] + s_a[1] + s_a[2] + s_a[3]));
    }
}

// residue / ring / amide membership of _make_selection (interactions.py:1413-1437)
__global__ __launch_bounds__(256) void k_res_mark(int n, const int* __restrict__ res_id, const uint8_t* __restrict__ sel,
                                                  const uint8_t* __restrict__ plus, uint8_t* __restrict__ res_sel,
                                                  uint8_t* __restrict__ res_plus) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (sel[i]) res_sel[res_id[i]] = 1;
        if (plus[i]) res_plus[res_id[i]] = 1;
    }
}
// rings [0, nring) and amides [nring, nring + namide) in one launch
__global__ __launch_bounds__(256) void k_group_mask(int nring, int namide, const int* __restrict__ ring_res,
                                                    const int* __restrict__ amide_res, const uint8_t* __restrict__ res_sel,
                                                    const uint8_t* __restrict__ res_plus, uint8_t* __restrict__ ring_sel,
                                                    uint8_t* __restrict__ ring_plus, uint8_t* __restrict__ amide_sel,
                                                    uint8_t* __restrict__ amide_plus) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nring + namide; i += gridDim.x * blockDim.x) {
        const bool ring = i < nring;
        const int k = ring ? i : i - nring;
        const int r = ring ? ring_res[k] : amide_res[k];
        const uint8_t s_ = (r >= 0 && res_sel[r]) ? 1 : 0;    // a ring whose residue is None never qualifies
        const uint8_t p_ = (r >= 0 && res_plus[r]) ? 1 : 0;
        if (ring) { ring_sel[k] = s_; ring_plus[k] = p_; }
        else { amide_sel[k] = s_; amide_plus[k] = p_; }
    }
}
