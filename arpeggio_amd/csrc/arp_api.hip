// arp_api.hip — C ABI (include/arpeggio_hip.h) over the HIP kernels.  gfx950 only.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC
//        arp_api.hip -o libarpeggio_hip.so
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <mutex>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/arpeggio_hip.h"
#include "arp_grid.h"
#include "arp_numerics.h"
#include "arp_pairs.h"
#include "arp_planes.h"
#include "arp_prepare.h"
#include "arp_json.h"
#include "arp_shard.h"
#include "arp_cif.h"
#include "arp_comm.h"
#include "arp_sort.h"

namespace {

std::string g_create_error;   // arp_create failures only (no context to hold the message); written under g_create_mutex
std::mutex g_create_mutex;
void set_create_error(const std::string& m) {
    std::lock_guard<std::mutex> lk(g_create_mutex);
    g_create_error = m;
}

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    bool borrowed = false;   // p points into memory owned by someone else (the device copy of an uploaded blob)
    // *fresh (optional) is set when new memory was allocated (contents undefined)
    hipError_t reserve(size_t n, bool* fresh = nullptr) {
        if (fresh) *fresh = false;
        if (n <= cap && p) return hipSuccess;
        release();
        size_t c = n + n / 4 + 64;
        hipError_t e = hipMalloc((void**)&p, c * sizeof(T));
        if (e == hipSuccess) { cap = c; if (fresh) *fresh = true; }
        else p = nullptr;
        return e;
    }
    void borrow(void* ptr, size_t n) {
        release();
        p = (T*)ptr;
        cap = n;
        borrowed = true;
    }
    void release() {
        if (p && !borrowed) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        borrowed = false;
    }
};

struct Grid {
    GridDesc d{};
    DevBuf<int> cell_of, cnt, start, perm, sums;
    // atom grids: {cell, rank in cell} per atom and a second histogram buffer (double-buffered: a build counts in
    // hist[cur] and clears hist[1 - cur], the one the previous build used)
    DevBuf<int2> cell_rank;
    DevBuf<int> cnt2;
    DevBuf<unsigned long long> chain;   // k_scan_tiles_chained: {launch number, tile total} per tile
    unsigned int chain_epoch = 0;
    int cur = 0;
    size_t used[2] = {0, 0};   // counters of hist[k] that may be non-zero
    int n_points = 0;   // points offered
    int n_binned = 0;   // points that passed the filter
    bool valid = false;
    double radius = 0;
    void release() { cell_of.release(); cnt.release(); start.release(); perm.release(); sums.release(); cell_rank.release();
                     cnt2.release(); chain.release(); valid = false; }
};

struct Bag {  // outputs of one ring/amide kernel, resident in HBM until fetched
    // the arrays are sub-ranges of ONE allocation (slab), so that a small bag reaches the host with one copy
    DevBuf<uint8_t> slab;
    DevBuf<int> a, b;
    DevBuf<double> d0, d1, d2, d3;
    DevBuf<float> f0, f1, f2;
    DevBuf<uint8_t> u0, u1, u2;
    size_t cap = 0;
    size_t slab_bytes = 0;
    int64_t count = 0;
    bool valid = false;
    uint64_t version = 0;        // bumped whenever a launch refills the bag
    uint64_t staged_version = 0; // version whose records are in the context's page-locked staging area, at stage_off[]
    uint32_t stage_off[12] = {0};
    void release() { a.release(); b.release(); d0.release(); d1.release(); d2.release(); d3.release(); f0.release();
                     f1.release(); f2.release(); u0.release(); u1.release(); u2.release(); slab.release(); cap = 0; slab_bytes = 0;
                     valid = false; staged_version = 0; }
};

// the used prefixes of every array of every bag, gathered into one device buffer for one copy to the host
// (perm: the segment's elements — es bytes each — leave in the order perm[0], perm[1], ...: the canonical order of a small bag)
struct PackSeg { const uint8_t* src; uint32_t dst, bytes; const uint32_t* perm; uint32_t es; };
struct PackTable { PackSeg s[48]; int n; };
// n 16-byte quads from the device to a page-locked host buffer (small results: see arp_fetch_packed)
__global__ __launch_bounds__(256) void k_copy_quads(const int4* __restrict__ src, int4* __restrict__ dst, unsigned n) {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_pack_segments(PackTable t, uint8_t* __restrict__ out) {
    const PackSeg g = t.s[blockIdx.y];        // segment blockIdx.y, spread over the gridDim.x blocks of its row
    if (g.perm) {
        const uint32_t n = g.bytes / g.es;
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
            const uint32_t q = g.perm[i];
            if (g.es == 4) reinterpret_cast<uint32_t*>(out + g.dst)[i] = reinterpret_cast<const uint32_t*>(g.src)[q];
            else if (g.es == 8) reinterpret_cast<unsigned long long*>(out + g.dst)[i] = reinterpret_cast<const unsigned long long*>(g.src)[q];
            else out[g.dst + i] = g.src[q];
        }
        return;
    }
    const uint32_t words = g.bytes >> 2;
    const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(g.src);      // (arrays of a slab start on 256-byte boundaries,
    uint32_t* __restrict__ dst = reinterpret_cast<uint32_t*>(out + g.dst);            //  destinations on 16-byte ones)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x < (g.bytes & 3u)) out[g.dst + (words << 2) + threadIdx.x] = g.src[(words << 2) + threadIdx.x];
}
// Canonical order of the four ring / amide bags (the order the reference creates their records in: by the ids of the two
// partners, I:947-1382): the kernels emit them through one atomic counter per bag, i.e. in any order.  A bag of up to
// BAG_SORT_MAX records is sorted by ONE block (bitonic network in LDS on {first id, second id, record index}); larger bags keep
// the device's order (the caller sorts them: config 5 has 10^5 records per bag, a protein a few hundred).
#define BAG_SORT_MAX 8192      // (96 KB of LDS: key + index)
static_assert(BAG_SORT_MAX == ARP_BAG_SORT_MAX, "include/arpeggio_hip.h");
struct BagOrderArgs { const int* first[4]; const int* second[4]; int n[4]; uint32_t* perm[4]; };
__global__ __launch_bounds__(1024) void k_bag_order(BagOrderArgs A) {
    __shared__ unsigned long long s_key[BAG_SORT_MAX];
    __shared__ uint32_t s_idx[BAG_SORT_MAX];
    const int b = blockIdx.x, n = A.n[b];
    if (n <= 0 || n > BAG_SORT_MAX) return;
    int m = 2;
    while (m < n) m <<= 1;
    const int tid = (int)threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < m; i += 1024) {
        s_key[i] = i < n ? (((unsigned long long)(uint32_t)A.first[b][i] << 32) | (unsigned long long)(uint32_t)A.second[b][i]) : ~0ull;
        s_idx[i] = (uint32_t)i;
    }
    __syncthreads();
    // Bitonic network over m slots.  Every wave owns a contiguous chunk of C slots: the steps whose partners lie inside a chunk
    // (distance j < C: 81 of the 91 steps of 8192 slots) need no block barrier — a wave's LDS operations are performed in
    // order — and a step walks the m / 2 PAIRS, not the m slots.  (One barrier per step over all slots: 176 us for 8192.)
    const int C = max(m / 16, min(m, 128));
    const int nwave_active = m / C;
    auto step = [&](int pr, int j, int k) {      // compare-exchange of pair pr at distance j inside the merge of size k
        const int i = ((pr & ~(j - 1)) << 1) | (pr & (j - 1)), l = i + j;
        const unsigned long long ka = s_key[i], kb = s_key[l];
        const uint32_t ia = s_idx[i], ib = s_idx[l];
        const bool up = (i & k) == 0;
        const bool gt = ka > kb || (ka == kb && ia > ib);      // (ties by record index: the order is a function of the records)
        if (gt == up) { s_key[i] = kb; s_key[l] = ka; s_idx[i] = ib; s_idx[l] = ia; }
    };
    for (int k = 2; k <= m; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= C) {
                __syncthreads();
                for (int pr = tid; pr < (m >> 1); pr += 1024) step(pr, j, k);
                __syncthreads();
            } else {
                if (w < nwave_active)
                    for (int q = lane; q < (C >> 1); q += 64) step(w * (C >> 1) + q, j, k);
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
    __syncthreads();
    for (int i = tid; i < n; i += 1024) A.perm[b][i] = s_idx[i];
}
// blocks per segment: one per 16 KiB of the largest segment, at most 64
inline dim3 pack_grid(const PackTable& t) {
    uint32_t mx = 0;
    for (int k = 0; k < t.n; ++k) mx = std::max(mx, t.s[k].bytes);
    return dim3(std::min<uint32_t>(std::max<uint32_t>((mx + 16383u) >> 14, 1u), 64u), (unsigned)t.n);
}

// arp_set_batch: item i belongs to the structure s with off[s] <= i < off[s + 1] (empty structures have none)
__global__ __launch_bounds__(256) void k_fill_sid(const long long* __restrict__ off, int nstruct, int* __restrict__ sid, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int lo = 0, hi = nstruct;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (off[mid] <= i) lo = mid; else hi = mid;
        }
        sid[i] = lo;
    }
}

enum Slot { SLOT_BIN = 0, SLOT_SCAN = 1, SLOT_SCATTER = 2, SLOT_UNUSED = 3, SLOT_SEARCH = 4, SLOT_SIFT = 5, SLOT_MARK = 6, SLOT_PLANES = 7, NSLOT = 8 };

struct EventPair { int slot; hipEvent_t a, b; };

}  // namespace

struct arp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;       // the stream every pass is enqueued on (own_stream unless arp_use_stream)
    hipStream_t own_stream = nullptr;
    bool external_stream = false;
    std::string err;
    int num_cu = 256;
    int search_resident = 768;          // blocks of k_search<MODE_CONTACTS> the chip holds at once (occupancy x CUs)
    int sift_per_cu = 4;                // blocks of the per-pair kernel a CU holds at once
    bool xcd_round_robin = true;   // consecutive blocks of a launch land on XCDs (x0 + b) % 8 (arp_create's probe)
    uint64_t seg_last[PAIR_SEGS] = {0, 0, 0, 0, 0, 0, 0, 0};   // pairs per segment of the last contact pass (the next one's sift blocks are shared out by it)

    // ---- sizes
    int64_t n = 0, nres = 0, nring = 0, namide = 0;
    // ---- host mirrors needed for later set_* calls / grid boxes
    std::vector<float> h_xyz;
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    double ring_lo[3] = {0, 0, 0}, ring_hi[3] = {0, 0, 0};
    double am_lo[3] = {0, 0, 0}, am_hi[3] = {0, 0, 0};
    // ---- raw inputs
    DevBuf<float4> xyz;          // w unused
    DevBuf<double2> rad;         // {vdw, cov}
    DevBuf<uint16_t> tmask, flags;
    DevBuf<int> res_id, res_prev, res_next;
    DevBuf<uint8_t> res_flags;
    DevBuf<int> bond_off, bond_idx, h_off;
    DevBuf<double> h_xyz_d;
    DevBuf<float4> sb;           // single-bond neighbour xyz, w = 1 if present
    DevBuf<int> gid;
    DevBuf<uint8_t> home, sel, plus, res_sel, res_plus;
    bool has_res = false, has_gid = false, has_home = false;
    int64_t max_res_id = -1, max_ring_res = -1, max_amide_res = -1;   // host-side range checks of the uploaded indices
    bool sel_made = false;
    bool sel_uploaded = false;   // arp_set_selection / arp_set_selection_state since the last arp_set_atoms
    bool sel_prefilled = false;  // sel already holds the default selection (all ones) for the resident structure: written while arp_set_blob waited for the validation
    size_t sp_cnt_zeroed = 0;    // leading ints of sp_cnt cleared the same way (ensure_static's fill of a fresh structure)
    bool sel_all = false;        // the uploaded selection covers every atom: selection_plus = selection, no expansion search
    int64_t nsel = -1;            // selected atoms of the uploaded mask; their indices are in sel_list when nsel <= SMALL_SEL_MAX
    DevBuf<int> sel_list;
    bool whole_structure = false; // caller's assertion (arp_set_whole_structure): the selection is the whole GLOBAL structure
    DevBuf<double> ring_c, ring_n;
    DevBuf<int> ring_res;
    DevBuf<uint8_t> ring_sel, ring_plus;
    DevBuf<float> am_c, am_n;
    DevBuf<int> am_res;
    DevBuf<uint8_t> am_sel, am_plus;
    DevBuf<uint8_t> ring_home, am_home;
    DevBuf<int> ring_gid, am_gid;
    bool has_group_owner = false;
    // ---- derived
    DevBuf<float4> s_xyzm;
    DevBuf<int4> s_aux;
    DevBuf<int4> s_qa;            // second quad of the sift records, cell-sorted (the first one is s_xyzm)
    DevBuf<int> st_h, sp_h, s_h;  // index of every atom's first hydrogen: static, in the spatial order, cell-sorted
    DevBuf<int> tmp_i32;          // scratch for index uploads
    DevBuf<int4> st_qa;           // selection-independent record columns, composed once per structure (k_prepare_static)
    DevBuf<float> longest_bond;   // k_longest_bond, once per uploaded structure (ensure_static)
    DevBuf<uint16_t> rad_idx;     // per atom: index of its {vdw, cov} pair in rad_tab (RAD_NONE: not in the table)
    DevBuf<double2> rad_tab;      // RAD_TABLE distinct radius pairs of the structure
    DevBuf<int4> st_aux;
    DevBuf<float4> st_xyzm;
    DevBuf<float4> sp_xyzm;       // the same columns in the spatial order of the structure (what the per-pass grid builds read)
    DevBuf<int4> sp_aux, sp_qa;
    DevBuf<int> sp_cnt;
    DevBuf<int> sp_sums;         // tile totals of the static order's scan (grids beyond 32768 cells)
    DevBuf<int2> sp_cr;
    DevBuf<int> sp_cell;          // cell of every row of the spatial order
    GridDesc sp_grid{};           // the grid the spatial order was made for
    double sp_radius = 0;         // ... and its cell edge (0: no order yet)
    DevBuf<unsigned long long> compact_chain;   // k_compact_atoms: one word per block
    DevBuf<int> s_cell;                         // cell of every row of the contact grid (k_search: blocks split by atoms)
    bool s_cell_valid = false;                  // ... written by the build of the grid that is in place
    DevBuf<int> sb_tile;                        // k_search's runs of tiles with equal numbers of atoms (k_balance_blocks): a hint from the pass before
    bool sb_valid = false;
    uint64_t sb_static_epoch = 0, sb_sel_epoch = 0;
    double sb_radius = 0.0;
    int sb_blocks = 0, sb_tx = 0;
    bool sb_whole = false;
    bool sb_by_atoms = false, sb_seen_by_atoms = false;      // what the entries of sb_tile are: tiles, or atom positions (k_balance_atoms)
    DevBuf<int> sb_cw, sb_cwp, sb_sums;                      // weight per cell, its running sum, tile totals of that scan
    // (the grid key of the last pass that ran WITHOUT a hint: the hint is worked out by the second such pass over one grid — a
    // structure that is evaluated once, the usual case, never pays for it)
    bool sb_seen = false;
    uint64_t sb_seen_static_epoch = 0, sb_seen_sel_epoch = 0;
    double sb_seen_radius = 0.0;
    int sb_seen_blocks = 0, sb_seen_tx = 0;
    unsigned int compact_epoch = 0;
    bool static_dirty = true;
    // The contact grid of a WHOLE-STRUCTURE pass (every atom selected: selection_plus = all atoms, I:1395 / 1407) depends on the
    // structure and the cell edge only — it is the structure's own neighbour grid, the counterpart of the KD-tree the reference
    // builds over `entity` (I:1394).  Such a pass keeps it: the next one with the same structure, cell edge and (whole)
    // selection launches no k_compact_atoms.  Any other selection compacts per pass, as the reference rebuilds
    // NeighborSearch(selection_plus) (I:1442).  arp_set_grid_reuse(ctx, 0) switches the reuse off (bench.py reports both).
    bool grid_reuse = true;
    bool sort_after_pass = false;     // arp_set_sort_after_pass
    bool cg_valid = false, cg_fuse = false, cg_init_plus = false, cg_all_res = false, cg_pending = false, cg_reused = false;
    double cg_radius = 0.0;
    uint64_t static_epoch = 0, sel_epoch = 0, cg_static_epoch = 0, cg_sel_epoch = 0;
    int64_t cg_binned = 0;
    double host_enqueue_us = 0, host_wait_us = 0;   // arp_run_launch: time spent enqueueing / waiting (arp_get_host_times)
    int64_t host_passes = 0;
    Grid atom_grid, all_grid, ring_grid, amide_grid;   // contact grid (selection_plus, no H) / every atom at 6 A
    DevBuf<float4> a_xyzm;        // cell-sorted records of all_grid
    DevBuf<int4> a_aux;
    bool all_grid_current = false;  // all_grid matches the current inputs and selection
    hipStream_t stream2 = nullptr;  // ring / amide kernels run here, concurrently with the contact pipeline
    hipEvent_t ev_sel = nullptr, ev_planes = nullptr, ev_lists = nullptr;
    // arp_set_blob: the centre grids and candidate lists of the new structure are made on the second stream, behind ev_upload (the
    // validation kernel on the main one) and in front of ev_uplists; the main stream joins them before anything reads or replaces
    // what they read or write (join_upload_lists)
    hipEvent_t ev_upload = nullptr, ev_uplists = nullptr;
    bool uplists_pending = false;
    DevBuf<int> upload_bad;           // number of the last upload whose device-side check failed (validate_resident_blob)
    bool upload_bad_cleared = false;
    int ahead_seq = 0;                // != 0: ensure_static is enqueued ahead of the verdict of upload number ahead_seq
    double last_cutoff = 0.0;         // cell edge of the context's last pass
    bool uplists_defer = false;       // inside enqueue_contacts: see join_upload_lists
    DevBuf<uint8_t> tmp_u8;
    // ---- pair list and outputs of the atom-contact pass
    DevBuf<int2> pairs;
    DevBuf<int> out_i, out_j;
    DevBuf<float> out_d;
    DevBuf<uint16_t> out_s;
    DevBuf<uint8_t> out_ct;
    int64_t n_contacts = 0;
    // ---- canonical (i, j) order of the atom-atom bag, made on the device (arp_sort.h; arp_atom_contacts_sort)
    DevBuf<unsigned long long> sort_key[2];
    DevBuf<unsigned long long> sort_val[2];
    DevBuf<int> sort_table;
    DevBuf<long long> sort_total;
    DevBuf<uint8_t> sorted_slab;        // the five sorted columns (+ the packed ring / amide bags of a packed fetch) in one piece
    size_t srt_off[5] = {0, 0, 0, 0, 0};  // byte offsets of i, j, distance, SIFt, contact type in sorted_slab
    size_t srt_bytes = 0;               // bytes of the five columns
    bool contacts_sorted = false;       // sorted_slab holds the records of the last launch in (i, j) order
    bool packed_csr = false;            // arp_set_packed_layout: the sorted bag's first column is N + 1 row offsets instead of k bgn ids
    bool sorted_is_csr = false;         // ... and that is what sorted_slab holds now
    int64_t gid_max = -1;               // largest global atom id of a shard (-1: not known, 31-bit keys)
    bool pass_pending = false;      // arp_run_enqueue without its arp_run_wait yet
    double pending_cutoff = 5.0, pending_comp = 0.1, pending_expand = 6.0;
    int pending_seq_adj = 0;
    int64_t contacts_expected = 0;   // contacts the previous pass over this structure found (0: none yet): sizes the sift launch
    bool contacts_valid = false;
    u64* d_ctr = nullptr;        // C_COUNT device counters
    u64 h_ctr[C_COUNT] = {0};
    int64_t stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t contact_cells = 0;
    bool ctr_clean = false;
    u64 publish_seq = 0;         // number of k_publish_counters launches; the kernel stores it in h_ctr_pinned[C_COUNT]
    bool ctr_zero_ok = false;    // the last arp_run_launch left the whole block zeroed (k_publish_counters) and nothing touched it since
    u64* h_ctr_pinned = nullptr;   // pinned mirror of the counter block + completion word
    PublishArgs pub{nullptr, nullptr, 0, 0};   // in-kernel end-of-pass publication (arp_run_launch sets it for one pass)
    // residue sets of arp_run_launch, tagged with the pass number (never cleared between passes; wrap -> one memset)
    // static candidate lists of the ring / amide loops (k_plane_lists), rebuilt when the structure changes
    DevBuf<int2> plist[4];
    DevBuf<u64> plist_count;       // [4]
    bool plist_count_cleared = false;   // k_point_grids has just zeroed them (same stream, same pass)
    bool lists_dirty = true;
    bool lists_from_upload = false;   // the lists in place were made with the upload of the resident structure (validate_resident_blob): composing its static columns does not stale them
    long long plist_known[4] = {-1, -1, -1, -1};   // entries the lists held at the end of the last pass (-1: not known yet)
    DevBuf<uint8_t> blob_dev;      // device copy of the last arp_set_blob upload (the input arrays are views into it)
    int64_t blob_nbond = 0, blob_nh = 0, blob_nrad = 0;
    bool validate_on_device = false;   // blob uploads: k_prepare_static checks what the classic setters check on the host
    DevBuf<int> blob_sb_nbr;
    uint64_t blob_bytes = 0;       // size of the resident blob (0: the structure came through the classic setters)
    // ---- sharded structures assembled on the device (arp_shard_*)
    DevBuf<uint8_t> rec_home, rec_face[2];
    arp_rec_header rec_home_hdr;
    bool has_rec_home = false;
    DevBuf<int> sh_scan;           // flags / counts and their exclusive scans
    DevBuf<int2> sh_src;
    DevBuf<int8_t> origin, ring_origin, am_origin;
    DevBuf<uint8_t> sh_sel;
    bool shard_resident = false;
    DevBuf<uint8_t> res_tag;       // [0, nres) = selection residues, [nres, 2 nres) = selection_plus residues
    int res_tag_value = 0;
    bool fuse_sets = false;        // the contact-grid build of the current pass also makes the residue / ring / amide sets
    bool init_plus_in_bin = false; // ... and writes selection_plus = selection (whole-structure selection)
    // ---- device-resident result bags of the ring / amide kernels
    Bag bag_ap, bag_pp, bag_gg, bag_gp;
    DevBuf<uint32_t> bag_perm_big[4];   // ... of a bag beyond BAG_SORT_MAX records (bag_order_large)
    DevBuf<unsigned long long> bagsort_key[2], bagsort_val[2];
    DevBuf<int> bagsort_table, bagsort_i, bagsort_j;
    DevBuf<long long> bagsort_total;
    DevBuf<uint16_t> bagsort_s;
    DevBuf<uint8_t> bagsort_ct;
    DevBuf<uint32_t> bag_perm;     // canonical order of the small ring / amide bags (k_bag_order), BAG_SORT_MAX indices per bag
    DevBuf<uint8_t> bag_pack;      // staging of small bags: device side ...
    uint8_t* bag_stage = nullptr;  // ... and its page-locked host copy
    size_t bag_stage_cap = 0;
    // ---- exchange between shards (arp_comm_*: RCCL on the context's stream)
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    DevBuf<uint8_t> comm_recv[2];          // what the left / right neighbour sent last
    DevBuf<unsigned long long> comm_words; // sizes on their way: [0, 1] out (left, right), [2, 3] in
    DevBuf<int> xl_send[2], xl_recv[2];    // per-pass selection exchange: local atom indices sent to / received from each neighbour
    DevBuf<uint8_t> xl_send_buf[2], xl_recv_buf[2];
    int64_t xl_nsend[2] = {0, 0}, xl_nrecv[2] = {0, 0};
    // ---- several structures in one pass (arp_set_batch): structure s = atoms [atom_off[s], atom_off[s + 1]), rings, amides likewise
    int64_t batch_n = 0;
    std::vector<int64_t> batch_atom_off, batch_ring_off, batch_amide_off;
    std::vector<double> batch_box;       // 6 per structure: lo xyz, hi xyz
    DevBuf<int> sid_atom, sid_ring, sid_amide;
    DevBuf<long long> batch_off_dev;   // the three offset tables of arp_set_batch, one after the other
    struct BatchGrid { double radius = 0; bool valid = false; DevBuf<BatchPlace> place; GridDesc d{}; } batch_grid[4];
    // ---- profiling
    bool profiling = false;
    std::vector<EventPair> ev_pool;
    size_t ev_used = 0;
    double k_ms[NSLOT] = {0};
    int64_t k_launches[NSLOT] = {0};
};

namespace {

#define FAIL(ctx, code, msg) do { (ctx)->err = (msg); return (code); } while (0)
#define HIPCHK(ctx, expr)                                                                       \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                     \
            return ARP_E_HIP;                                                                   \
        }                                                                                       \
    } while (0)
#define CHK(expr) do { int rc_ = (expr); if (rc_ != ARP_OK) return rc_; } while (0)

template <class T>
int upload(arp_ctx* c, DevBuf<T>& buf, const T* src, size_t n) {
    HIPCHK(c, buf.reserve(n ? n : 1));
    if (n) HIPCHK(c, hipMemcpyAsync(buf.p, src, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ARP_OK;
}
// several uploads of one setter: enqueue them all, synchronise once (upload_done) before the sources go away
template <class T>
int upload_async(arp_ctx* c, DevBuf<T>& buf, const T* src, size_t n) {
    HIPCHK(c, buf.reserve(n ? n : 1));
    if (n) HIPCHK(c, hipMemcpyAsync(buf.p, src, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return ARP_OK;
}
int upload_done(arp_ctx* c) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ARP_OK;
}
template <class T>
int download_async(arp_ctx* c, T* dst, const T* src, size_t n) {
    if (n && dst) HIPCHK(c, hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyDeviceToHost, c->stream));
    return ARP_OK;
}
template <class T>
int download(arp_ctx* c, T* dst, const T* src, size_t n) {
    if (n && dst) HIPCHK(c, hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ARP_OK;
}

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

inline int nblocks(int64_t work, int threads, int max_blocks = 2048) {
    int64_t b = (work + threads - 1) / threads;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (int)b;
}

struct Prof {  // brackets one launch with events when profiling is on
    arp_ctx* c;
    int slot;
    hipStream_t st;
    EventPair* ep = nullptr;
    Prof(arp_ctx* c_, int slot_, hipStream_t st_ = nullptr) : c(c_), slot(slot_), st(st_ ? st_ : c_->stream) {
        if (!c->profiling) return;
        if (c->ev_pool.capacity() < 256) c->ev_pool.reserve(256);
        if (c->ev_used >= 250) return;
        if (c->ev_used == c->ev_pool.size()) {
            EventPair e{slot, nullptr, nullptr};
            if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return;
            c->ev_pool.push_back(e);
        }
        ep = &c->ev_pool[c->ev_used++];
        ep->slot = slot;
        (void)hipEventRecord(ep->a, st);
    }
    ~Prof() {
        if (ep) (void)hipEventRecord(ep->b, st);
    }
};

void collect_events(arp_ctx* c) {  // call after the pass has ended
    for (size_t k = 0; k < c->ev_used; ++k) {
        float ms = 0;
        (void)hipEventSynchronize(c->ev_pool[k].b);   // the host may have seen the completion word before the last event landed
        if (hipEventElapsedTime(&ms, c->ev_pool[k].a, c->ev_pool[k].b) == hipSuccess) {
            c->k_ms[c->ev_pool[k].slot] += ms;
            c->k_launches[c->ev_pool[k].slot] += 1;
        }
    }
    c->ev_used = 0;
}

int check_launch(arp_ctx* c, const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        c->err = std::string(what) + ": " + hipGetErrorString(e);
        return ARP_E_HIP;
    }
    return ARP_OK;
}

// Choose grid dimensions: cell edge >= radius (slightly larger so that rounding in the
// binning can never separate a pair within the radius by two cells), at most 2^26 cells.
void make_grid_desc(GridDesc& d, const double lo[3], const double hi[3], double radius) {
    double edge = radius * (1.0 + 1e-6);
    if (!(edge > 0)) edge = 1.0;
    for (;;) {
        double nx = std::floor((hi[0] - lo[0]) / edge) + 1, ny = std::floor((hi[1] - lo[1]) / edge) + 1,
               nz = std::floor((hi[2] - lo[2]) / edge) + 1;
        if (nx * ny * nz <= (double)(1 << 26) && nx < 2e9 && ny < 2e9 && nz < 2e9) {
            d.nx = (int)nx; d.ny = (int)ny; d.nz = (int)nz;
            break;
        }
        edge *= 1.26;
    }
    d.ncell = d.nx * d.ny * d.nz;
    d.ox = lo[0]; d.oy = lo[1]; d.oz = lo[2];
    d.inv = 1.0 / edge;
    d.place = nullptr; d.sid_atom = nullptr; d.sid_ring = nullptr; d.sid_amide = nullptr;
}

// Several structures in one grid (arp_set_batch): every structure gets the cells its own box needs at this cell edge and a
// place in a common grid — shelves along x, rows along y, layers along z, one empty cell between neighbours in every
// direction, the whole as near to a cube as the largest structure allows.  Cached per radius.
int batch_grid_desc(arp_ctx* c, GridDesc& d, double radius) {
    arp_ctx::BatchGrid* slot = nullptr;
    for (auto& g : c->batch_grid)
        if (g.valid && g.radius == radius) { d = g.d; return ARP_OK; }
    for (auto& g : c->batch_grid)
        if (!g.valid) { slot = &g; break; }
    if (!slot) {   // every slot holds another radius: start over (grids built with the old tables are rebuilt)
        for (auto& g : c->batch_grid) g.valid = false;
        slot = &c->batch_grid[0];
        c->static_dirty = true; c->lists_from_upload = false; c->lists_dirty = true;
        c->atom_grid.valid = false; c->all_grid_current = false; c->ring_grid.valid = false; c->amide_grid.valid = false;
    }
    const int64_t B = c->batch_n;
    double edge = radius * (1.0 + 1e-6);
    if (!(edge > 0)) edge = 1.0;
    std::vector<BatchPlace> pl((size_t)B);
    int NX = 1, NY = 1, NZ = 1;
    for (;;) {
        double vol = 0;
        int mx = 1, my = 1, mz = 1;
        bool too_big = false;
        for (int64_t s_ = 0; s_ < B; ++s_) {
            const double* lo = &c->batch_box[(size_t)s_ * 6];
            const double* hi = lo + 3;
            const double nx = std::floor((hi[0] - lo[0]) / edge) + 1, ny = std::floor((hi[1] - lo[1]) / edge) + 1, nz = std::floor((hi[2] - lo[2]) / edge) + 1;
            if (!(nx < 4096 && ny < 4096 && nz < 4096)) { too_big = true; break; }
            pl[(size_t)s_] = BatchPlace{lo[0], lo[1], lo[2], 0, 0, 0, (int)nx, (int)ny, (int)nz};
            vol += (nx + 1) * (ny + 1) * (nz + 1);
            mx = std::max(mx, (int)nx); my = std::max(my, (int)ny); mz = std::max(mz, (int)nz);
        }
        if (!too_big) {
            const int side = (int)std::ceil(std::cbrt(vol));
            const int LX = std::max(side, mx), LY = std::max(side, my);
            int x = 0, y = 0, z = 0, row_h = 0, layer_h = 0;
            NX = NY = NZ = 1;
            for (int64_t s_ = 0; s_ < B; ++s_) {
                BatchPlace& b = pl[(size_t)s_];
                if (x > 0 && x + b.nx > LX) { x = 0; y += row_h + 1; row_h = 0; }
                if (y > 0 && y + b.ny > LY) { x = 0; y = 0; z += layer_h + 1; layer_h = 0; row_h = 0; }
                b.cx = x; b.cy = y; b.cz = z;
                NX = std::max(NX, x + b.nx); NY = std::max(NY, y + b.ny); NZ = std::max(NZ, z + b.nz);
                x += b.nx + 1;
                row_h = std::max(row_h, b.ny);
                layer_h = std::max(layer_h, b.nz);
            }
            if ((double)NX * NY * NZ <= (double)(1 << 26)) break;
        }
        edge *= 1.26;
    }
    HIPCHK(c, slot->place.reserve((size_t)B));
    HIPCHK(c, hipMemcpy(slot->place.p, pl.data(), (size_t)B * sizeof(BatchPlace), hipMemcpyHostToDevice));
    GridDesc g{};
    g.nx = NX; g.ny = NY; g.nz = NZ; g.ncell = NX * NY * NZ;
    g.ox = c->lo[0]; g.oy = c->lo[1]; g.oz = c->lo[2];   // (not used for binning: every structure has its own origin)
    g.inv = 1.0 / edge;
    g.place = slot->place.p; g.sid_atom = c->sid_atom.p; g.sid_ring = c->sid_ring.p; g.sid_amide = c->sid_amide.p;
    slot->d = g; slot->radius = radius; slot->valid = true;
    d = g;
    return ARP_OK;
}
void batch_reset(arp_ctx* c) {   // a new upload: one structure until arp_set_batch says otherwise
    c->batch_n = 0;
    for (auto& g : c->batch_grid) g.valid = false;
}
// the grid of a pass over the resident structure(s)
int grid_desc_for(arp_ctx* c, GridDesc& d, const double lo[3], const double hi[3], double radius) {
    if (c->batch_n > 0) return batch_grid_desc(c, d, radius);
    make_grid_desc(d, lo, hi, radius);
    return ARP_OK;
}

// exclusive scan of the cell histogram: one launch up to 32768 cells, two (tiles + fix-up) above
int enqueue_scan(arp_ctx* c, Grid& G, u64* total_out = nullptr, hipStream_t st = nullptr) {
    if (!st) st = c->stream;
    const int ncell = G.d.ncell;
    Prof p(c, SLOT_SCAN, st);
    if (ncell <= 4096) {
        hipLaunchKernelGGL((k_scan_small<4>), dim3(1), dim3(1024), 0, st, G.cnt.p, ncell, G.start.p, total_out);
    } else if (ncell <= 16384) {
        hipLaunchKernelGGL((k_scan_small<16>), dim3(1), dim3(1024), 0, st, G.cnt.p, ncell, G.start.p, total_out);
    } else if (ncell <= 32768) {
        hipLaunchKernelGGL((k_scan_small<32>), dim3(1), dim3(1024), 0, st, G.cnt.p, ncell, G.start.p, total_out);
    } else {
        // tiles of 16384 counters, two launches (a 1 M-atom contact grid has 10 tiles)
        const int ntiles = (ncell + TILE_CELLS - 1) / TILE_CELLS;
        hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(1024), 0, st, G.cnt.p, ncell, G.start.p, G.sums.p);
        hipLaunchKernelGGL(k_scan_fix, dim3((ncell + 4095) / 4096), dim3(1024), 0, st, G.start.p, ncell, G.sums.p, ntiles, total_out);
    }
    return check_launch(c, "k_scan");
}

// the scans read / write whole int4s and whole tiles: histogram and start table are padded to a tile multiple
size_t scan_padded(int ncell) {
    return std::max<size_t>(((size_t)ncell / TILE_CELLS + 1) * TILE_CELLS + 8, 65536 + 8);
}

// grid buffers sized for the current descriptor; the histogram is zero on entry (see below)
int reserve_grid(arp_ctx* c, Grid& G, int n, hipStream_t st = nullptr) {
    if (!st) st = c->stream;
    const int ncell = G.d.ncell;
    HIPCHK(c, G.cell_of.reserve((size_t)std::max(n, 1)));
    HIPCHK(c, G.perm.reserve((size_t)std::max(n, 1)));
    {   // k_scatter's atomicSub takes every counter back to 0, so only a fresh allocation needs clearing
        bool fresh = false;
        HIPCHK(c, G.cnt.reserve(scan_padded(ncell), &fresh));
        if (fresh) HIPCHK(c, hipMemsetAsync(G.cnt.p, 0, G.cnt.cap * sizeof(int), st));
    }
    HIPCHK(c, G.start.reserve(scan_padded(ncell)));
    HIPCHK(c, G.sums.reserve((size_t)(ncell + TILE_CELLS - 1) / TILE_CELLS + 2));
    return ARP_OK;
}

// bin + scan + scatter (+ optional cell sort) for rings / amides.  P = point accessor.
template <class P>
int build_grid(arp_ctx* c, Grid& G, P pts, int n, const double lo[3], const double hi[3], double radius, const int* sid) {
    CHK(grid_desc_for(c, G.d, lo, hi, radius));
    G.radius = radius;
    G.n_points = n;
    const int ncell = G.d.ncell;
    CHK(reserve_grid(c, G, n));
    {
        Prof p(c, SLOT_BIN);
        if (n > 0) {
            hipLaunchKernelGGL((k_bin<P>), dim3(nblocks(n, 256)), dim3(256), 0, c->stream, pts, n, G.d, sid, G.cell_of.p, G.cnt.p);
            CHK(check_launch(c, "k_bin"));
        }
    }
    CHK(enqueue_scan(c, G));
    {
        Prof p(c, SLOT_SCATTER);
        if (n > 0) {
            hipLaunchKernelGGL(k_scatter, dim3(nblocks(n, 256)), dim3(256), 0, c->stream, n, G.cell_of.p, G.start.p, G.cnt.p, G.perm.p);
            // The order inside a cell does not change any result set (pairs are oriented by packed
            // id and callers sort); ARP_DETERMINISTIC=1 additionally fixes the device-side order.
            static const int deterministic = env_int("ARP_DETERMINISTIC", 0);
            if (deterministic)
                hipLaunchKernelGGL(k_cellsort, dim3(nblocks(ncell, 256)), dim3(256), 0, c->stream, ncell, G.start.p, G.perm.p);
            CHK(check_launch(c, "k_scatter/k_cellsort"));
        }
    }
    G.valid = true;
    return ARP_OK;
}

// The main stream waits for the grids / lists the last arp_set_blob left to the second stream (no-op when nothing is pending)
// (Inside a pass — uplists_defer — the callers that only prepare arguments ask without waiting: the one reader of the lists is the
// last launch of the pass, enqueue_contacts joins in front of it, and by then the event has usually fired: no wait on the stream,
// which would cost the main stream ~6 us between two of its kernels.)
int join_upload_lists(arp_ctx* c, bool must = false) {
    if (!c->uplists_pending) return ARP_OK;
    if (hipEventQuery(c->ev_uplists) == hipSuccess) { c->uplists_pending = false; return ARP_OK; }
    (void)hipGetLastError();      // (hipErrorNotReady is not an error)
    if (c->uplists_defer && !must) return ARP_OK;
    if (c->uplists_defer) {
        // In front of the last launch of a pass: the host is tens of microseconds ahead of the device there (grid build and search
        // are queued), the lists a few short of done — asking again for a little while costs the device nothing, a wait on the
        // stream costs it ~6 us between search and per-pair kernel.
        static const int spin_us = env_int("ARP_JOIN_SPIN_US", 25);
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < spin_us) {
            if (hipEventQuery(c->ev_uplists) == hipSuccess) { c->uplists_pending = false; return ARP_OK; }
            (void)hipGetLastError();
        }
    }
    c->uplists_pending = false;
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_uplists, 0));
    return ARP_OK;
}

// Selection-independent part of every atom record (rebuilt only when an input changed) and its SPATIAL ORDER: the rows sorted
// by cell of the grid with cell edge `radius` (counting sort, x fastest), which a pass with that cell edge compacts into its
// contact grid in one launch (k_compact_atoms).  radius = 0: any order will do (the one that exists, 6 A when there is none).
int ensure_static(arp_ctx* c, double radius = 0.0) {
    if (radius <= 0) radius = c->sp_radius > 0 ? c->sp_radius : 6.0;
    const bool columns = c->static_dirty;
    if (!columns && c->sp_radius == radius) return ARP_OK;
    const int n = (int)c->n;
    if (columns) {
        if (!c->lists_from_upload) c->lists_dirty = true;      // (lists made with the upload read what these columns are composed from)
        c->lists_from_upload = false;
        c->contacts_expected = 0;
        HIPCHK(c, c->st_qa.reserve((size_t)std::max(n, 1)));
        HIPCHK(c, c->st_h.reserve((size_t)std::max(n, 1)));
        HIPCHK(c, c->st_aux.reserve((size_t)std::max(n, 1)));
        HIPCHK(c, c->st_xyzm.reserve((size_t)std::max(n, 1)));
    }
    RawAtoms r;
    r.xyz = c->xyz.p; r.tmask = c->tmask.p; r.flags = c->flags.p; r.res_id = c->res_id.p;
    r.res_flags = c->has_res ? c->res_flags.p : nullptr;
    r.res_prev = c->has_res ? c->res_prev.p : nullptr;
    r.res_next = c->has_res ? c->res_next.p : nullptr;
    r.home = c->has_home ? c->home.p : nullptr;
    r.rad = c->rad.p; r.rad_idx = c->rad_idx.p; r.h_off = c->h_off.p; r.bond_off = c->bond_off.p; r.bond_idx = c->bond_idx.p; r.sb = c->sb.p;
    if (n > 0) {
        // counting sort by cell.  The longest-bond words sit behind the histogram, so ONE fill clears both.
        GridDesc d;
        CHK(grid_desc_for(c, d, c->lo, c->hi, radius));
        HIPCHK(c, c->sp_xyzm.reserve((size_t)n)); HIPCHK(c, c->sp_aux.reserve((size_t)n)); HIPCHK(c, c->sp_qa.reserve((size_t)n)); HIPCHK(c, c->sp_h.reserve((size_t)n));
        HIPCHK(c, c->sp_cr.reserve((size_t)n)); HIPCHK(c, c->sp_cell.reserve((size_t)n));
        // layout of sp_cnt: [longest bond, longest atom - hydrogen distance, 2 words of padding | histogram of ncell + 1 cells]:
        // a fresh structure clears all of it with ONE fill, a new order for resident columns only the histogram
        float keep_longest[2] = {0.0f, 0.0f};
        // (the one-block scan reads and writes whole int4s up to its 32768 slots; the tiled one of larger grids whole tiles)
        const size_t want = std::max<size_t>(d.ncell > 32768 ? scan_padded(d.ncell) + 8 : (size_t)d.ncell + 8, 32768 + 8);
        const bool regrow = !columns && c->sp_cnt.cap < want;
        if (regrow) {      // (a larger grid for resident columns: the two words survive the reallocation through the host)
            HIPCHK(c, hipMemcpyAsync(keep_longest, c->sp_cnt.p, sizeof(keep_longest), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        const bool cleared = c->sp_cnt.cap >= want && c->sp_cnt_zeroed >= want;      // (arp_set_blob did, while it waited for the validation)
        c->sp_cnt_zeroed = 0;
        HIPCHK(c, c->sp_cnt.reserve(want));
        int* const hist = c->sp_cnt.p + 4;
        c->longest_bond.borrow(c->sp_cnt.p, 2);
        if (columns) {
            if (!cleared) HIPCHK(c, hipMemsetAsync(c->sp_cnt.p, 0, want * sizeof(int), c->stream));
            hipLaunchKernelGGL(k_prepare_static, dim3(nblocks(n, 256)), dim3(256), 0, c->stream, r, n, c->st_xyzm.p, c->st_aux.p, c->st_qa.p, c->st_h.p,
                               d, hist, c->sp_cr.p, c->h_xyz_d.p, (unsigned int*)c->longest_bond.p,
                               c->ahead_seq ? (const int*)c->upload_bad.p : (const int*)nullptr, c->ahead_seq);
        } else {
            if (regrow) HIPCHK(c, hipMemcpyAsync(c->sp_cnt.p, keep_longest, sizeof(keep_longest), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemsetAsync(hist, 0, ((size_t)d.ncell + 4) * sizeof(int), c->stream));
            hipLaunchKernelGGL(k_static_bin, dim3(nblocks(n, 256)), dim3(256), 0, c->stream, n, c->st_xyzm.p, d, hist, c->sp_cr.p);
        }
        CHK(check_launch(c, "k_prepare_static"));
        if (d.ncell <= 4096) hipLaunchKernelGGL((k_scan_inplace<4>), dim3(1), dim3(1024), 0, c->stream, hist, d.ncell);
        else if (d.ncell <= 16384) hipLaunchKernelGGL((k_scan_inplace<16>), dim3(1), dim3(1024), 0, c->stream, hist, d.ncell);
        else if (d.ncell <= 32768) hipLaunchKernelGGL((k_scan_inplace<32>), dim3(1), dim3(1024), 0, c->stream, hist, d.ncell);
        else {      // larger grids (a batch of structures side by side: 10^6 cells): tiles of 16384 counters, two launches — one block
                    // walking them with a running carry was 177 us of a 64-structure batch's first pass
            const int ntiles = (d.ncell + TILE_CELLS - 1) / TILE_CELLS;
            HIPCHK(c, c->sp_sums.reserve((size_t)ntiles + 2));
            hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(1024), 0, c->stream, hist, d.ncell, hist, c->sp_sums.p);
            hipLaunchKernelGGL(k_scan_fix, dim3((d.ncell + 4095) / 4096), dim3(1024), 0, c->stream, hist, d.ncell, c->sp_sums.p, ntiles, (unsigned long long*)nullptr);
        }
        hipLaunchKernelGGL(k_static_permute, dim3(nblocks(n, 256)), dim3(256), 0, c->stream, n, c->sp_cr.p, hist, c->st_xyzm.p,
                           c->st_aux.p, c->st_qa.p, c->st_h.p, c->sp_xyzm.p, c->sp_aux.p, c->sp_qa.p, c->sp_h.p, c->sp_cell.p,
                           c->ahead_seq ? (const int*)c->upload_bad.p : (const int*)nullptr, c->ahead_seq);
        CHK(check_launch(c, "k_static_permute"));
        c->sp_grid = d;
    }
    else {      // no atoms: nothing to order; the two longest-distance words still exist (zero)
        c->sp_cnt_zeroed = 0;
        HIPCHK(c, c->sp_cnt.reserve(8));
        HIPCHK(c, hipMemsetAsync(c->sp_cnt.p, 0, 8 * sizeof(int), c->stream));
        c->longest_bond.borrow(c->sp_cnt.p, 2);
    }
    c->sp_radius = radius;
    c->static_dirty = false;
    ++c->static_epoch;      // (whatever was derived from the old columns or their order is stale)
    return ARP_OK;
}

StaticAtoms static_atoms(arp_ctx* c) {
    StaticAtoms r;
    r.xyzm = c->sp_xyzm.p;
    r.qa = c->sp_qa.p;
    r.hoff = c->sp_h.p;
    r.aux = c->sp_aux.p;
    r.sel = c->sel_made ? c->sel.p : nullptr;
    r.plus = c->sel_made ? c->plus.p : nullptr;
    r.all = (c->sel_made && c->sel_all) ? 1 : 0;
    return r;
}

// Grid over the atoms selected by the (req, forb) meta masks (or an explicit mask): three launches —
// k_bin_atoms (records composed on the fly), scan, k_scatter_atoms (writes the cell-sorted search and
// sift records directly).
int build_atom_grid(arp_ctx* c, Grid& G, DevBuf<float4>& sx, DevBuf<int4>& sa, DevBuf<int4>* srec, double radius,
                    uint32_t req, uint32_t forb, const uint8_t* active, u64* total_out = nullptr, uint8_t* plus_init = nullptr,
                    hipStream_t st = nullptr, ResMarks rm = ResMarks{nullptr, nullptr, 0}, GroupMasks gm = GroupMasks{}) {
    if (!st) st = c->stream;
    const int n = (int)c->n;
    CHK(grid_desc_for(c, G.d, c->lo, c->hi, radius));
    G.radius = radius;
    G.n_points = n;
    const int ncell = G.d.ncell;
    HIPCHK(c, G.cell_rank.reserve((size_t)std::max(n, 1)));
    {   // both histograms: zero when freshly allocated; afterwards each build clears the other one
        bool f0 = false, f1 = false;
        HIPCHK(c, G.cnt.reserve(scan_padded(ncell), &f0));
        HIPCHK(c, G.cnt2.reserve(scan_padded(ncell), &f1));
        if (f0) { HIPCHK(c, hipMemsetAsync(G.cnt.p, 0, G.cnt.cap * sizeof(int), st)); G.used[0] = 0; }
        if (f1) { HIPCHK(c, hipMemsetAsync(G.cnt2.p, 0, G.cnt2.cap * sizeof(int), st)); G.used[1] = 0; }
    }
    HIPCHK(c, G.start.reserve(scan_padded(ncell)));
    HIPCHK(c, G.sums.reserve((size_t)(ncell + TILE_CELLS - 1) / TILE_CELLS + 2));
    HIPCHK(c, sx.reserve((size_t)std::max(n, 1)));
    HIPCHK(c, sa.reserve((size_t)std::max(n, 1)));
    if (srec) { HIPCHK(c, srec->reserve((size_t)std::max(n, 1))); HIPCHK(c, c->s_h.reserve((size_t)std::max(n, 1))); }
    CHK(ensure_static(c));
    const StaticAtoms r = static_atoms(c);
    int* const hist = G.cur ? G.cnt2.p : G.cnt.p;
    int* const other = G.cur ? G.cnt.p : G.cnt2.p;
    if (n > 0) {
        {
            Prof p(c, SLOT_BIN, st);
            const int nzero = (int)G.used[1 - G.cur];
            if (active) hipLaunchKernelGGL((k_bin_atoms<1>), dim3(nblocks(n, 256)), dim3(256), 0, st, r, n, G.d, active, 0u, 0u, G.cell_rank.p, hist, plus_init, other, nzero, rm);
            else hipLaunchKernelGGL((k_bin_atoms<2>), dim3(nblocks(n, 256)), dim3(256), 0, st, r, n, G.d, (const uint8_t*)nullptr, req, forb, G.cell_rank.p, hist, plus_init, other, nzero, rm);
            CHK(check_launch(c, "k_bin_atoms"));
            G.used[1 - G.cur] = 0;
            G.used[G.cur] = ((size_t)ncell + 3) & ~(size_t)3;
        }
        int4* const rec = srec ? srec->p : (int4*)nullptr;
        const int nb = (n + SCAT_ATOMS - 1) / SCAT_ATOMS;
        if (ncell <= SCAN_LDS_CELLS) {   // start table in LDS: scan + scatter in one launch
            Prof p(c, SLOT_SCATTER, st);
            const int steps = (ncell + 16 * 256 - 1) / (16 * 256);
#define LAUNCH_SS(S) hipLaunchKernelGGL((k_scan_scatter_atoms<S>), dim3(nb), dim3(1024), (S) * 16384, st, r, n, ncell, G.cell_rank.p, hist, \
                                        G.start.p, total_out, sx.p, sa.p, rec, c->s_h.p, gm)
            switch (steps) {
                case 1: LAUNCH_SS(1); break;
                case 2: LAUNCH_SS(2); break;
                case 3: LAUNCH_SS(3); break;
                case 4: LAUNCH_SS(4); break;
                case 5: LAUNCH_SS(5); break;
                case 6: LAUNCH_SS(6); break;
                case 7: LAUNCH_SS(7); break;
                case 8: LAUNCH_SS(8); break;
                default: LAUNCH_SS(9); break;
            }
#undef LAUNCH_SS
            CHK(check_launch(c, "k_scan_scatter_atoms"));
        } else {
            {
                Prof p(c, SLOT_SCAN, st);
                const int ntiles = (ncell + TILE_CELLS - 1) / TILE_CELLS;
                static const int chained = env_int("ARP_CHAINED_SCAN", 1);
                if (chained && ntiles <= CHAIN_TILES && ntiles <= 2 * c->num_cu) {   // one launch: the tiles hand their totals on themselves (all resident at once)
                    bool fresh = false;
                    HIPCHK(c, G.chain.reserve(CHAIN_TILES, &fresh));
                    if (fresh || G.chain_epoch == 0xFFFFFFFFu) {   // new buffer, or the launch number wraps: no stale word may match a future one
                        HIPCHK(c, hipMemsetAsync(G.chain.p, 0, G.chain.cap * sizeof(unsigned long long), st));
                        G.chain_epoch = 0;
                    }
                    ++G.chain_epoch;
                    hipLaunchKernelGGL(k_scan_tiles_chained, dim3(ntiles), dim3(1024), 0, st, hist, ncell, G.start.p, G.chain.p, G.chain_epoch, total_out,
                                       (int*)(c->d_ctr + ctr_dev(C_ERR)));
                } else {
                    hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(1024), 0, st, hist, ncell, G.start.p, G.sums.p);
                    hipLaunchKernelGGL(k_scan_fix, dim3((ncell + 4095) / 4096), dim3(1024), 0, st, G.start.p, ncell, G.sums.p, ntiles, total_out);
                }
                CHK(check_launch(c, "k_scan"));
            }
            Prof p(c, SLOT_SCATTER, st);
            hipLaunchKernelGGL(k_scatter_atoms, dim3(nblocks(n, 256)), dim3(256), 0, st, r, n, G.cell_rank.p, G.start.p, sx.p, sa.p, rec, c->s_h.p, gm);
            CHK(check_launch(c, "k_scatter_atoms"));
        }
        G.cur = 1 - G.cur;
    } else {
        HIPCHK(c, hipMemsetAsync(G.start.p, 0, ((size_t)ncell + 1) * sizeof(int), st));
        if (total_out) HIPCHK(c, hipMemsetAsync(total_out, 0, sizeof(u64), st));
    }
    G.valid = true;
    G.n_binned = -1;  // known on the device only (start[ncell])
    return ARP_OK;
}
int build_contact_grid(arp_ctx* c, double radius, uint32_t req, uint32_t forb, const uint8_t* active, u64* total_out = nullptr,
                       ResMarks rm = ResMarks{nullptr, nullptr, 0}, GroupMasks gm = GroupMasks{}) {
    c->cg_valid = false;      // (the buffers of the pass's grid are rewritten)
    c->s_cell_valid = false;
    return build_atom_grid(c, c->atom_grid, c->s_xyzm, c->s_aux, &c->s_qa, radius, req, forb, active, total_out, nullptr, nullptr, rm, gm);
}
// The contact grid of a pass as an ordered compaction of the static columns (k_compact_atoms): ONE launch.
int build_contact_grid_compact(arp_ctx* c, double radius, uint32_t req, uint32_t forb, u64* total_out, uint8_t* plus_init, ResMarks rm) {
    Grid& G = c->atom_grid;
    const int n = (int)c->n;
    CHK(grid_desc_for(c, G.d, c->lo, c->hi, radius));
    G.radius = radius;
    G.n_points = n;
    const int ncell = G.d.ncell;
    HIPCHK(c, G.start.reserve(scan_padded(ncell)));
    HIPCHK(c, c->s_xyzm.reserve((size_t)std::max(n, 1)));
    HIPCHK(c, c->s_aux.reserve((size_t)std::max(n, 1)));
    HIPCHK(c, c->s_qa.reserve((size_t)std::max(n, 1)));
    HIPCHK(c, c->s_h.reserve((size_t)std::max(n, 1)));
    CHK(ensure_static(c, radius));
    if (n > 0) {
        Prof p(c, SLOT_BIN);
        static const int compact_small_max = env_int("ARP_COMPACT_512_MAX_ROWS", 150000);
        const int rows_per_block = n <= compact_small_max ? 512 : 1024;
        const int nb = (n + rows_per_block - 1) / rows_per_block;
        bool fresh = false;
        HIPCHK(c, c->compact_chain.reserve((size_t)nb, &fresh));
        if (fresh || c->compact_epoch >= (1u << 30) - 1u) {   // new buffer, or the 30-bit launch number wraps: no stale word may match
            HIPCHK(c, hipMemsetAsync(c->compact_chain.p, 0, c->compact_chain.cap * sizeof(unsigned long long), c->stream));
            c->compact_epoch = 0;
        }
        ++c->compact_epoch;
        CompactArgs A;
        A.r = static_atoms(c);
        A.sp_cell = c->sp_cell.p; A.n = n; A.ncell = ncell; A.req = req; A.forb = forb;
        A.s_xyzm = c->s_xyzm.p; A.s_aux = c->s_aux.p; A.s_qa = c->s_qa.p; A.s_h = c->s_h.p;
        HIPCHK(c, c->s_cell.reserve((size_t)std::max(n, 1)));
        A.s_cell = c->s_cell.p;
        A.start = G.start.p; A.chain = c->compact_chain.p; A.epoch = c->compact_epoch; A.total_out = total_out;
        A.plus_init = plus_init; A.rm = rm; A.err = (int*)(c->d_ctr + ctr_dev(C_ERR));
        if (rows_per_block == 512) hipLaunchKernelGGL(k_compact_atoms<512>, dim3(nb), dim3(512), 0, c->stream, A);
        else hipLaunchKernelGGL(k_compact_atoms<1024>, dim3(nb), dim3(1024), 0, c->stream, A);
        CHK(check_launch(c, "k_compact_atoms"));
        c->s_cell_valid = true;
    } else {
        HIPCHK(c, hipMemsetAsync(G.start.p, 0, ((size_t)ncell + 1) * sizeof(int), c->stream));
        if (total_out) HIPCHK(c, hipMemsetAsync(total_out, 0, sizeof(u64), c->stream));
    }
    G.valid = true;
    G.n_binned = -1;
    return ARP_OK;
}
// every atom (hydrogens included), used by the expansion and atom-plane; plus_init: also start selection_plus
int build_all_grid(arp_ctx* c, double radius, uint8_t* plus_init = nullptr, hipStream_t st = nullptr) {
    CHK(build_atom_grid(c, c->all_grid, c->a_xyzm, c->a_aux, nullptr, radius, 0, 0, nullptr, nullptr, plus_init, st));
    c->all_grid_current = true;
    return ARP_OK;
}

// Entries per segment of the contact pair list: an even number (the per-pair kernel fetches the pairs of a batch as 16-byte
// quads), and 64 entries short of the allocation (it reads up to a batch beyond the end of a segment and ignores what it gets).
inline size_t pair_segcap(size_t cap) { return cap > 64 ? ((cap - 64) / PAIR_SEGS) & ~(size_t)1 : 0; }
// Blocks of the neighbour search: ~ARP_SEARCH_CPW cells per wave, a multiple of 8 (one per XCD).
int search_blocks(const GridDesc& d, int cpw = 1) {
    int nb = (d.ncell + SEARCH_WAVES * cpw - 1) / (SEARCH_WAVES * cpw);
    nb = std::max(8, std::min(nb, 8192));
    return (nb + 7) & ~7;
}
// ... rounded to whole rounds of resident blocks once the launch comes near one: with 656 blocks on a chip that holds 768,
// the CUs given three blocks have half as much work again as those given two, and the kernel lasts as long as the former
// (100 k atoms: 38.3 -> 35.3 us with 768)
int search_blocks_balanced(const arp_ctx* c, const GridDesc& d, int cpw) {
    static const int forced = env_int("ARP_SEARCH_BLOCKS", 0);
    if (forced > 0) return (forced + 7) & ~7;
    const int nb = search_blocks(d, cpw), R = c->search_resident;
    if (R < 8 || 2 * nb < R) return nb;
    return std::min(std::max(1, (nb + R / 2) / R) * R, 8192) & ~7;
}

// zero a range of LOGICAL counters (callers outside a whole pass: a pass zeroes the block as it publishes it)
int zero_counter(arp_ctx* c, int first, int count) {
    if (c->ctr_clean) return ARP_OK;  // arp_run_launch cleared the whole block with one memset
    c->ctr_zero_ok = false;
    if ((first == C_STAT_CAND || first == C_STAT_MCAND) && count == 2 * STAT_SLOTS) {   // two neighbouring words of every slot line: one 2-D fill
        HIPCHK(c, hipMemset2DAsync(c->d_ctr + ctr_dev(first), sizeof(u64) * CTR_LINE, 0, 2 * sizeof(u64), STAT_SLOTS, c->stream));
        return ARP_OK;
    }
    if (first == C_SEG_PAIRS && count == PAIR_SEGS) {                                   // eight whole lines
        HIPCHK(c, hipMemsetAsync(c->d_ctr + ctr_dev(first), 0, sizeof(u64) * PAIR_SEGS * CTR_LINE, c->stream));
        return ARP_OK;
    }
    for (int k = first; k < first + count; ++k) HIPCHK(c, hipMemsetAsync(c->d_ctr + ctr_dev(k), 0, sizeof(u64), c->stream));
    return ARP_OK;
}
int enqueue_counter_copy(arp_ctx* c, int zero = 0) {  // counter block -> pinned host memory (capturable)
    ++c->publish_seq;
    hipLaunchKernelGGL(k_publish_counters, dim3(1), dim3(256), 0, c->stream, c->d_ctr, c->h_ctr_pinned, (int)C_COUNT, zero, c->publish_seq);
    CHK(check_launch(c, "k_publish_counters"));
    return ARP_OK;
}
int collect_counters(arp_ctx* c) {  // the only stream sync of a pass
    // The pass ends by storing its number in pinned memory (pass_end in the last kernel, or k_publish_counters):
    // polling that word wakes the host ~10 us sooner than hipStreamSynchronize.  Bounded: after 2 ms (or with a
    // caller-owned stream or ARP_SPIN_WAIT=0) the runtime's own wait takes over.
    static const int spin = env_int("ARP_SPIN_WAIT", 1);
    if (spin && !c->external_stream) {
        volatile u64* flag = c->h_ctr_pinned + C_COUNT;
        const auto t0 = std::chrono::steady_clock::now();
        for (int it = 0; *flag != c->publish_seq; ++it) {
            if ((it & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
            __builtin_ia32_pause();
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (*flag != c->publish_seq) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (*flag != c->publish_seq) FAIL(c, ARP_E_HIP, "the pass ended without publishing its counters");
        }
        else if ((c->publish_seq & 15) == 0) (void)hipStreamQuery(c->stream);   // lets the runtime retire finished commands
    } else {
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    memcpy(c->h_ctr, c->h_ctr_pinned, sizeof(u64) * C_COUNT);
    auto fold = [&](int first, int into) {
        u64 t = 0;
        for (int k = 0; k < STAT_SLOTS; ++k) t += c->h_ctr[first + k];
        c->h_ctr[into] = t;
    };
    fold(C_STAT_CAND, C_CAND); fold(C_STAT_ACC, C_ACC); fold(C_STAT_MCAND, C_MARK_CAND); fold(C_STAT_MACC, C_MARK_ACC);
    return ARP_OK;
}
int read_counters(arp_ctx* c) {
    CHK(enqueue_counter_copy(c));
    return collect_counters(c);
}

// CSR offsets: start at 0, never decrease (the kernels index with them unchecked)
bool csr_ok(const int32_t* off, int64_t n) {
    if (off[0] != 0) return false;
    for (int64_t i = 0; i < n; ++i)
        if (off[i + 1] < off[i]) return false;
    return true;
}

template <class T>
bool all_finite(const T* v, int64_t n) {
    for (int64_t i = 0; i < n; ++i)
        if (!std::isfinite((double)v[i])) return false;
    return true;
}

void host_bbox(const float* xyz, int64_t n, double lo[3], double hi[3]) {
    for (int k = 0; k < 3; ++k) { lo[k] = 0; hi[k] = 0; }
    for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            double v = xyz[i * 3 + k];
            if (i == 0 || v < lo[k]) lo[k] = v;
            if (i == 0 || v > hi[k]) hi[k] = v;
        }
}
void host_bbox_d(const double* xyz, int64_t n, double lo[3], double hi[3]) {
    for (int k = 0; k < 3; ++k) { lo[k] = 0; hi[k] = 0; }
    for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            double v = xyz[i * 3 + k];
            if (i == 0 || v < lo[k]) lo[k] = v;
            if (i == 0 || v > hi[k]) hi[k] = v;
        }
}

int ensure_ring_grid(arp_ctx* c) {
    CHK(join_upload_lists(c));
    if (c->ring_grid.valid) return ARP_OK;
    PtsD3 pts{c->ring_c.p};
    return build_grid<PtsD3>(c, c->ring_grid, pts, (int)c->nring, c->ring_lo, c->ring_hi, 6.0, c->sid_ring.p);
}
int ensure_amide_grid(arp_ctx* c) {
    CHK(join_upload_lists(c));
    if (c->amide_grid.valid) return ARP_OK;
    PtsF3 pts{c->am_c.p};
    return build_grid<PtsF3>(c, c->amide_grid, pts, (int)c->namide, c->am_lo, c->am_hi, 6.0, c->sid_amide.p);
}

// both centre grids in one launch when they are small (k_point_grids); otherwise each by the general path
int ensure_center_grids(arp_ctx* c) {
    CHK(join_upload_lists(c));      // (whoever asks for the grids on the main stream reads them there)
    static const int deterministic = env_int("ARP_DETERMINISTIC", 0);
    const bool want_r = c->nring > 0 && !c->ring_grid.valid, want_a = c->namide > 0 && !c->amide_grid.valid;
    const int small_points = 16384;
    if ((want_r || want_a) && !deterministic && c->nring <= small_points && c->namide <= small_points) {
        PointGridJob<PtsD3> jr{};
        PointGridJob<PtsF3> ja{};
        bool fits = true;
        if (want_r) {
            Grid& G = c->ring_grid;
            CHK(grid_desc_for(c, G.d, c->ring_lo, c->ring_hi, 6.0));
            fits = fits && G.d.ncell <= POINT_GRID_ITEMS * 1024;
        }
        if (want_a) {
            Grid& G = c->amide_grid;
            CHK(grid_desc_for(c, G.d, c->am_lo, c->am_hi, 6.0));
            fits = fits && G.d.ncell <= POINT_GRID_ITEMS * 1024;
        }
        if (fits) {
            if (want_r) {
                Grid& G = c->ring_grid;
                G.radius = 6.0; G.n_points = (int)c->nring;
                CHK(reserve_grid(c, G, (int)c->nring));
                jr = PointGridJob<PtsD3>{PtsD3{c->ring_c.p}, (int)c->nring, G.d, c->sid_ring.p, G.cell_of.p, G.cnt.p, G.start.p, G.perm.p};
            }
            if (want_a) {
                Grid& G = c->amide_grid;
                G.radius = 6.0; G.n_points = (int)c->namide;
                CHK(reserve_grid(c, G, (int)c->namide));
                ja = PointGridJob<PtsF3>{PtsF3{c->am_c.p}, (int)c->namide, G.d, c->sid_amide.p, G.cell_of.p, G.cnt.p, G.start.p, G.perm.p};
            }
            // (the entry counts of the candidate lists, which are rebuilt whenever the grids are, cleared on the way)
            const bool with_counts = c->lists_dirty || !c->plist_count.p;
            if (with_counts) { HIPCHK(c, c->plist_count.reserve(4)); c->lists_dirty = true; }
            Prof p(c, SLOT_BIN);
            hipLaunchKernelGGL(k_point_grids, dim3(2), dim3(1024), 0, c->stream, jr, ja, with_counts ? c->plist_count.p : (u64*)nullptr);
            CHK(check_launch(c, "k_point_grids"));
            c->plist_count_cleared = with_counts;
            if (want_r) c->ring_grid.valid = true;
            if (want_a) c->amide_grid.valid = true;
            return ARP_OK;
        }
    }
    if (c->nring > 0) CHK(ensure_ring_grid(c));
    if (c->namide > 0) CHK(ensure_amide_grid(c));
    return ARP_OK;
}

// ---- enqueue-only building blocks (no host synchronisation) -----------------------------------

// _make_selection, part 1 (I:1384-1424): selection_plus from the selection mask already in c->sel
int enqueue_expansion(arp_ctx* c, double radius, hipStream_t st = nullptr) {
    ++c->sel_epoch;
    if (!st) st = c->stream;
    const int n = (int)c->n;
    HIPCHK(c, c->plus.reserve((size_t)std::max(n, 1)));
    c->sel_made = true;   // (I:1407 selection_plus = selection is written by the binning kernel below)
    // I:1420-1424: search_all(6.0) over ALL atoms (hydrogens included)
    CHK(build_all_grid(c, radius, c->plus.p, st));
    CHK(zero_counter(c, C_STAT_MCAND, 2 * STAT_SLOTS));
    // With every atom selected, selection_plus = selection whatever the pairs are: the search is skipped (the
    // grid is still built, the atom-plane kernel walks it).
    if (n > 0 && !c->sel_all) {
        Prof p(c, SLOT_MARK, st);
        hipLaunchKernelGGL((k_search<MODE_MARK>), dim3(search_blocks(c->all_grid.d)), dim3(64 * SEARCH_WAVES), 0, st,
                           c->all_grid.d, c->all_grid.start.p, c->a_xyzm.p, c->a_aux.p, radius * radius, 1, 0, (int2*)nullptr,
                           0ull, c->d_ctr + ctr_dev(C_SCRATCH0), c->d_ctr + ctr_dev(C_STAT_MCAND), c->d_ctr + ctr_dev(C_STAT_MACC), c->plus.p, GroupMasks{}, (const int*)nullptr, (const int*)nullptr);
        CHK(check_launch(c, "k_search<MARK>"));
    }
    // all_grid stays usable for the atom-plane kernel: it reads plus[] directly, M_SEL is current
    c->contacts_valid = false;
    c->bag_ap.valid = c->bag_pp.valid = c->bag_gg.valid = c->bag_gp.valid = false;
    return ARP_OK;
}

int check_residue_ranges(arp_ctx* c) {   // the set kernels index the residue table with the uploaded ids, unchecked
    if (std::max({c->max_res_id, c->max_ring_res, c->max_amide_res}) >= std::max<int64_t>(c->nres, 1))
        FAIL(c, ARP_E_ARG, "selection sets: an atom, ring or amide refers to a residue beyond the residue table (arp_set_residues)");
    return ARP_OK;
}

// I:1413-1437: residue, ring and amide sets of the selection and of selection_plus.  Only the ring / amide
// kernels consume them, so arp_run_launch puts this stage on their stream.
int enqueue_selection_sets(arp_ctx* c, hipStream_t st) {
    const int n = (int)c->n;
    const size_t nres = (size_t)std::max<int64_t>(c->nres, 1);
    CHK(check_residue_ranges(c));
    HIPCHK(c, c->res_sel.reserve(2 * nres));   // [0, nres) = selection residues, [nres, 2 nres) = selection_plus residues
    uint8_t* res_sel = c->res_sel.p;
    uint8_t* res_plus = c->res_sel.p + nres;
    // arp_set_whole_structure: every residue of the (global) table is selected — also those whose atoms live on
    // another rank of a sharded run, which the local atoms could not tell
    const bool all_res = c->whole_structure && c->sel_all;
    HIPCHK(c, hipMemsetAsync(res_sel, all_res ? 1 : 0, 2 * nres, st));
    if (n > 0 && !all_res)
        hipLaunchKernelGGL(k_res_mark, dim3(nblocks(n, 256)), dim3(256), 0, st, n, c->res_id.p, c->sel.p, c->plus.p,
                           res_sel, res_plus);
    if (c->nring + c->namide > 0)
        hipLaunchKernelGGL(k_group_mask, dim3(nblocks(c->nring + c->namide, 256)), dim3(256), 0, st, (int)c->nring, (int)c->namide,
                           c->ring_res.p, c->am_res.p, res_sel, res_plus, c->ring_sel.p, c->ring_plus.p, c->am_sel.p, c->am_plus.p);
    return check_launch(c, "selection sets");
}

int enqueue_selection(arp_ctx* c, double radius) {   // the whole _make_selection on the main stream
    ++c->sel_epoch;
    CHK(enqueue_expansion(c, radius));
    return enqueue_selection_sets(c, c->stream);
}

// the default selection of a structure nobody uploaded one for: everything (I:1395 with no selectors)
int default_selection(arp_ctx* c) {
    if (c->sel_uploaded) return ARP_OK;
    const size_t n = (size_t)std::max<int64_t>(c->n, 1);
    HIPCHK(c, c->sel.reserve(n));
    if (!c->sel_prefilled) HIPCHK(c, hipMemsetAsync(c->sel.p, 1, n, c->stream));
    c->sel_prefilled = false;
    c->sel_all = true;
    c->sel_uploaded = true;
    return ARP_OK;
}

int ensure_default_selection(arp_ctx* c) {  // whole structure selected (I:1395 with no selectors)
    if (c->sel_made) return ARP_OK;
    CHK(default_selection(c));
    return enqueue_selection(c, 6.0);   // an uploaded selection that was not expanded yet is expanded here
}

int bag_reserve(arp_ctx* c, Bag& b, size_t cap, bool d, bool f) {
    cap = cap + cap / 4 + 64;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t bi = al(cap * sizeof(int)), bd = al(cap * sizeof(double)), bf = al(cap * sizeof(float)), bu = al(cap);
    const size_t total = 2 * bi + (d ? 4 * bd : 0) + (f ? 3 * bf : 0) + 3 * bu;
    b.release();
    HIPCHK(c, b.slab.reserve(total));
    uint8_t* p = b.slab.p;
    b.a.borrow(p, cap); p += bi; b.b.borrow(p, cap); p += bi;
    if (d) { b.d0.borrow(p, cap); p += bd; b.d1.borrow(p, cap); p += bd; b.d2.borrow(p, cap); p += bd; b.d3.borrow(p, cap); p += bd; }
    if (f) { b.f0.borrow(p, cap); p += bf; b.f1.borrow(p, cap); p += bf; b.f2.borrow(p, cap); p += bf; }
    b.u0.borrow(p, cap); p += bu; b.u1.borrow(p, cap); p += bu; b.u2.borrow(p, cap); p += bu;
    b.cap = cap;
    b.slab_bytes = total;
    return ARP_OK;
}

// Small bags travel to the host in one piece: the first *_fetch after a pass gathers the used prefix of every array of
// EVERY valid bag into one device buffer (k_pack_segments), copies it to page-locked memory and synchronises once; the
// fetch calls hand the arrays out from there — instead of six to nine small copies and a synchronisation per bag.
#define BAG_STAGE_MAX ((size_t)4 << 20)
int stage_bags(arp_ctx* c);
inline bool bag_is_staged(const arp_ctx* c, const Bag& b) { return c->bag_stage && b.staged_version == b.version && b.version != 0; }
template <class T>
int bag_download(arp_ctx* c, const Bag& b, T* dst, const DevBuf<T>& src, int idx, size_t m) {
    if (bag_is_staged(c, b)) {
        if (m && dst) memcpy(dst, c->bag_stage + b.stage_off[idx], m * sizeof(T));
        return ARP_OK;
    }
    return download_async(c, dst, src.p, m);
}

// One wavefront per ring / amide up to PLANE_BLOCKS blocks, several items per wave beyond: the waves queue their
// records in LDS and a block pays ONE counter atomic when it ends.
#define PLANE_BLOCKS 1024
// items (rings / amides) per wavefront of the ring / amide kernels: their candidate pairs gather in the wave's LDS queue
// until 64 are there for a full-lane evaluation, so a wave should see a few items; every item costs a few dependent
// loads, so not too many (sweep in profiles/README.md)
int plane_blocks(int64_t items) {
    static const int ipw = std::max(1, env_int("ARP_PLANE_IPW", 4));
    return nblocks((items + ipw - 1) / ipw * 64, 256, PLANE_BLOCKS);
}
// The four ring / amide kernels: prepare_* sizes the bag, clears its counter and fills the kernel arguments
// (nb = number of blocks, 0 when there is nothing to do); enqueue_* launches one of them, enqueue_planes all
// four as ONE launch (k_planes).
int prepare_atom_plane(arp_ctx* c, AtomPlaneArgs& a, int& nb, bool contact_grid = false) {  // I:947-1062
    Bag& b = c->bag_ap;
    nb = 0;
    if (!b.cap) CHK(bag_reserve(c, b, (size_t)c->nring * 8 + 256, true, false));
    CHK(zero_counter(c, C_AP, 1));
    if (c->nring == 0 || c->n == 0) return ARP_OK;
    if (contact_grid) {   // the grid of the pass (selection_plus without hydrogens): atom_plane_cg_body
        a = AtomPlaneArgs{c->atom_grid.d, c->atom_grid.start.p, c->s_xyzm.p, c->s_aux.p, (int)c->nring, c->ring_c.p, c->ring_n.p,
                          c->ring_res.p, c->ring_sel.p, c->ring_plus.p, c->plus.p, c->has_group_owner ? c->ring_home.p : nullptr,
                          c->has_group_owner ? c->ring_gid.p : nullptr, c->has_gid ? c->gid.p : nullptr, (long long)b.cap,
                          b.a.p, b.b.p, b.d0.p, b.d1.p, b.u0.p, b.u1.p, c->d_ctr + ctr_dev(C_AP),
                          c->st_xyzm.p, c->sel_made ? c->sel.p : nullptr, (c->sel_made && c->sel_all) ? 1 : 0};
        nb = plane_blocks(c->nring);
        return ARP_OK;
    }
    // all-atom 6 A grid (I:960 radius): the one the selection expansion has just built, if it is still current
    if (!(c->all_grid_current && c->all_grid.valid && c->all_grid.radius == 6.0)) CHK(build_all_grid(c, 6.0));
    a = AtomPlaneArgs{c->all_grid.d, c->all_grid.start.p, c->a_xyzm.p, c->a_aux.p, (int)c->nring, c->ring_c.p, c->ring_n.p,
                      c->ring_res.p, c->ring_sel.p, c->ring_plus.p, c->plus.p, c->has_group_owner ? c->ring_home.p : nullptr,
                      c->has_group_owner ? c->ring_gid.p : nullptr, c->has_gid ? c->gid.p : nullptr, (long long)b.cap,
                      b.a.p, b.b.p, b.d0.p, b.d1.p, b.u0.p, b.u1.p, c->d_ctr + ctr_dev(C_AP),
                      c->st_xyzm.p, c->sel_made ? c->sel.p : nullptr, (c->sel_made && c->sel_all) ? 1 : 0};
    nb = plane_blocks(c->nring);
    return ARP_OK;
}
int prepare_plane_plane(arp_ctx* c, PlanePlaneArgs& a, int& nb) {  // I:1064-1194
    Bag& b = c->bag_pp;
    nb = 0;
    if (!b.cap) CHK(bag_reserve(c, b, (size_t)c->nring * 16 + 256, true, false));
    CHK(zero_counter(c, C_PP, 1));
    if (c->nring == 0) return ARP_OK;
    CHK(ensure_ring_grid(c));
    a = PlanePlaneArgs{c->ring_grid.d, c->ring_grid.start.p, c->ring_grid.perm.p, (int)c->nring, c->ring_c.p, c->ring_n.p,
                       c->ring_res.p, c->ring_sel.p, c->ring_plus.p, c->has_group_owner ? c->ring_home.p : nullptr,
                       c->has_group_owner ? c->ring_gid.p : nullptr, (long long)b.cap, b.a.p, b.b.p, b.d0.p, b.d1.p, b.d2.p,
                       b.d3.p, b.u0.p, b.u1.p, b.u2.p, c->d_ctr + ctr_dev(C_PP)};
    nb = plane_blocks(c->nring);
    return ARP_OK;
}
int prepare_group_group(arp_ctx* c, GroupGroupArgs& a, int& nb) {  // I:1217-1300
    Bag& b = c->bag_gg;
    nb = 0;
    if (!b.cap) CHK(bag_reserve(c, b, (size_t)c->namide * 8 + 256, false, true));
    CHK(zero_counter(c, C_GG, 1));
    if (c->namide == 0) return ARP_OK;
    CHK(ensure_amide_grid(c));
    a = GroupGroupArgs{c->amide_grid.d, c->amide_grid.start.p, c->amide_grid.perm.p, (int)c->namide, c->am_c.p, c->am_n.p,
                       c->am_sel.p, c->am_plus.p, c->has_group_owner ? c->am_home.p : nullptr,
                       c->has_group_owner ? c->am_gid.p : nullptr, (long long)b.cap, b.a.p, b.b.p, b.f0.p, b.f1.p, b.f2.p,
                       b.u0.p, c->d_ctr + ctr_dev(C_GG)};
    nb = plane_blocks(c->namide);
    return ARP_OK;
}
int prepare_group_plane(arp_ctx* c, GroupPlaneArgs& a, int& nb) {  // I:1302-1382
    Bag& b = c->bag_gp;
    nb = 0;
    if (!b.cap) CHK(bag_reserve(c, b, (size_t)c->namide * 8 + 256, true, false));
    CHK(zero_counter(c, C_GP, 1));
    if (c->namide == 0 || c->nring == 0) return ARP_OK;
    CHK(ensure_ring_grid(c));
    a = GroupPlaneArgs{c->ring_grid.d, c->ring_grid.start.p, c->ring_grid.perm.p, (int)c->namide, c->am_c.p, c->am_n.p,
                       c->am_sel.p, c->am_plus.p, c->ring_c.p, c->ring_n.p, c->ring_sel.p, c->ring_plus.p,
                       c->has_group_owner ? c->am_home.p : nullptr, c->has_group_owner ? c->am_gid.p : nullptr,
                       c->has_group_owner ? c->ring_gid.p : nullptr, (long long)b.cap, b.a.p, b.b.p, b.d0.p, b.d1.p, b.d2.p,
                       b.u0.p, c->d_ctr + ctr_dev(C_GP)};
    nb = plane_blocks(c->namide);
    return ARP_OK;
}

int enqueue_atom_plane(arp_ctx* c, hipStream_t st) {
    AtomPlaneArgs a{};
    int nb = 0;
    CHK(prepare_atom_plane(c, a, nb));
    if (!nb) return ARP_OK;
    Prof p(c, SLOT_PLANES, st);
    hipLaunchKernelGGL(k_atom_plane, dim3(nb), dim3(256), 0, st, a);
    return check_launch(c, "k_atom_plane");
}
int enqueue_plane_plane(arp_ctx* c, hipStream_t st) {
    PlanePlaneArgs a{};
    int nb = 0;
    CHK(prepare_plane_plane(c, a, nb));
    if (!nb) return ARP_OK;
    Prof p(c, SLOT_PLANES, st);
    hipLaunchKernelGGL(k_plane_plane, dim3(nb), dim3(256), 0, st, a);
    return check_launch(c, "k_plane_plane");
}
int enqueue_group_group(arp_ctx* c, hipStream_t st) {
    GroupGroupArgs a{};
    int nb = 0;
    CHK(prepare_group_group(c, a, nb));
    if (!nb) return ARP_OK;
    Prof p(c, SLOT_PLANES, st);
    hipLaunchKernelGGL(k_group_group, dim3(nb), dim3(256), 0, st, a);
    return check_launch(c, "k_group_group");
}
int enqueue_group_plane(arp_ctx* c, hipStream_t st) {
    GroupPlaneArgs a{};
    int nb = 0;
    CHK(prepare_group_plane(c, a, nb));
    if (!nb) return ARP_OK;
    Prof p(c, SLOT_PLANES, st);
    hipLaunchKernelGGL(k_group_plane, dim3(nb), dim3(256), 0, st, a);
    return check_launch(c, "k_group_plane");
}
// Static candidate lists of the ring / amide loops (arp_planes.h): built on the stream when the structure changed.
PlaneLists plane_lists(arp_ctx* c) {
    PlaneLists L;
    for (int k = 0; k < 4; ++k) { L.pairs[k] = c->plist[k].p; L.cap[k] = (long long)c->plist[k].cap; }
    L.count = c->plist_count.p;
    return L;
}
int ensure_plane_lists(arp_ctx* c) {
    CHK(join_upload_lists(c));
    if (!c->lists_dirty && c->plist_count.p) return ARP_OK;
    const size_t want[4] = {(size_t)c->nring * 96 + 256, (size_t)c->nring * 16 + 256, (size_t)c->namide * 16 + 256, (size_t)c->namide * 8 + 256};
    for (int k = 0; k < 4; ++k) HIPCHK(c, c->plist[k].reserve(want[k]));
    HIPCHK(c, c->plist_count.reserve(4));
    if (!c->plist_count_cleared) HIPCHK(c, hipMemsetAsync(c->plist_count.p, 0, 4 * sizeof(u64), c->stream));   // (else: k_point_grids did, on this stream)
    c->plist_count_cleared = false;
    AtomRingListArgs ar{};
    PlanePlaneArgs pp{};
    GroupGroupArgs gg{};
    GroupPlaneArgs gp{};
    int n0 = 0, n1 = 0, n2 = 0, n3 = 0;
    {   // (argument blocks and bag buffers only: the counters of the output bags are the evaluation's affair — a pass clears them
        // itself, and at upload time each would be a fill launch in front of the lists)
        const bool was_clean = c->ctr_clean;
        c->ctr_clean = true;
        int rc = prepare_plane_plane(c, pp, n1);      // (ring / amide grids: built here if need be)
        if (rc == ARP_OK) rc = prepare_group_group(c, gg, n2);
        if (rc == ARP_OK) rc = prepare_group_plane(c, gp, n3);
        c->ctr_clean = was_clean;
        CHK(rc);
    }
    // atom-plane candidates atom by atom against the ring grid (ar_enumerate): the atoms as uploaded, no atom grid
    if (c->nring > 0 && c->n > 0) {
        ar = AtomRingListArgs{c->ring_grid.d, c->ring_grid.start.p, c->ring_grid.perm.p, (int)c->n, c->xyz.p, c->tmask.p, c->flags.p, c->ring_c.p};
        n0 = nblocks(c->n, 256, 4096);
    }
    // one wavefront per ring / amide: a stencil walk is a chain of dependent loads, and this kernel runs alone
    if (n1) n1 = nblocks(c->nring * 64, 256, 4096);
    if (n2) n2 = nblocks(c->namide * 64, 256, 4096);
    if (n3) n3 = nblocks(c->namide * 64, 256, 4096);
    if (n0 + n1 + n2 + n3 > 0) {
        hipLaunchKernelGGL(k_plane_lists, dim3(n0 + n1 + n2 + n3), dim3(256), 0, c->stream, ar, pp, gg, gp, plane_lists(c), n0, n0 + n1,
                           n0 + n1 + n2, n0 + n1 + n2 + n3);
        CHK(check_launch(c, "k_plane_lists"));
    }
    c->lists_dirty = false;
    return ARP_OK;
}

// The pass proper: _calculate_atom_contacts (I:693-936) = contact grid (bin + scan/scatter), neighbour search, fused
// per-pair kernel; with_planes: the four ring / amide loops (I:938-1382) ride in the last launch (k_sift_planes).
// fuse_sets (arp_run_launch): the grid build also produces the residue / ring / amide sets of I:1413-1437.
int enqueue_contacts(arp_ctx* c, double cutoff, double vdw_comp, int include_seq_adj, bool with_planes) {
    struct Defer { arp_ctx* c; ~Defer() { c->uplists_defer = false; } } defer{c};
    c->uplists_defer = !c->external_stream;
    // the tree is built on selection_plus (I:1442); hydrogens are dropped at I:712
    if (!c->ctr_clean) CHK(zero_counter(c, C_BINNED, 1));
    CHK(zero_counter(c, C_ERR, 1));      // (before the grid build: its chained scan may raise the flag)
    ResMarks rm{nullptr, nullptr, 0};
    GroupMasks gm{};
    bool masks_after_bin = false;
    // the grid of the previous whole-structure pass, if nothing it depends on has changed (see grid_reuse)
    const bool all_res_now = c->whole_structure && c->sel_all;
    const bool whole = c->sel_made && c->sel_all && c->n > 0;
    const bool reuse_grid = c->grid_reuse && whole && c->cg_valid && !c->cg_pending && c->atom_grid.valid && c->cg_radius == cutoff &&
                            c->cg_static_epoch == c->static_epoch && c->cg_sel_epoch == c->sel_epoch && c->cg_fuse == c->fuse_sets &&
                            c->cg_init_plus == c->init_plus_in_bin && c->cg_all_res == all_res_now && !c->static_dirty && c->sp_radius == cutoff;
    c->cg_reused = reuse_grid;
    if (c->fuse_sets) {
        const size_t nres = (size_t)std::max<int64_t>(c->nres, 1);
        CHK(check_residue_ranges(c));
        bool fresh = false;
        HIPCHK(c, c->res_tag.reserve(2 * nres, &fresh));
        if (!reuse_grid && (fresh || c->res_tag_value >= 255)) {
            HIPCHK(c, hipMemsetAsync(c->res_tag.p, 0, c->res_tag.cap, c->stream));
            c->res_tag_value = 0;
        }
        // (a reused grid keeps the residue tags its build left: the masks below are made from the same tag value)
        const uint8_t tag = reuse_grid ? (uint8_t)c->res_tag_value : (uint8_t)++c->res_tag_value;
        const bool all_res = c->whole_structure && c->sel_all;   // every residue of the (global) table is selected
        if (!all_res) rm = ResMarks{c->res_tag.p, c->res_tag.p + nres, tag};
        gm = GroupMasks{(int)c->nring, (int)c->namide, c->ring_res.p, c->am_res.p, c->res_tag.p, c->res_tag.p + nres, tag,
                        all_res ? 1 : 0, c->ring_sel.p, c->ring_plus.p, c->am_sel.p, c->am_plus.p};
        if (c->n == 0) masks_after_bin = true;   // no scatter launch to carry them
    }
    // The static candidate lists of a structure's ring / amide loops — two centre grids, k_plane_lists — are made with its upload
    // when it comes as a blob (arp_set_blob: on the second stream, behind the validation kernel, joined below).  Structures set up
    // by the classic setters or assembled from shards build them in their first pass: nothing before the last launch of the pass
    // reads them, so they go on the second stream, beside the grid build and the search of this pass, and are enqueued AFTER the
    // search (the host needs ~5 us per launch: enqueued first they held the main stream's kernels back by as much).
    // (The other way round — search on the second stream, lists on the main one, so that the last launch follows the longer
    // chain in stream order — was no faster: the search then shares the chip with the 1024-thread scan blocks of the lists.)
    const bool fork_lists = with_planes && c->nring + c->namide > 0 && c->n > 0 &&
                            (c->lists_dirty || !c->plist_count.p || (c->nring > 0 && !c->ring_grid.valid) || (c->namide > 0 && !c->amide_grid.valid)) &&
                            !c->external_stream && c->stream2;
    if (fork_lists) {
        CHK(ensure_static(c, cutoff));                                  // (what the lists read is in place on the main stream here)
        HIPCHK(c, hipEventRecord(c->ev_sel, c->stream));
    }
    if (!reuse_grid) {
        c->cg_valid = false;
        CHK(build_contact_grid_compact(c, cutoff, M_PLUS, M_HYDROGEN, c->d_ctr + ctr_dev(C_BINNED), c->init_plus_in_bin ? c->plus.p : nullptr, rm));
        if (whole) {      // what this grid was built from; its atom count arrives with the counters of the pass (finish_contacts)
            c->cg_valid = true; c->cg_pending = true; c->cg_radius = cutoff; c->cg_static_epoch = c->static_epoch; c->cg_sel_epoch = c->sel_epoch;
            c->cg_fuse = c->fuse_sets; c->cg_init_plus = c->init_plus_in_bin; c->cg_all_res = all_res_now;
        }
    }
    if (masks_after_bin && c->nring + c->namide > 0) {
        hipLaunchKernelGGL(k_group_masks, dim3(nblocks(c->nring + c->namide, 256)), dim3(256), 0, c->stream, gm);
        CHK(check_launch(c, "k_group_masks"));
    }
    c->contact_cells = c->atom_grid.d.ncell;
    if (!c->pairs.p) HIPCHK(c, c->pairs.reserve((size_t)c->n * 16 + 8192));
    const size_t segcap = pair_segcap(c->pairs.cap);   // the pair list is PAIR_SEGS segments of segcap entries
    const size_t cap = segcap * PAIR_SEGS;
    if (cap >= ((size_t)1 << 32)) FAIL(c, ARP_E_CAPACITY, "contact list beyond 2^32 entries (k_sift's task queue holds 32-bit output indices)");
    HIPCHK(c, c->out_i.reserve(cap)); HIPCHK(c, c->out_j.reserve(cap)); HIPCHK(c, c->out_d.reserve(cap));
    HIPCHK(c, c->out_s.reserve(cap)); HIPCHK(c, c->out_ct.reserve(cap));
    CHK(zero_counter(c, C_SEG_PAIRS, PAIR_SEGS));
    CHK(zero_counter(c, C_STAT_CAND, 2 * STAT_SLOTS));
    bool search_launched = false;
    auto launch_search = [&]() -> int {
        if (search_launched || c->n == 0) return ARP_OK;
        search_launched = true;
        Prof p(c, SLOT_SEARCH);
        // the contact search ends with a block-level flush of its pair queues, which amortises better over
        // ~3 cells per wave; the flush-free expansion search prefers 1 (sweeps in profiles/README.md)
        // Small grids (a protein of a few thousand atoms has ~1000 cells) get fewer cells per wave: the chip is far from full
        // and a wave's cells are a serial chain (1tqn_h stand-in: 22 -> 16 us).
        // (twelve since the waves of a block CLAIM their work: a longer list per block evens out better — 1 M atoms 196 -> 176 us —;
        // three was the optimum of the static three-cells-per-wave assignment)
        static const int cpw_max = std::max(1, env_int("ARP_SEARCH_CPW", 12));
        const int cpw = std::max(1, std::min(cpw_max, c->atom_grid.d.ncell / (SEARCH_WAVES * 2 * c->num_cu)));
        // tiles of two x-adjacent cells (k_search): fewer, fuller units — worth it where a wave still has several to claim
        static const int tile_mode = env_int("ARP_SEARCH_TILE", -1);      // -1: by size, 1 / 2: always
        static const int tile_min_cells = env_int("ARP_SEARCH_TILE_MIN_CELLS", 60000);
        const int tile_x = tile_mode > 0 ? std::min(tile_mode, 2) : (c->atom_grid.d.ncell >= tile_min_cells ? 2 : 1);
        // Sparse grids (fewer atoms than cells: a protein in its box, a ligand's selection_plus, a batch): the blocks split the
        // ATOMS of the grid evenly instead of its cells (k_search, cell_of_pos) and their number goes with the atoms — what the
        // grid's build reported last time, all atoms of the structure before that is known.
        static const int balance_mode = env_int("ARP_SEARCH_BALANCE", -1);     // -1: by sparsity, 0: never, 1: whenever the cells of the rows are known
        const int64_t atoms_est = c->cg_reused ? c->cg_binned : (c->stats[4] == c->atom_grid.d.ncell && c->stats[3] > 0 ? c->stats[3] : c->n);
        // ... and so do the blocks of a CLUMPED structure (cells with hundreds of atoms: a chain folded onto itself), which the
        // previous pass over this structure gives away by its distance tests per atom (uniform protein density: ~80)
        const bool clumped = c->stats[3] > 0 && c->stats[4] == c->atom_grid.d.ncell && c->stats[0] > 150 * c->stats[3];
        const bool by_atoms = c->s_cell_valid && (balance_mode == 1 || (balance_mode < 0 && ((int64_t)c->atom_grid.d.ncell > atoms_est || clumped)));
        int nblocks_search = search_blocks_balanced(c, c->atom_grid.d, cpw);
        if (by_atoms) {
            const int R = c->search_resident;
            static const int atoms_per_block = std::max(8, env_int("ARP_SEARCH_APB", 128));
            int nbk = (int)std::min<int64_t>(std::max<int64_t>((atoms_est + atoms_per_block - 1) / atoms_per_block, 8), 8192);
            nbk = (nbk + 7) & ~7;
            nblocks_search = (R >= 8 && 2 * nbk >= R) ? std::min(std::max(1, (nbk + R / 2) / R) * R, 8192) & ~7 : nbk;
        }
        static const int hint_mode = env_int("ARP_SEARCH_BALANCE_HINT", 1);
        static const int cell_w16 = std::max(0, env_int("ARP_SEARCH_CELL_WEIGHT_X16", 64));      // weight of a cell in sixteenths of an atom
        static const int hint_atoms = env_int("ARP_SEARCH_BALANCE_HINT_ATOMS", 1);      // ... also for blocks that take runs of atoms (k_balance_atoms)
        const bool hint_wanted = hint_mode && whole && (!by_atoms || hint_atoms);
        const bool hint_ok = hint_wanted && c->sb_valid && c->sb_by_atoms == by_atoms && c->sb_static_epoch == c->static_epoch && c->sb_sel_epoch == c->sel_epoch &&
                             c->sb_radius == cutoff && c->sb_blocks == nblocks_search && c->sb_tx == tile_x && c->sb_tile.p;
        const int* const balance_hint = hint_ok ? (const int*)c->sb_tile.p : (const int*)nullptr;
        if (tile_x == 2) {
            hipLaunchKernelGGL((k_search<MODE_CONTACTS, 2>), dim3(nblocks_search), dim3(64 * SEARCH_WAVES), 0,
                               c->stream, c->atom_grid.d, c->atom_grid.start.p, c->s_xyzm.p, c->s_aux.p, cutoff * cutoff,
                               include_seq_adj, c->has_home ? 1 : 0, c->pairs.p, (u64)segcap, c->d_ctr + ctr_dev(C_SEG_PAIRS), c->d_ctr + ctr_dev(C_STAT_CAND),
                               c->d_ctr + ctr_dev(C_STAT_ACC), (uint8_t*)nullptr, masks_after_bin ? GroupMasks{} : gm,
                               by_atoms ? (const int*)c->s_cell.p : (const int*)nullptr, balance_hint);
        } else {
            hipLaunchKernelGGL((k_search<MODE_CONTACTS>), dim3(nblocks_search), dim3(64 * SEARCH_WAVES), 0,
                               c->stream, c->atom_grid.d, c->atom_grid.start.p, c->s_xyzm.p, c->s_aux.p, cutoff * cutoff,
                               include_seq_adj, c->has_home ? 1 : 0, c->pairs.p, (u64)segcap, c->d_ctr + ctr_dev(C_SEG_PAIRS), c->d_ctr + ctr_dev(C_STAT_CAND),
                               c->d_ctr + ctr_dev(C_STAT_ACC), (uint8_t*)nullptr, masks_after_bin ? GroupMasks{} : gm,
                               by_atoms ? (const int*)c->s_cell.p : (const int*)nullptr, balance_hint);
        }
        // the hint for the LATER passes over this grid (same structure, selection, cutoff, launch shape), made behind the search of
        // the SECOND pass over it: the launch sits between the search and the per-pair kernel of that pass (6 us + a gap: a tenth
        // of a structure's first pass when it was made there), and a structure that is evaluated once never needs it.  First and
        // second pass run on equal runs of cells.
        if (hint_wanted && !hint_ok && nblocks_search >= 16) {
            const bool seen = c->sb_seen && c->sb_seen_by_atoms == by_atoms && c->sb_seen_static_epoch == c->static_epoch && c->sb_seen_sel_epoch == c->sel_epoch &&
                              c->sb_seen_radius == cutoff && c->sb_seen_blocks == nblocks_search && c->sb_seen_tx == tile_x;
            static const int hint_eager = env_int("ARP_SEARCH_BALANCE_HINT_EAGER", 0);
            if (seen || hint_eager) {
                HIPCHK(c, c->sb_tile.reserve((size_t)nblocks_search + 1));
                if (!by_atoms) {
                    hipLaunchKernelGGL(k_balance_blocks, dim3(nblocks(nblocks_search + 1, 256)), dim3(256), 0, c->stream, c->atom_grid.d, c->atom_grid.start.p,
                                       nblocks_search, tile_x, cell_w16, c->sb_tile.p);
                } else {
                    // runs of ATOMS of equal weight: weight per cell, its running sum (the scans of the grid builds), one binary search per block
                    const int ncell = c->atom_grid.d.ncell;
                    static const int w_unit = std::max(0, env_int("ARP_SEARCH_W_UNIT", 100)), w_chunk = std::max(0, env_int("ARP_SEARCH_W_CHUNK", 85)),
                                     w_test8 = std::max(0, env_int("ARP_SEARCH_W_TEST_X8", 2));      // (k_cell_weights: wave instructions per unit, per chunk, per eight distance tests)
                    HIPCHK(c, c->sb_cw.reserve(scan_padded(ncell))); HIPCHK(c, c->sb_cwp.reserve(scan_padded(ncell)));
                    hipLaunchKernelGGL(k_cell_weights, dim3(nblocks(ncell, 256, 2048)), dim3(256), 0, c->stream, c->atom_grid.d, c->atom_grid.start.p, w_unit, w_chunk, w_test8, c->sb_cw.p);
                    if (ncell <= 4096) hipLaunchKernelGGL((k_scan_small<4>), dim3(1), dim3(1024), 0, c->stream, c->sb_cw.p, ncell, c->sb_cwp.p, (u64*)nullptr);
                    else if (ncell <= 16384) hipLaunchKernelGGL((k_scan_small<16>), dim3(1), dim3(1024), 0, c->stream, c->sb_cw.p, ncell, c->sb_cwp.p, (u64*)nullptr);
                    else if (ncell <= 32768) hipLaunchKernelGGL((k_scan_small<32>), dim3(1), dim3(1024), 0, c->stream, c->sb_cw.p, ncell, c->sb_cwp.p, (u64*)nullptr);
                    else {
                        const int ntiles = (ncell + TILE_CELLS - 1) / TILE_CELLS;
                        HIPCHK(c, c->sb_sums.reserve((size_t)ntiles + 2));
                        hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(1024), 0, c->stream, c->sb_cw.p, ncell, c->sb_cwp.p, c->sb_sums.p);
                        hipLaunchKernelGGL(k_scan_fix, dim3((ncell + 4095) / 4096), dim3(1024), 0, c->stream, c->sb_cwp.p, ncell, c->sb_sums.p, ntiles, (unsigned long long*)nullptr);
                    }
                    hipLaunchKernelGGL(k_balance_atoms, dim3(nblocks(nblocks_search + 1, 256)), dim3(256), 0, c->stream, c->atom_grid.d, c->atom_grid.start.p,
                                       (const int*)c->sb_cwp.p, nblocks_search, c->sb_tile.p);
                }
                c->sb_valid = true; c->sb_by_atoms = by_atoms; c->sb_static_epoch = c->static_epoch; c->sb_sel_epoch = c->sel_epoch; c->sb_radius = cutoff;
                c->sb_blocks = nblocks_search; c->sb_tx = tile_x; c->sb_whole = whole;
            } else {
                c->sb_seen = true; c->sb_seen_by_atoms = by_atoms; c->sb_seen_static_epoch = c->static_epoch; c->sb_seen_sel_epoch = c->sel_epoch; c->sb_seen_radius = cutoff;
                c->sb_seen_blocks = nblocks_search; c->sb_seen_tx = tile_x;
            }
        }
        return check_launch(c, "k_search<CONTACTS>");
    };
    // the per-pair kernel of the pass; merged_: with the list blocks of the ring / amide loops in front (np of them)
    AtomPlaneArgs ap{};
    PlanePlaneArgs pp{};
    GroupGroupArgs gg{};
    GroupPlaneArgs gp{};
    int np = 0;
    bool sift_launched = false;
    auto launch_sift = [&](bool merged_) -> int {
        sift_launched = true;
        CHK(join_upload_lists(c, /*must=*/true));      // (the list blocks of this launch read what the upload's second stream made)
        Prof p(c, SLOT_SIFT);
        // (the blocks split their work statically: all of them must be resident from the start)
        static const int sift_bpc_env = env_int("ARP_SIFT_BPC", 0);
        const int sift_blocks_per_cu = sift_bpc_env > 0 ? sift_bpc_env : c->sift_per_cu;
        SiftArgs sa{c->pairs.p, c->d_ctr + ctr_dev(C_SEG_PAIRS), (u64)segcap, c->s_xyzm.p, c->s_qa.p, c->s_h.p, {0, 0, 0, 0, 0, 0, 0, 0},
                          SiftSide{c->rad_tab.p, c->rad.p, c->xyz.p, c->h_off.p, c->bond_off.p, c->sb.p, c->longest_bond.p}, c->bond_idx.p, c->h_xyz_d.p,
                          c->has_gid ? c->gid.p : nullptr, vdw_comp, c->out_i.p, c->out_j.p, c->out_d.p, c->out_s.p, c->out_ct.p,
                          (int*)(c->d_ctr + ctr_dev(C_ERR)), c->xcd_round_robin ? 0 : 1};
        // No more sift blocks than the pairs can feed (one batch of 64 per wave and block at least): what the previous pass over
        // this structure found, or ~13 per heavy atom for the first one.  A protein-sized structure then runs 70-odd blocks
        // instead of 1024, whose start-up and end-of-pass tickets were most of the kernel (stand-in: 25 -> 16 us).
        static const int pairs_per_block = std::max(64, env_int("ARP_SIFT_PPB", 256));
        const int64_t expect = (c->contacts_expected > 0) ? c->contacts_expected : (int64_t)c->n * 13;
        static const int64_t stream_bytes = (int64_t)env_int("ARP_STREAM_OUT_MB", 96) << 20;
        const bool stream_out = expect * 15 > stream_bytes;     // (15 B per record)
        const int by_work = (int)std::min<int64_t>((expect + pairs_per_block - 1) / pairs_per_block + PAIR_SEGS, 1 << 20);
        const int slots = merged_ ? std::max(c->num_cu * sift_blocks_per_cu - np, 8 * PAIR_SEGS) : c->num_cu * sift_blocks_per_cu;
        const int nsift = std::max(std::min(slots, by_work), 8 * PAIR_SEGS) & ~(PAIR_SEGS - 1);
        // (four variants each: streaming stores or not, global ids or not — template parameters of the kernels, see sift_body)
        {   // a segment's share of the sift blocks goes with its size in the pass before (k_sift: nblk)
            static const int shares = env_int("ARP_SIFT_SHARES", 1);
            const int B = nsift / PAIR_SEGS;
            uint64_t tot = 0;
            for (int k = 0; k < PAIR_SEGS; ++k) tot += c->seg_last[k];
            int given = 0, big = 0;
            for (int k = 0; k < PAIR_SEGS; ++k) {
                sa.nblk[k] = (shares && tot > 0 && B > 1) ? std::max(1, (int)((double)nsift * (double)c->seg_last[k] / (double)tot + 0.5)) : B;
                given += sa.nblk[k];
                if (sa.nblk[k] > sa.nblk[big]) big = k;
            }
            // the shares add up to the launch (the largest one takes the rounding; never below one block)
            while (given != nsift) {
                const int step = given > nsift ? -1 : 1;
                if (sa.nblk[big] + step < 1) { for (int k = 0; k < PAIR_SEGS; ++k) sa.nblk[k] = B; break; }
                sa.nblk[big] += step; given += step;
                big = 0;
                for (int k = 1; k < PAIR_SEGS; ++k) if (sa.nblk[k] > sa.nblk[big]) big = k;
            }
        }
#define LAUNCH_SIFT_PLANES(S, G) hipLaunchKernelGGL((k_sift_planes<S, G>), dim3(np + nsift), dim3(256), 0, c->stream, sa, nsift, ap, pp, gg, gp, plane_lists(c), \
                                                    c->d_ctr + ctr_dev(C_PLIST), np, c->pub)
#define LAUNCH_SIFT(S, G) hipLaunchKernelGGL((k_sift<S, G>), dim3(nsift), dim3(256), 0, c->stream, sa, c->pub)
        const bool with_gid = sa.gid != nullptr;
        if (merged_) {
            if (stream_out) { if (with_gid) LAUNCH_SIFT_PLANES(1, 1); else LAUNCH_SIFT_PLANES(1, 0); }
            else { if (with_gid) LAUNCH_SIFT_PLANES(0, 1); else LAUNCH_SIFT_PLANES(0, 0); }
        } else {
            if (stream_out) { if (with_gid) LAUNCH_SIFT(1, 1); else LAUNCH_SIFT(1, 0); }
            else { if (with_gid) LAUNCH_SIFT(0, 1); else LAUNCH_SIFT(0, 0); }
        }
#undef LAUNCH_SIFT_PLANES
#undef LAUNCH_SIFT
        return check_launch(c, "k_sift");
    };
    // A structure's FIRST pass (fork_lists): the list chain (~55 us at 100 k atoms) is longer than grid build + search (~43 us), and
    // the host needs ~5 us per launch.  So the main stream gets ALL its kernels first — grid, search, per-pair kernel — and the
    // second stream the lists and, behind them and behind the search (ring / amide masks), their evaluation as a kernel of its own
    // (k_planes); the two last kernels end the pass together (pass_end: expected = 2).  Later passes evaluate the resident lists
    // in the leading blocks of the per-pair launch.
    bool lists_forked = false;
    if (fork_lists) {
        if (c->pub.expected) c->pub.expected = 2;
        CHK(launch_search());
        HIPCHK(c, hipEventRecord(c->ev_lists, c->stream));              // (the masks are in place)
        CHK(launch_sift(false));
        // (enqueued behind the two big launches: a list kernel that starts beside the search takes CU slots the search counts on —
        // its blocks are all meant to be resident at once — and doubles it: 27 -> 54 us at 100 k atoms)
        CHK(join_upload_lists(c, true));           // (on the MAIN stream, before the two change places)
        HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_sel, 0));
        std::swap(c->stream, c->stream2);
        int rc = ensure_center_grids(c);
        if (rc == ARP_OK) rc = ensure_plane_lists(c);
        std::swap(c->stream, c->stream2);
        CHK(rc);
        lists_forked = true;
    }
    // ---- ring / amide loops (I:938-1382): evaluated from the static candidate lists by the leading blocks of the
    // last launch (ARP_PLANES_MODE=1: as a kernel of their own on the second stream, beside the search)
    int n0 = 0, n1 = 0, n2 = 0, n3 = 0;
    if (with_planes && c->nring + c->namide > 0) {
        CHK(ensure_center_grids(c));
        CHK(ensure_plane_lists(c));
        CHK(prepare_atom_plane(c, ap, n0, /*contact_grid=*/true));   // (argument blocks and bag buffers; no grid is walked)
        CHK(prepare_plane_plane(c, pp, n1));
        CHK(prepare_group_group(c, gg, n2));
        CHK(prepare_group_plane(c, gp, n3));
    }
    const bool have_planes = n0 + n1 + n2 + n3 > 0;
    // blocks for the list evaluation: one wave per 64 entries, from what the lists held last time (their capacity at first)
    if (have_planes) {
        long long entries = 0;
        for (int k = 0; k < 4; ++k) entries += std::min<long long>((long long)c->plist[k].cap, c->plist_known[k] >= 0 ? c->plist_known[k] + 64 : (long long)c->plist[k].cap / 4);
        // Few, fat blocks: they and the sift blocks of the same launch must ALL be resident from the start (the sift
        // blocks split their work statically: one that had to wait for a slot would finish that much later than the rest),
        // so the sift part gets num_cu * blocks-per-CU minus these.  Sixteen list chunks (of 64 entries) per wave are still
        // inside the time the sift blocks need (sweep in profiles/README.md); ring-heavy structures get more blocks, up to half the slots.
        // ... and fewer when the sift part itself is short: a list chunk takes about as long as a batch of 64 pairs (2.3 vs 2.5-3 us
        // in the block traces of round 2, profiles/README.md), and a sift wave of a small structure has only a batch or two.
        static const int cpw_max = std::max(1, env_int("ARP_PLANE_CPW", 16));
        const int64_t expect_pairs = (c->contacts_expected > 0) ? c->contacts_expected : (int64_t)c->n * 13;
        const double batches_per_wave = (double)expect_pairs / (64.0 * 4.0 * std::min<double>(c->num_cu * 4.0, std::max(1.0, expect_pairs / 256.0)));
        static const double chunk_factor = env_int("ARP_PLANE_CHUNK_FACTOR_X10", 10) / 10.0;
        const int chunks_per_wave = std::max(2, std::min(cpw_max, (int)(chunk_factor * batches_per_wave + 0.5)));
        np = (int)std::min<long long>(std::max<long long>((entries + 256 * chunks_per_wave - 1) / (256 * chunks_per_wave), 8), 2 * c->num_cu);
        np = (np + 7) & ~7;
    }
    static const int planes_mode = env_int("ARP_PLANES_MODE", 0);   // 0: one grid with the sift kernel; 1: second stream
    const bool merged = have_planes && c->n > 0 && !lists_forked && (planes_mode == 0 || c->external_stream);
    const bool planes_alone = (have_planes && !merged) || lists_forked;   // (forked: launched whatever it holds — it ends the pass with the per-pair kernel)
    hipStream_t st2 = c->external_stream ? c->stream : c->stream2;   // a caller-owned stream: everything in order on it
    if (planes_alone && !lists_forked && c->pub.expected) c->pub.expected = (c->n > 0) ? 2 : 1;
    CHK(launch_search());
    if (planes_alone) {
        if (st2 != c->stream) {      // the ring / amide masks are made by the search launch (or by k_group_masks before it)
            if (!lists_forked) HIPCHK(c, hipEventRecord(c->ev_lists, c->stream));
            HIPCHK(c, hipStreamWaitEvent(st2, c->ev_lists, 0));
        }
        Prof p(c, SLOT_PLANES, st2);
        hipLaunchKernelGGL(k_planes, dim3(std::max(np, 8)), dim3(256), 0, st2, ap, pp, gg, gp, plane_lists(c), c->d_ctr + ctr_dev(C_PLIST), c->pub, 15);
        CHK(check_launch(c, "k_planes"));
    }
    if (c->n > 0) {
        if (!sift_launched) CHK(launch_sift(merged));
    } else if (!planes_alone) {
        c->pub.expected = 0;   // nothing was launched that could publish: the caller falls back to k_publish_counters
    }
    if (planes_alone && st2 != c->stream) {   // join: whatever follows on the main stream sees the bags
        HIPCHK(c, hipEventRecord(c->ev_planes, c->stream2));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_planes, 0));
    }
    return ARP_OK;
}

// After read_counters(): publish contact results; returns true when the pair buffer overflowed.
bool finish_contacts(arp_ctx* c) {
    const u64 segcap = pair_segcap(c->pairs.cap);
    u64 np = 0, worst = 0;
    for (int k = 0; k < PAIR_SEGS; ++k) { np += c->h_ctr[C_SEG_PAIRS + k]; worst = std::max(worst, c->h_ctr[C_SEG_PAIRS + k]); }
    c->h_ctr[C_PAIRS] = np;
    c->h_ctr[C_SCRATCH0] = worst;
    for (int k = 0; k < PAIR_SEGS; ++k) c->seg_last[k] = c->h_ctr[C_SEG_PAIRS + k];
    if (worst > segcap) return true;
    c->n_contacts = (int64_t)np;
    c->contacts_expected = (int64_t)np;
    c->contacts_valid = true;
    c->contacts_sorted = false;
    c->stats[0] = (int64_t)c->h_ctr[C_CAND];
    c->stats[1] = (int64_t)c->h_ctr[C_ACC];
    c->stats[2] = (int64_t)np;
    if (c->cg_reused) c->h_ctr[C_BINNED] = (u64)c->cg_binned;         // (no grid build in this pass: the count its build reported)
    else if (c->cg_pending) { c->cg_binned = (int64_t)(uint32_t)c->h_ctr[C_BINNED]; c->cg_pending = false; }
    c->stats[3] = (int64_t)(uint32_t)c->h_ctr[C_BINNED];
    c->stats[4] = c->contact_cells;
    return false;
}

// ---- canonical order of the atom-atom bag (arp_sort.h) --------------------------------------------------------
// Layout of the sorted slab: the five columns one after the other, each on a 256-byte boundary; what follows them
// (sorted_extra bytes) is the caller's (the packed ring / amide bags of arp_fetch_packed).
inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
void sorted_layout(size_t k, size_t off[5], size_t* bytes, size_t first_col_entries) {      // (first column: k bgn ids, or N + 1 row offsets)
    off[0] = 0;
    off[1] = off[0] + al256(first_col_entries * sizeof(int32_t));
    off[2] = off[1] + al256(k * sizeof(int32_t));
    off[3] = off[2] + al256(k * sizeof(float));
    off[4] = off[3] + al256(k * sizeof(uint16_t));
    *bytes = off[4] + al256(k * sizeof(uint8_t));
}
int sort_contacts(arp_ctx* c, size_t extra_bytes = 0) {
    if (!c->contacts_valid) FAIL(c, ARP_E_ARG, "arp_atom_contacts_sort: no atom-contact results (call a launch first)");
    const size_t k = (size_t)c->n_contacts;
    const bool csr = c->packed_csr && !c->has_gid;      // (a shard's records carry global ids: no table of rows)
    const size_t rows = (size_t)std::max<int64_t>(c->n, 0);
    size_t off[5], bytes;
    sorted_layout(k, off, &bytes, csr ? rows + 1 : k);
    if (c->contacts_sorted && c->sorted_is_csr == csr && c->sorted_slab.cap >= bytes + extra_bytes) return ARP_OK;
    {   // sized from the capacity of the unsorted columns, so that the slab is allocated once per structure size
        size_t off_cap[5], bytes_cap;
        sorted_layout(std::max(k, c->out_i.cap), off_cap, &bytes_cap, std::max(std::max(k, c->out_i.cap), rows + 1));
        HIPCHK(c, c->sorted_slab.reserve(std::max(bytes_cap, bytes + extra_bytes)));
    }
    for (int q = 0; q < 5; ++q) c->srt_off[q] = off[q];
    c->srt_bytes = bytes;
    c->sorted_is_csr = csr;
    if (k == 0) {
        if (csr) HIPCHK(c, hipMemsetAsync(c->sorted_slab.p + off[0], 0, (rows + 1) * sizeof(int32_t), c->stream));
        c->contacts_sorted = true;
        return ARP_OK;
    }
    if (k >= ((size_t)1 << 31)) FAIL(c, ARP_E_CAPACITY, "arp_atom_contacts_sort: 2^31 records or more (the digit table's prefixes are 32-bit)");
    // significant bits of an atom index: packed ids of the resident structure, global ids on a shard
    const int64_t idmax = c->has_gid ? (c->gid_max >= 0 ? c->gid_max : ((int64_t)1 << 31) - 1) : std::max<int64_t>(c->n - 1, 1);
    int idbits = 1;
    while (((int64_t)1 << idbits) <= idmax) ++idbits;
    // radix passes over the bits of i only; the runs of equal i are ordered by j in one launch (k_sort_runs)
    const int passes = (idbits + SORT_MAX_BITS - 1) / SORT_MAX_BITS;
    const size_t cap = std::max(k, c->out_i.cap);
    HIPCHK(c, c->sort_key[0].reserve(cap)); HIPCHK(c, c->sort_val[0].reserve(cap));
    if (passes > 1) { HIPCHK(c, c->sort_key[1].reserve(cap)); HIPCHK(c, c->sort_val[1].reserve(cap)); }
    const long long tiles = ((long long)k + SORT_TILE - 1) / SORT_TILE;
    if (tiles > ((long long)1 << 30) / SORT_BINS) FAIL(c, ARP_E_CAPACITY, "arp_atom_contacts_sort: too many records for the digit table");
    const int tstride = (int)((tiles + 3) & ~3ll);
    HIPCHK(c, c->sort_table.reserve((size_t)SORT_BINS * (size_t)tstride));
    HIPCHK(c, c->sort_total.reserve(SORT_BINS));
    SortArgs A{};
    A.ci = c->out_i.p; A.cj = c->out_j.p;
    A.d_in = c->out_d.p; A.s_in = c->out_s.p; A.ct_in = c->out_ct.p;
    uint8_t* slab = c->sorted_slab.p;
    A.i_out = (int*)(slab + off[0]); A.j_out = (int*)(slab + off[1]); A.d_out = (float*)(slab + off[2]);
    A.s_out = (uint16_t*)(slab + off[3]); A.ct_out = slab + off[4];
    A.n = (long long)k;
    A.T = (int)tiles;
    A.tstride = tstride;
    A.jbits = idbits;
    A.table = c->sort_table.p;
    A.total = c->sort_total.p;
    A.row_out = csr ? A.i_out : nullptr;
    A.nrows = (int)rows;
    // a small bag: one block groups the records by bgn atom in one launch (k_sort_small)
    static const int small_mode = env_int("ARP_SORT_SMALL", 1);
    const bool small = small_mode && k <= (size_t)SORT_SMALL_MAX_RECORDS && idmax + 1 <= (int64_t)SORT_SMALL_MAX_BINS;
    if (small) {
        const int nbin = (int)idmax + 1;
        A.key_out = c->sort_key[0].p;
        A.val_out = c->sort_val[0].p;
        static const int small_blocks = std::max(1, std::min(64, env_int("ARP_SORT_SMALL_BLOCKS", SORT_SMALL_BLOCKS)));
        hipLaunchKernelGGL(k_sort_small, dim3(std::min(small_blocks, nbin)), dim3(SORT_SMALL_THREADS), 0, c->stream, A, nbin);
        A.key_in = c->sort_key[0].p;
        A.val_in = c->sort_val[0].p;
        hipLaunchKernelGGL(k_sort_runs, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, c->stream, A);
        CHK(check_launch(c, "k_sort_small"));
        c->contacts_sorted = true;
        return ARP_OK;
    }
    int shift = idbits;       // (key = i << idbits | j)
    for (int ps = 0; ps < passes; ++ps) {
        A.first = ps == 0; A.last = 0;
        A.shift = shift;
        A.bits = idbits / passes + (ps < idbits % passes ? 1 : 0);
        shift += A.bits;
        A.key_in = ps > 0 ? c->sort_key[(ps - 1) & 1].p : nullptr;
        A.val_in = ps > 0 ? c->sort_val[(ps - 1) & 1].p : nullptr;
        A.key_out = c->sort_key[ps & 1].p;
        A.val_out = c->sort_val[ps & 1].p;
        hipLaunchKernelGGL(k_sort_hist, dim3(A.T), dim3(SORT_THREADS), 0, c->stream, A);
        hipLaunchKernelGGL(k_sort_scan, dim3(1 << A.bits), dim3(SORT_THREADS), 0, c->stream, A);
        hipLaunchKernelGGL(k_sort_scatter, dim3(A.T), dim3(SORT_THREADS), 0, c->stream, A);
    }
    A.key_in = c->sort_key[(passes - 1) & 1].p;
    A.val_in = c->sort_val[(passes - 1) & 1].p;
    hipLaunchKernelGGL(k_sort_runs, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, c->stream, A);
    CHK(check_launch(c, "k_sort_scatter"));
    c->contacts_sorted = true;
    return ARP_OK;
}
// static candidate lists: entry counts travel with the counters of the pass; a list that was too small is re-sized and
// rebuilt before the pass is repeated
int finish_plane_lists(arp_ctx* c, bool grow) {
    if (!c->plist_count.p || c->nring + c->namide == 0) return 0;
    int again = 0;
    for (int k = 0; k < 4; ++k) {
        const u64 cnt = c->h_ctr[C_PLIST + k];
        c->plist_known[k] = (long long)cnt;
        if (cnt > (u64)c->plist[k].cap) {
            again = 1;
            if (grow) {
                c->plist[k].release();
                HIPCHK(c, c->plist[k].reserve((size_t)cnt + (size_t)cnt / 8 + 64));
                c->lists_dirty = true;
            }
        }
    }
    return again;
}
bool finish_bag(arp_ctx* c, Bag& b, int slot) {
    const u64 k = c->h_ctr[slot];
    if (k > b.cap) return true;
    b.count = (int64_t)k;
    b.valid = true;
    ++b.version;
    return false;
}
int grow_pairs(arp_ctx* c) {
    const size_t need = ((size_t)c->h_ctr[C_SCRATCH0] + (size_t)c->h_ctr[C_SCRATCH0] / 8 + 64) * PAIR_SEGS;
    c->pairs.release(); c->out_i.release(); c->out_j.release(); c->out_d.release(); c->out_s.release(); c->out_ct.release();
    HIPCHK(c, c->pairs.reserve(need));
    return ARP_OK;
}
int grow_bag(arp_ctx* c, Bag& b, int slot, bool d, bool f) {
    const size_t need = (size_t)c->h_ctr[slot];
    b.release();
    return bag_reserve(c, b, need, d, f);
}
int device_error(arp_ctx* c) {
    if ((int)(uint32_t)c->h_ctr[C_ERR] == ARP_E_XBOND_NBR)
        FAIL(c, ARP_E_XBOND_NBR, "xbond donor without a single-bond heavy neighbour (reference: AttributeError at utils.py:173)");
    if ((int)(uint32_t)c->h_ctr[C_ERR] == ARP_E_HIP)
        FAIL(c, ARP_E_HIP, "the grid's prefix scan timed out waiting for another tile (device shared or partly masked?): results of this pass are invalid; "
                           "ARP_CHAINED_SCAN=0 selects the two-launch scan");
    return ARP_OK;
}

// One ring / amide loop alone (arp_plane_plane_launch ...) from the static candidate lists of the structure, like the loops of a
// whole pass: the entries of the list in chunks of 64 over the waves of ONE launch (k_planes with the other three kinds masked
// out) — a launch of >= 1024 waves that each take a chunk or two, instead of one wave per ring / amide walking its 27-cell
// stencil as a chain of dependent loads (the grid walks of arp_planes.h, which the lists are built by: 10 k rings took 22 - 37 us).
// kind: 0 atom-plane, 1 plane-plane, 2 group-group, 3 group-plane.
int enqueue_bag_from_lists(arp_ctx* c, int kind) {
    AtomPlaneArgs ap{};
    PlanePlaneArgs pp{};
    GroupGroupArgs gg{};
    GroupPlaneArgs gp{};
    int nb = 0;
    if (c->n > 0 && kind == 0) CHK(ensure_static(c, c->last_cutoff));      // (the atoms' records by local id: st_xyzm)
    CHK(ensure_center_grids(c));
    CHK(ensure_plane_lists(c));
    if (kind == 0) CHK(prepare_atom_plane(c, ap, nb, /*contact_grid=*/true));
    else if (kind == 1) CHK(prepare_plane_plane(c, pp, nb));
    else if (kind == 2) CHK(prepare_group_group(c, gg, nb));
    else CHK(prepare_group_plane(c, gp, nb));
    if (!nb) return ARP_OK;
    const long long entries = std::min<long long>((long long)c->plist[kind].cap, c->plist_known[kind] >= 0 ? c->plist_known[kind] + 64 : (long long)c->plist[kind].cap);
    // one chunk of 64 entries per wave while the chip has room (8 waves per SIMD), more beyond
    const int blocks = (int)std::min<long long>(std::max<long long>((entries + 255) / 256, 8), (long long)c->num_cu * 8);
    Prof p(c, SLOT_PLANES, c->stream);
    hipLaunchKernelGGL(k_planes, dim3(blocks), dim3(256), 0, c->stream, ap, pp, gg, gp, plane_lists(c), c->d_ctr + ctr_dev(C_PLIST), PublishArgs{nullptr, nullptr, 0, 0}, 1 << kind);
    return check_launch(c, "k_planes (one list)");
}
int bag_launch_lists(arp_ctx* c, Bag& b, int slot, bool d, bool f, int kind, int64_t* count) {
    HIPCHK(c, hipSetDevice(c->device));
    CHK(ensure_default_selection(c));
    for (int attempt = 0;; ++attempt) {
        CHK(enqueue_bag_from_lists(c, kind));
        CHK(read_counters(c));
        collect_events(c);
        const bool list_small = finish_plane_lists(c, /*grow=*/true) != 0;      // (a list that was too small is re-sized and rebuilt by the next attempt)
        const bool bag_small = finish_bag(c, b, slot);
        if (!list_small && !bag_small) break;
        if (attempt == 3) FAIL(c, ARP_E_CAPACITY, "result bag could not be sized");
        if (bag_small) CHK(grow_bag(c, b, slot, d, f));
    }
    if (count) *count = b.count;
    return ARP_OK;
}

template <class F>
int bag_launch(arp_ctx* c, Bag& b, int slot, bool d, bool f, F enqueue, int64_t* count) {
    HIPCHK(c, hipSetDevice(c->device));
    CHK(ensure_default_selection(c));
    for (int attempt = 0;; ++attempt) {
        CHK(enqueue(c, c->stream));
        CHK(read_counters(c));
        collect_events(c);
        if (!finish_bag(c, b, slot)) break;
        if (attempt == 2) FAIL(c, ARP_E_CAPACITY, "result bag could not be sized");
        CHK(grow_bag(c, b, slot, d, f));
    }
    if (count) *count = b.count;
    return ARP_OK;
}

}  // namespace

// =====================================================================================
extern "C" {

const char* arp_version(void) { return "arpeggio_hip 0.2.0 (gfx950)"; }

int arp_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int arp_device_synchronize(arp_ctx* c) {
    if (!c) return ARP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    return ARP_OK;
}

int arp_create(int device, arp_ctx** out) {
    if (!out) return ARP_E_ARG;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        set_create_error(std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
        return ARP_E_HIP;
    }
    if (device < 0 || device >= ndev) {
        set_create_error("device ordinal out of range");
        return ARP_E_ARG;
    }
    e = hipSetDevice(device);
    if (e != hipSuccess) { set_create_error(hipGetErrorString(e)); return ARP_E_HIP; }
    arp_ctx* c = new (std::nothrow) arp_ctx();
    if (!c) return ARP_E_NOMEM;
    c->device = device;
    // the fused scan + scatter keeps the whole start table in LDS: up to 144 KB of the CU's 160 KB
    (void)hipFuncSetAttribute((const void*)k_scan_scatter_atoms<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 16384);
    (void)hipFuncSetAttribute((const void*)k_scan_scatter_atoms<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 16384);
    (void)hipFuncSetAttribute((const void*)k_scan_scatter_atoms<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 16384);
    (void)hipFuncSetAttribute((const void*)k_scan_scatter_atoms<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 7 * 16384);
    (void)hipFuncSetAttribute((const void*)k_scan_scatter_atoms<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384);
    (void)hipFuncSetAttribute((const void*)k_scan_scatter_atoms<9>, hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 16384);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cu = prop.multiProcessorCount;
    {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_search<MODE_CONTACTS>, 64 * SEARCH_WAVES, 0) == hipSuccess && per_cu > 0)
            c->search_resident = per_cu * c->num_cu;
        per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_sift_planes<0, 0>, 256, 0) == hipSuccess && per_cu > 0)
            c->sift_per_cu = per_cu;
    }
    e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    c->stream = c->own_stream;
    if (e == hipSuccess) {   // the second stream (ring / amide kernel) yields to the main one
        int lo_prio = 0, hi_prio = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio);
        e = hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, lo_prio);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_sel, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_planes, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_lists, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_upload, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_uplists, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_ctr, sizeof(u64) * C_DEV_WORDS);
    if (e == hipSuccess) e = hipMemset(c->d_ctr, 0, sizeof(u64) * C_DEV_WORDS);
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_ctr_pinned, sizeof(u64) * (C_COUNT + 1), hipHostMallocDefault);
    if (e == hipSuccess) memset(c->h_ctr_pinned, 0, sizeof(u64) * (C_COUNT + 1));
    if (e != hipSuccess) {
        set_create_error(hipGetErrorString(e));
        delete c;
        return ARP_E_HIP;
    }
    {   // k_sift deals its blocks to the pair-list segments by (XCD, blockIdx / 8): exact only if consecutive blocks of a launch go
        // round-robin over eight XCDs (up to a rotation).  Looked at once; otherwise the blocks are dealt out by index (seg_by_block).
        int* probe = nullptr;
        int h_probe[64];
        bool rr = false;
        if (hipMalloc((void**)&probe, sizeof(h_probe)) == hipSuccess) {
            hipLaunchKernelGGL(k_xcc_probe, dim3(64), dim3(64), 0, c->stream, probe);
            if (hipMemcpyAsync(h_probe, probe, sizeof(h_probe), hipMemcpyDeviceToHost, c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess) {
                rr = true;
                for (int b = 0; b < 64; ++b) rr = rr && ((h_probe[b] - h_probe[0] - b) & 7) == 0;
            }
            (void)hipFree(probe);
        }
        (void)hipGetLastError();
        static const int force_by_block = env_int("ARP_SIFT_SEG_BY_BLOCK", 0);
        c->xcd_round_robin = rr && !force_by_block;
    }
    *out = c;
    return ARP_OK;
}

void arp_destroy(arp_ctx* c) {
    if (c && c->comm) { (void)rccl().CommDestroy(c->comm); c->comm = nullptr; }
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    for (auto& e : c->ev_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    c->xyz.release(); c->rad.release(); c->tmask.release(); c->flags.release(); c->res_id.release();
    c->res_prev.release(); c->res_next.release(); c->res_flags.release(); c->bond_off.release();
    c->bond_idx.release(); c->h_off.release(); c->h_xyz_d.release(); c->sb.release(); c->gid.release();
    c->home.release(); c->sel.release(); c->plus.release(); c->res_sel.release(); c->res_plus.release();
    c->ring_c.release(); c->ring_n.release(); c->ring_res.release(); c->ring_sel.release(); c->ring_plus.release();
    c->am_c.release(); c->am_n.release(); c->am_res.release(); c->am_sel.release(); c->am_plus.release();
    c->s_xyzm.release(); c->s_aux.release(); c->s_qa.release(); c->tmp_i32.release(); c->sel_list.release(); c->st_qa.release(); c->rad_idx.release(); c->rad_tab.release(); c->st_aux.release(); c->st_xyzm.release();
    c->upload_bad.release(); c->sb_tile.release(); c->sb_cw.release(); c->sb_cwp.release(); c->sb_sums.release();
    c->sp_xyzm.release(); c->sp_aux.release(); c->sp_qa.release(); c->st_h.release(); c->sp_h.release(); c->s_h.release(); c->sp_cnt.release(); c->sp_cr.release();
    c->atom_grid.release(); c->all_grid.release(); c->a_xyzm.release(); c->a_aux.release(); c->ring_grid.release(); c->amide_grid.release(); c->tmp_u8.release();
    c->pairs.release(); c->out_i.release(); c->out_j.release(); c->out_d.release(); c->out_s.release(); c->out_ct.release();
    c->bag_ap.release(); c->bag_pp.release(); c->bag_gg.release(); c->bag_gp.release();
    c->bag_pack.release(); c->bag_perm.release();
    c->sort_key[0].release(); c->sort_key[1].release(); c->sort_val[0].release(); c->sort_val[1].release();
    c->sort_table.release(); c->sort_total.release(); c->sorted_slab.release();
    if (c->bag_stage) (void)hipHostFree(c->bag_stage);
    c->res_tag.release(); c->blob_sb_nbr.release(); c->blob_dev.release(); c->longest_bond.release();
    c->rec_home.release(); c->rec_face[0].release(); c->rec_face[1].release(); c->sh_scan.release(); c->sh_src.release();
    c->origin.release(); c->ring_origin.release(); c->am_origin.release(); c->sh_sel.release();
    for (auto& l : c->plist) l.release();
    c->plist_count.release();   // (views into the blob were released above: no-ops)
    c->ring_home.release(); c->am_home.release(); c->ring_gid.release(); c->am_gid.release();
    if (c->h_ctr_pinned) (void)hipHostFree(c->h_ctr_pinned);
    if (c->d_ctr) (void)hipFree(c->d_ctr);
    if (c->ev_sel) (void)hipEventDestroy(c->ev_sel);
    if (c->ev_planes) (void)hipEventDestroy(c->ev_planes);
    if (c->ev_lists) (void)hipEventDestroy(c->ev_lists);
    if (c->ev_upload) (void)hipEventDestroy(c->ev_upload);
    if (c->ev_uplists) (void)hipEventDestroy(c->ev_uplists);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

const char* arp_last_error(arp_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

// ---- inputs --------------------------------------------------------------------------
int arp_set_atoms(arp_ctx* c, int64_t n, const float* xyz, const double* vdw, const double* cov, const uint16_t* type_mask,
                  const uint16_t* flags, const int32_t* res_id) {
    if (!c) return ARP_E_ARG;
    CHK(join_upload_lists(c));
    if (n < 0 || n > 0x7FFFFFF0LL) FAIL(c, ARP_E_ARG, "arp_set_atoms: n out of range");
    if (n > 0 && (!xyz || !vdw || !cov || !type_mask || !flags || !res_id)) FAIL(c, ARP_E_ARG, "arp_set_atoms: null input");
    // the grids are sized from the bounding box: a NaN / inf coordinate has no cell
    if (!all_finite(xyz, 3 * n)) FAIL(c, ARP_E_ARG, "arp_set_atoms: non-finite coordinate");
    c->blob_bytes = 0;           // the resident structure no longer is the blob that was uploaded / assembled
    c->shard_resident = false;
    if (!all_finite(vdw, n) || !all_finite(cov, n)) FAIL(c, ARP_E_ARG, "arp_set_atoms: non-finite radius");
    int64_t max_res = -1;
    for (int64_t i = 0; i < n; ++i) {
        if (res_id[i] < 0) FAIL(c, ARP_E_ARG, "arp_set_atoms: negative residue index");
        max_res = std::max<int64_t>(max_res, res_id[i]);
    }
    c->max_res_id = max_res;
    HIPCHK(c, hipSetDevice(c->device));
    // uploads below are enqueued together; whatever the exit path, they are complete before the staging vectors die
    struct SyncOnExit { arp_ctx* c; ~SyncOnExit() { (void)hipStreamSynchronize(c->stream); } } sync_on_exit{c};
    c->static_dirty = true; c->lists_from_upload = false;
    c->n = n;
    c->h_xyz.assign(xyz, xyz + 3 * n);
    host_bbox(xyz, n, c->lo, c->hi);
    std::vector<float4> x4((size_t)n);
    std::vector<double2> r2((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        x4[i] = make_float4(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], 0.f);
        r2[i] = make_double2(vdw[i], cov[i]);
    }
    CHK(upload_async(c, c->xyz, x4.data(), (size_t)n));
    CHK(upload_async(c, c->rad, r2.data(), (size_t)n));
    {   // dictionary of the distinct {vdw, cov} pairs (element values: a handful per structure), compared bit for bit
        std::vector<double2> tab((size_t)RAD_TABLE, make_double2(0.0, 0.0));
        std::vector<uint16_t> idx((size_t)std::max<int64_t>(n, 1), (uint16_t)RAD_NONE);
        std::vector<std::pair<uint64_t, uint64_t>> keys;   // bit patterns, in table order
        int last = 0;
        for (int64_t i = 0; i < n; ++i) {
            uint64_t kv, kc;
            memcpy(&kv, &vdw[i], 8);
            memcpy(&kc, &cov[i], 8);
            int hit = -1;
            if (!keys.empty() && keys[(size_t)last].first == kv && keys[(size_t)last].second == kc) hit = last;
            for (size_t k = 0; hit < 0 && k < keys.size(); ++k)
                if (keys[k].first == kv && keys[k].second == kc) hit = (int)k;
            if (hit < 0) {
                if ((int)keys.size() >= RAD_TABLE) continue;   // stays RAD_NONE: k_sift fetches the uploaded radii
                hit = (int)keys.size();
                keys.emplace_back(kv, kc);
                tab[(size_t)hit] = make_double2(vdw[i], cov[i]);
            }
            last = hit;
            idx[(size_t)i] = (uint16_t)hit;
        }
        CHK(upload_async(c, c->rad_idx, idx.data(), (size_t)std::max<int64_t>(n, 1)));
        CHK(upload_async(c, c->rad_tab, tab.data(), (size_t)RAD_TABLE));
        CHK(upload_done(c));   // tab / idx go out of scope here
    }
    CHK(upload_async(c, c->tmask, type_mask, (size_t)n));
    CHK(upload_async(c, c->flags, flags, (size_t)n));
    CHK(upload_async(c, c->res_id, res_id, (size_t)n));
    // defaults for the optional per-atom inputs: no bonds, no hydrogens, no neighbours
    std::vector<int> zeros((size_t)n + 1, 0);
    CHK(upload_async(c, c->bond_off, zeros.data(), (size_t)n + 1));
    CHK(upload_async(c, c->h_off, zeros.data(), (size_t)n + 1));
    HIPCHK(c, c->bond_idx.reserve(1));
    HIPCHK(c, c->h_xyz_d.reserve(3));
    std::vector<float4> sb0((size_t)n, make_float4(0, 0, 0, 0));
    CHK(upload_async(c, c->sb, sb0.data(), (size_t)n));
    CHK(upload_done(c));   // (the staging vectors above live until here)
    c->has_gid = c->has_home = false;
    c->gid_max = -1;
    batch_reset(c);
    c->sel_made = false;
    c->sel_uploaded = false;   // a new structure starts with the default selection: everything (I:1395)
    c->sel_prefilled = false;
    c->nsel = -1;
    c->sel_all = false;
    c->whole_structure = false;
    c->contacts_valid = false;
    c->atom_grid.valid = false;
    c->all_grid_current = false;
    return ARP_OK;
}

int arp_set_residues(arp_ctx* c, int64_t nres, const uint8_t* res_flags, const int32_t* prev, const int32_t* next) {
    if (!c) return ARP_E_ARG;
    if (nres < 0 || (nres > 0 && (!res_flags || !prev || !next))) FAIL(c, ARP_E_ARG, "arp_set_residues: bad input");
    if (nres <= c->max_res_id) FAIL(c, ARP_E_ARG, "arp_set_residues: an atom refers to a residue beyond the table");
    HIPCHK(c, hipSetDevice(c->device));
    c->static_dirty = true; c->lists_from_upload = false;
    c->nres = nres;
    CHK(upload(c, c->res_flags, res_flags, (size_t)nres));
    CHK(upload(c, c->res_prev, prev, (size_t)nres));
    CHK(upload(c, c->res_next, next, (size_t)nres));
    c->has_res = true;
    c->atom_grid.valid = false;
    c->all_grid_current = false;
    c->sel_made = false;
    c->contacts_valid = false;
    return ARP_OK;
}

int arp_set_bonds(arp_ctx* c, const int32_t* bond_off, const int32_t* bond_idx) {
    if (!c || !bond_off) return ARP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    c->static_dirty = true; c->lists_from_upload = false;
    if (!csr_ok(bond_off, c->n)) FAIL(c, ARP_E_ARG, "arp_set_bonds: offsets must start at 0 and never decrease");
    const int64_t m = bond_off[c->n];
    if (m < 0 || (m > 0 && !bond_idx)) FAIL(c, ARP_E_ARG, "arp_set_bonds: bad CSR");
    CHK(upload(c, c->bond_off, bond_off, (size_t)c->n + 1));
    CHK(upload(c, c->bond_idx, bond_idx, (size_t)m));
    c->contacts_valid = false;
    return ARP_OK;
}

int arp_set_hydrogens(arp_ctx* c, const int32_t* h_off, const double* h_xyz) {
    if (!c || !h_off) return ARP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    c->static_dirty = true; c->lists_from_upload = false;
    if (!csr_ok(h_off, c->n)) FAIL(c, ARP_E_ARG, "arp_set_hydrogens: offsets must start at 0 and never decrease");
    const int64_t m = h_off[c->n];
    if (m < 0 || (m > 0 && !h_xyz)) FAIL(c, ARP_E_ARG, "arp_set_hydrogens: bad CSR");
    CHK(upload(c, c->h_off, h_off, (size_t)c->n + 1));
    CHK(upload(c, c->h_xyz_d, h_xyz, (size_t)m * 3));
    c->contacts_valid = false;
    return ARP_OK;
}

int arp_set_single_bond_neighbours(arp_ctx* c, const int32_t* sb_nbr) {
    if (!c || (c->n > 0 && !sb_nbr)) return ARP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    c->static_dirty = true; c->lists_from_upload = false;
    for (int64_t i = 0; i < c->n; ++i)
        if (sb_nbr[i] < -1 || sb_nbr[i] >= c->n) FAIL(c, ARP_E_ARG, "arp_set_single_bond_neighbours: index out of range");
    // the neighbour's coordinates are gathered on the device from the uploaded atoms (x, y, z, 1) / (0, 0, 0, 0)
    CHK(upload(c, c->tmp_i32, sb_nbr, (size_t)c->n));
    HIPCHK(c, c->sb.reserve((size_t)std::max<int64_t>(c->n, 1)));
    if (c->n > 0) {
        hipLaunchKernelGGL(k_gather_neighbours, dim3(nblocks(c->n, 256)), dim3(256), 0, c->stream, (int)c->n, c->tmp_i32.p, c->xyz.p, c->sb.p);
        CHK(check_launch(c, "k_gather_neighbours"));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    c->contacts_valid = false;
    return ARP_OK;
}

int arp_set_rings(arp_ctx* c, int64_t nring, const double* center, const double* normal, const int32_t* ring_res) {
    if (!c) return ARP_E_ARG;
    CHK(join_upload_lists(c));
    if (nring < 0 || (nring > 0 && (!center || !normal || !ring_res))) FAIL(c, ARP_E_ARG, "arp_set_rings: bad input");
    if (!all_finite(center, 3 * nring)) FAIL(c, ARP_E_ARG, "arp_set_rings: non-finite ring centre");   // (normals may be NaN: class '')
    c->max_ring_res = -1;
    for (int64_t i = 0; i < nring; ++i) {
        if (ring_res[i] < -1) FAIL(c, ARP_E_ARG, "arp_set_rings: residue index below -1");
        c->max_ring_res = std::max<int64_t>(c->max_ring_res, ring_res[i]);
    }
    HIPCHK(c, hipSetDevice(c->device));
    batch_reset(c);      // (the partition of a batch was declared for the arrays that were resident then: arp_set_batch again after this call)
    c->static_dirty = true; c->lists_from_upload = false;
    c->nring = nring;
    host_bbox_d(center, nring, c->ring_lo, c->ring_hi);
    CHK(upload(c, c->ring_c, center, (size_t)nring * 3));
    CHK(upload(c, c->ring_n, normal, (size_t)nring * 3));
    CHK(upload(c, c->ring_res, ring_res, (size_t)nring));
    HIPCHK(c, c->ring_sel.reserve((size_t)std::max<int64_t>(nring, 1)));
    HIPCHK(c, c->ring_plus.reserve((size_t)std::max<int64_t>(nring, 1)));
    c->ring_grid.valid = false;
    c->sel_made = false;
    c->has_group_owner = false;
    return ARP_OK;
}

int arp_set_amides(arp_ctx* c, int64_t namide, const float* center, const float* normal, const int32_t* amide_res) {
    if (!c) return ARP_E_ARG;
    CHK(join_upload_lists(c));
    if (namide < 0 || (namide > 0 && (!center || !normal || !amide_res))) FAIL(c, ARP_E_ARG, "arp_set_amides: bad input");
    if (!all_finite(center, 3 * namide)) FAIL(c, ARP_E_ARG, "arp_set_amides: non-finite amide centre");
    c->max_amide_res = -1;
    for (int64_t i = 0; i < namide; ++i) {
        if (amide_res[i] < -1) FAIL(c, ARP_E_ARG, "arp_set_amides: residue index below -1");
        c->max_amide_res = std::max<int64_t>(c->max_amide_res, amide_res[i]);
    }
    HIPCHK(c, hipSetDevice(c->device));
    batch_reset(c);      // (the partition of a batch was declared for the arrays that were resident then: arp_set_batch again after this call)
    c->static_dirty = true; c->lists_from_upload = false;
    c->namide = namide;
    host_bbox(center, namide, c->am_lo, c->am_hi);
    CHK(upload(c, c->am_c, center, (size_t)namide * 3));
    CHK(upload(c, c->am_n, normal, (size_t)namide * 3));
    CHK(upload(c, c->am_res, amide_res, (size_t)namide));
    HIPCHK(c, c->am_sel.reserve((size_t)std::max<int64_t>(namide, 1)));
    HIPCHK(c, c->am_plus.reserve((size_t)std::max<int64_t>(namide, 1)));
    c->amide_grid.valid = false;
    c->sel_made = false;
    c->has_group_owner = false;
    return ARP_OK;
}

// ---- one-blob upload -----------------------------------------------------------------------------------------------
namespace {
struct BlobSizes { uint64_t esize[ARP_BLOB_ARRAYS]; uint64_t count[ARP_BLOB_ARRAYS]; };
bool blob_sizes(int64_t n, int64_t nres, int64_t nbond, int64_t nh, int64_t nring, int64_t namide, BlobSizes& z) {
    if (n < 0 || nres < 0 || nbond < 0 || nh < 0 || nring < 0 || namide < 0) return false;
    if (n > 0x7FFFFFF0LL || nres > 0x7FFFFFF0LL || nbond > 0x7FFFFFF0LL || nh > 0x7FFFFFF0LL / 3 || nring > 0x7FFFFFF0LL / 3 ||
        namide > 0x7FFFFFF0LL / 3)
        return false;
    const uint64_t N = (uint64_t)n, NR = (uint64_t)nres, R = (uint64_t)nring, A = (uint64_t)namide;
    const uint64_t es[ARP_BLOB_ARRAYS] = {4, 8, 2, 2, 4, 1, 4, 4, 4, 4, 4, 8, 4, 8, 8, 4, 4, 4, 4, 2, 8};
    const uint64_t ct[ARP_BLOB_ARRAYS] = {4 * N, 2 * N, N, N, N, NR, NR, NR, N + 1, (uint64_t)nbond, N + 1, 3 * (uint64_t)nh, N, 3 * R, 3 * R,
                                          R, 3 * A, 3 * A, A, N, 2 * RAD_TABLE};
    for (int k = 0; k < ARP_BLOB_ARRAYS; ++k) { z.esize[k] = es[k]; z.count[k] = ct[k]; }
    return true;
}
uint64_t align16(uint64_t v) { return (v + 15ull) & ~15ull; }
}  // namespace

uint64_t arp_blob_size(int64_t n, int64_t nres, int64_t nbond, int64_t nh, int64_t nring, int64_t namide) {
    BlobSizes z;
    if (!blob_sizes(n, nres, nbond, nh, nring, namide, z)) return 0;
    uint64_t off = align16(sizeof(arp_blob_header));
    for (int k = 0; k < ARP_BLOB_ARRAYS; ++k) off = align16(off + z.esize[k] * z.count[k]);
    return off;
}

int arp_blob_layout(void* blob, uint64_t bytes, int64_t n, int64_t nres, int64_t nbond, int64_t nh, int64_t nring, int64_t namide) {
    const uint64_t need = arp_blob_size(n, nres, nbond, nh, nring, namide);
    if (!blob || need == 0 || bytes < need) return ARP_E_ARG;
    BlobSizes z;
    blob_sizes(n, nres, nbond, nh, nring, namide, z);
    arp_blob_header h;
    memset(&h, 0, sizeof(h));
    h.magic = ARP_BLOB_MAGIC;
    h.bytes = need;
    h.n = n; h.nres = nres; h.nbond = nbond; h.nh = nh; h.nring = nring; h.namide = namide;
    uint64_t off = align16(sizeof(arp_blob_header));
    for (int k = 0; k < ARP_BLOB_ARRAYS; ++k) {
        h.off[k] = off;
        off = align16(off + z.esize[k] * z.count[k]);
    }
    memcpy(blob, &h, sizeof(h));
    return ARP_OK;
}

int arp_blob_fill(void* blob, uint64_t bytes, const float* xyz, const double* vdw, const double* cov, const uint16_t* type_mask,
                  const uint16_t* flags, const int32_t* res_id, const uint8_t* res_flags, const int32_t* res_prev,
                  const int32_t* res_next, const int32_t* bond_off, const int32_t* bond_idx, const int32_t* h_off,
                  const double* h_xyz, const int32_t* sb_nbr, const double* ring_center, const double* ring_normal,
                  const int32_t* ring_res, const float* amide_center, const float* amide_normal, const int32_t* amide_res) {
    if (!blob || bytes < sizeof(arp_blob_header)) return ARP_E_ARG;
    arp_blob_header h;
    memcpy(&h, blob, sizeof(h));
    if (h.magic != ARP_BLOB_MAGIC || h.bytes > bytes || h.bytes != arp_blob_size(h.n, h.nres, h.nbond, h.nh, h.nring, h.namide)) return ARP_E_ARG;
    const int64_t n = h.n;
    if ((n > 0 && (!xyz || !vdw || !cov || !type_mask || !flags || !res_id || !bond_off || !h_off || !sb_nbr)) ||
        (h.nres > 0 && (!res_flags || !res_prev || !res_next)) || (h.nbond > 0 && !bond_idx) || (h.nh > 0 && !h_xyz) ||
        (h.nring > 0 && (!ring_center || !ring_normal || !ring_res)) || (h.namide > 0 && (!amide_center || !amide_normal || !amide_res)))
        return ARP_E_ARG;
    uint8_t* const b = (uint8_t*)blob;
    auto at = [&](int k) { return b + h.off[k]; };
    float* x4 = (float*)at(0);
    double* r2 = (double*)at(1);
    for (int64_t i = 0; i < n; ++i) {
        x4[4 * i] = xyz[3 * i]; x4[4 * i + 1] = xyz[3 * i + 1]; x4[4 * i + 2] = xyz[3 * i + 2]; x4[4 * i + 3] = 0.0f;
        r2[2 * i] = vdw[i]; r2[2 * i + 1] = cov[i];
    }
    auto copy = [&](int k, const void* src, size_t nbytes) { if (nbytes) memcpy(at(k), src, nbytes); };
    copy(2, type_mask, (size_t)n * 2); copy(3, flags, (size_t)n * 2); copy(4, res_id, (size_t)n * 4);
    copy(5, res_flags, (size_t)h.nres); copy(6, res_prev, (size_t)h.nres * 4); copy(7, res_next, (size_t)h.nres * 4);
    if (n > 0) { copy(8, bond_off, ((size_t)n + 1) * 4); copy(10, h_off, ((size_t)n + 1) * 4); }
    else { const int32_t z = 0; copy(8, &z, 4); copy(10, &z, 4); }
    copy(9, bond_idx, (size_t)h.nbond * 4); copy(11, h_xyz, (size_t)h.nh * 24); copy(12, sb_nbr, (size_t)n * 4);
    copy(13, ring_center, (size_t)h.nring * 24); copy(14, ring_normal, (size_t)h.nring * 24); copy(15, ring_res, (size_t)h.nring * 4);
    copy(16, amide_center, (size_t)h.namide * 12); copy(17, amide_normal, (size_t)h.namide * 12); copy(18, amide_res, (size_t)h.namide * 4);
    // dictionary of the distinct {vdw, cov} pairs, compared bit for bit: a handful of element values in practice, so a
    // small open-addressing table keyed by the 128 bits; entries are numbered in ascending (vdw bits, cov bits) order
    uint16_t* ridx = (uint16_t*)at(19);
    double* tab = (double*)at(20);
    memset(tab, 0, sizeof(double) * 2 * RAD_TABLE);
    struct Key { uint64_t a, b; int64_t count; int slot; };
    std::vector<Key> keys;
    std::vector<int> hash(4096, -1);
    std::vector<int> key_of((size_t)n);
    auto bits = [](double d) { uint64_t u; memcpy(&u, &d, 8); return u; };
    for (int64_t i = 0; i < n; ++i) {
        const uint64_t ka = bits(vdw[i]), kb = bits(cov[i]);
        size_t hpos = (size_t)((ka * 0x9E3779B97F4A7C15ull) ^ (kb * 0xC2B2AE3D27D4EB4Full)) >> 20;
        int found = -1;
        for (;;) {
            hpos &= hash.size() - 1;
            const int k = hash[hpos];
            if (k < 0) break;
            if (keys[(size_t)k].a == ka && keys[(size_t)k].b == kb) { found = k; break; }
            ++hpos;
        }
        if (found < 0) {
            if (keys.size() * 2 >= hash.size()) {   // grow and re-insert
                std::vector<int> bigger(hash.size() * 4, -1);
                for (size_t k = 0; k < keys.size(); ++k) {
                    size_t p = (size_t)((keys[k].a * 0x9E3779B97F4A7C15ull) ^ (keys[k].b * 0xC2B2AE3D27D4EB4Full)) >> 20;
                    for (;; ++p) { p &= bigger.size() - 1; if (bigger[p] < 0) { bigger[p] = (int)k; break; } }
                }
                hash.swap(bigger);
                hpos = (size_t)((ka * 0x9E3779B97F4A7C15ull) ^ (kb * 0xC2B2AE3D27D4EB4Full)) >> 20;
                for (;; ++hpos) { hpos &= hash.size() - 1; if (hash[hpos] < 0) break; }
            }
            found = (int)keys.size();
            keys.push_back(Key{ka, kb, 0, -1});
            hash[hpos] = found;
        }
        ++keys[(size_t)found].count;
        key_of[(size_t)i] = found;
    }
    std::vector<int> order(keys.size());
    for (size_t k = 0; k < keys.size(); ++k) order[k] = (int)k;
    std::sort(order.begin(), order.end(), [&](int p, int q) { return keys[(size_t)p].a != keys[(size_t)q].a ? keys[(size_t)p].a < keys[(size_t)q].a : keys[(size_t)p].b < keys[(size_t)q].b; });
    if (keys.size() <= (size_t)RAD_TABLE) {
        for (size_t r = 0; r < order.size(); ++r) keys[(size_t)order[r]].slot = (int)r;
        h.n_rad = (int64_t)keys.size();
    } else {   // the 256 most frequent pairs (ties: the smaller pair first), numbered by descending frequency
        std::stable_sort(order.begin(), order.end(), [&](int p, int q) { return keys[(size_t)p].count > keys[(size_t)q].count; });
        for (size_t r = 0; r < (size_t)RAD_TABLE; ++r) keys[(size_t)order[r]].slot = (int)r;
        h.n_rad = RAD_TABLE;
    }
    for (const Key& k : keys)
        if (k.slot >= 0) { memcpy(&tab[2 * k.slot], &k.a, 8); memcpy(&tab[2 * k.slot + 1], &k.b, 8); }
    for (int64_t i = 0; i < n; ++i) {
        const int slot = keys[(size_t)key_of[(size_t)i]].slot;
        ridx[i] = slot >= 0 ? (uint16_t)slot : (uint16_t)RAD_NONE;
    }
    // bounding boxes (0 for an empty set)
    auto box = [](auto* pts, int64_t cnt, double* lo, double* hi) {
        for (int k = 0; k < 3; ++k) { lo[k] = hi[k] = cnt > 0 ? (double)pts[k] : 0.0; }
        for (int64_t i = 1; i < cnt; ++i)
            for (int k = 0; k < 3; ++k) {
                const double v = (double)pts[3 * i + k];
                lo[k] = std::min(lo[k], v);
                hi[k] = std::max(hi[k], v);
            }
    };
    box(xyz, n, h.lo, h.hi);
    box(ring_center, h.nring, h.ring_lo, h.ring_hi);
    box(amide_center, h.namide, h.amide_lo, h.amide_hi);
    memcpy(blob, &h, sizeof(h));
    return ARP_OK;
}

namespace {
// header of a blob: counts, size and offsets as arp_blob_layout writes them, finite boxes
int check_blob_header(arp_ctx* c, const arp_blob_header& h, uint64_t bytes) {
    if (h.magic != ARP_BLOB_MAGIC) FAIL(c, ARP_E_ARG, "arp_set_blob: bad magic");
    BlobSizes z;
    if (!blob_sizes(h.n, h.nres, h.nbond, h.nh, h.nring, h.namide, z)) FAIL(c, ARP_E_ARG, "arp_set_blob: counts out of range");
    if (h.bytes != arp_blob_size(h.n, h.nres, h.nbond, h.nh, h.nring, h.namide) || h.bytes > bytes)
        FAIL(c, ARP_E_ARG, "arp_set_blob: size does not match the counts");
    {   // the offsets must be the ones arp_blob_layout writes
        uint64_t off = align16(sizeof(arp_blob_header));
        for (int k = 0; k < ARP_BLOB_ARRAYS; ++k) {
            if (h.off[k] != off) FAIL(c, ARP_E_ARG, "arp_set_blob: unexpected array offset");
            off = align16(off + z.esize[k] * z.count[k]);
        }
    }
    if (h.n_rad < 0 || h.n_rad > RAD_TABLE) FAIL(c, ARP_E_ARG, "arp_set_blob: n_rad out of range");
    if (h.n > 0 && h.nres <= 0) FAIL(c, ARP_E_ARG, "arp_set_blob: atoms without a residue table");
    for (int k = 0; k < 3; ++k) {
        const bool ok = std::isfinite(h.lo[k]) && std::isfinite(h.hi[k]) && h.lo[k] <= h.hi[k] && std::isfinite(h.ring_lo[k]) &&
                        std::isfinite(h.ring_hi[k]) && h.ring_lo[k] <= h.ring_hi[k] && std::isfinite(h.amide_lo[k]) &&
                        std::isfinite(h.amide_hi[k]) && h.amide_lo[k] <= h.amide_hi[k];
        if (!ok) FAIL(c, ARP_E_ARG, "arp_set_blob: bad bounding box");
    }
    return ARP_OK;
}

// the input arrays of the context become views into c->blob_dev (which holds, or is about to hold, a blob with header h)
void borrow_blob_views(arp_ctx* c, const arp_blob_header& h) {
    uint8_t* const d = c->blob_dev.p;
    const size_t n1 = (size_t)std::max<int64_t>(h.n, 1);
    c->xyz.borrow(d + h.off[0], n1); c->rad.borrow(d + h.off[1], n1); c->tmask.borrow(d + h.off[2], n1);
    c->flags.borrow(d + h.off[3], n1); c->res_id.borrow(d + h.off[4], n1);
    c->res_flags.borrow(d + h.off[5], (size_t)std::max<int64_t>(h.nres, 1)); c->res_prev.borrow(d + h.off[6], (size_t)std::max<int64_t>(h.nres, 1));
    c->res_next.borrow(d + h.off[7], (size_t)std::max<int64_t>(h.nres, 1));
    c->bond_off.borrow(d + h.off[8], (size_t)h.n + 1); c->bond_idx.borrow(d + h.off[9], (size_t)std::max<int64_t>(h.nbond, 1));
    c->h_off.borrow(d + h.off[10], (size_t)h.n + 1); c->h_xyz_d.borrow(d + h.off[11], (size_t)std::max<int64_t>(3 * h.nh, 3));
    c->blob_sb_nbr.borrow(d + h.off[12], n1);
    c->ring_c.borrow(d + h.off[13], (size_t)std::max<int64_t>(3 * h.nring, 1)); c->ring_n.borrow(d + h.off[14], (size_t)std::max<int64_t>(3 * h.nring, 1));
    c->ring_res.borrow(d + h.off[15], (size_t)std::max<int64_t>(h.nring, 1));
    c->am_c.borrow(d + h.off[16], (size_t)std::max<int64_t>(3 * h.namide, 1)); c->am_n.borrow(d + h.off[17], (size_t)std::max<int64_t>(3 * h.namide, 1));
    c->am_res.borrow(d + h.off[18], (size_t)std::max<int64_t>(h.namide, 1));
    c->rad_idx.borrow(d + h.off[19], n1); c->rad_tab.borrow(d + h.off[20], (size_t)RAD_TABLE);
    c->n = h.n; c->nres = h.nres; c->nring = h.nring; c->namide = h.namide;
    c->has_res = true;
    c->blob_nbond = h.nbond; c->blob_nh = h.nh; c->blob_nrad = h.n_rad;
    c->blob_bytes = h.bytes;
    c->max_res_id = h.nres - 1; c->max_ring_res = h.nres - 1; c->max_amide_res = h.nres - 1;   // (ranges verified on the device)
    for (int k = 0; k < 3; ++k) {
        c->lo[k] = h.lo[k]; c->hi[k] = h.hi[k];
        c->ring_lo[k] = h.ring_lo[k]; c->ring_hi[k] = h.ring_hi[k];
        c->am_lo[k] = h.amide_lo[k]; c->am_hi[k] = h.amide_hi[k];
    }
    c->h_xyz.clear();
}

// Device-side validation of the resident blob (what arp_set_atoms ... check on the host) + the bookkeeping of a new
// structure.  Waits for the stream.  `also` = further device error words OR-ed in (shard assembly), may be null.
int validate_resident_blob(arp_ctx* c, const arp_blob_header& h, const char* who, const int* also = nullptr, bool gather_sb = false, bool rad_from_table = false) {
    int* const d_err = (int*)(c->d_ctr + ctr_dev(C_ERR));
    const bool after_batch = c->batch_n > 0;      // (before the bookkeeping below resets it)
    // The verdict reaches the host the way the counters of a pass do: the last block of the kernel stores the counter block in
    // the pinned mirror and the host polls the completion word (pass_end) — no copy launch behind the kernel, no
    // hipStreamSynchronize (~15 us per structure).  That needs the whole block zero at the start (its tickets live there) and
    // leaves it zero.  With further device words to read (`also`) or a caller-owned stream: a copy and a stream wait.
    const bool polled = !also && !c->external_stream;
    const bool counters_were_zero = c->ctr_zero_ok;      // (the error word is the only counter touched here, and it ends as zero when all is well)
    if (polled) {
        if (!c->ctr_zero_ok) HIPCHK(c, hipMemsetAsync(c->d_ctr, 0, sizeof(u64) * C_DEV_WORDS, c->stream));
    } else {
        HIPCHK(c, hipMemsetAsync(d_err, 0, sizeof(u64), c->stream));
    }
    c->ctr_zero_ok = false;
    if (gather_sb) HIPCHK(c, c->sb.reserve((size_t)std::max<int64_t>(h.n, 1)));
    BlobCheck bc;
    bc.n = (int)h.n; bc.nres = (int)h.nres; bc.nbond = (int)h.nbond; bc.nh = (int)h.nh; bc.nring = (int)h.nring; bc.namide = (int)h.namide;
    bc.nrad = (int)h.n_rad;
    for (int k = 0; k < 3; ++k) {   // float32 coordinates against the double box: widen by one float ulp either way
        bc.lo[k] = std::nextafter((float)h.lo[k], -INFINITY);
        bc.hi[k] = std::nextafter((float)h.hi[k], INFINITY);
    }
    bc.xyz = c->xyz.p; bc.rad = c->rad.p; bc.rad_idx = c->rad_idx.p; bc.res_id = c->res_id.p; bc.res_prev = c->res_prev.p;
    bc.res_next = c->res_next.p; bc.bond_off = c->bond_off.p; bc.bond_idx = c->bond_idx.p; bc.h_off = c->h_off.p;
    bc.h_xyz = c->h_xyz_d.p; bc.sb_nbr = c->blob_sb_nbr.p; bc.ring_c = c->ring_c.p; bc.ring_res = c->ring_res.p;
    bc.am_c = c->am_c.p; bc.am_res = c->am_res.p; bc.err = d_err;
    bc.sb_out = gather_sb ? c->sb.p : nullptr;
    bc.rad_out = rad_from_table ? (double2*)c->rad.p : nullptr;
    bc.rad_tab = (const double2*)c->rad_tab.p;
    // The two fills the first pass over a new structure would otherwise begin with ride in this launch (each is ~4 us of
    // launch on the host and a gap on the device): the default selection and the cleared histogram of the static order.
    const size_t sel_words = ((size_t)std::max<int64_t>(h.n, 1) + 3) / 4;
    HIPCHK(c, c->sel.reserve(sel_words * 4));
    bc.fill_ones = (uint32_t*)c->sel.p; bc.fill_ones_n = (int)sel_words;
    const size_t zero_ints = std::min<size_t>(c->sp_cnt.cap & ~(size_t)3, (size_t)1 << 30);
    bc.fill_zero = (int4*)c->sp_cnt.p; bc.fill_zero_n = (int)(zero_ints / 4);
    const int64_t work = std::max({h.n, h.nbond, 3 * h.nh, h.nring, h.namide, h.nres, (int64_t)1});
    PublishArgs pub{c->d_ctr, c->h_ctr_pinned, 0, 0};
    if (polled) { pub.expected = 1; pub.seq = ++c->publish_seq; }
    // (what the second stream waits for below is the structure's copy, not its validation: the event goes in front of the kernel)
    static const int grids_with_upload = env_int("ARP_GRIDS_WITH_UPLOAD", 1);
    static const int upload_aside = env_int("ARP_UPLOAD_ASIDE", 1);
    // (not when the context's last structure was a batch: the next call is arp_set_batch again, which throws away whatever was made
    // for the concatenation as ONE structure — whose overlaid coordinates make for enormous candidate lists on top: a batch of 64
    // stand-ins took 3.2 instead of 1.4 ms end to end)
    const bool with_upload = grids_with_upload && polled && h.n > 0 && h.nring + h.namide > 0 && !after_batch;
    const bool on_second = with_upload && upload_aside && c->stream2 && c->ev_upload && c->ev_uplists;
    // single-bond neighbour coordinates, ring / amide masks, bookkeeping: as the classic setters leave them (before the launches
    // below, which read some of it: ownership flags, the batch, the static state)
    const size_t n1 = (size_t)std::max<int64_t>(h.n, 1);
    HIPCHK(c, c->sb.reserve(n1));
    HIPCHK(c, c->ring_sel.reserve((size_t)std::max<int64_t>(h.nring, 1))); HIPCHK(c, c->ring_plus.reserve((size_t)std::max<int64_t>(h.nring, 1)));
    HIPCHK(c, c->am_sel.reserve((size_t)std::max<int64_t>(h.namide, 1))); HIPCHK(c, c->am_plus.reserve((size_t)std::max<int64_t>(h.namide, 1)));
    c->static_dirty = true;
    c->lists_dirty = true;
    c->lists_from_upload = false;
    c->has_gid = c->has_home = c->has_group_owner = false;
    c->gid_max = -1;
    c->shard_resident = false;
    batch_reset(c);
    c->sel_made = false; c->sel_uploaded = false; c->nsel = -1; c->sel_all = false; c->whole_structure = false;
    c->sel_prefilled = true; c->sp_cnt_zeroed = zero_ints;      // (k_validate_blob's fills)
    c->contacts_valid = false;
    c->atom_grid.valid = false; c->all_grid_current = false;
    // a failed check also leaves the number of this upload in a word of its own (the error word goes back to zero when the verdict
    // is published): kernels enqueued ahead of the verdict look at it and leave (speculative static order, below)
    HIPCHK(c, c->upload_bad.reserve(1));
    if (!c->upload_bad_cleared) { HIPCHK(c, hipMemsetAsync(c->upload_bad.p, 0, sizeof(int), c->stream)); c->upload_bad_cleared = true; }
    bc.bad_seq = polled ? c->upload_bad.p : nullptr;
    bc.seq = (int)(pub.seq & 0x7FFFFFFF);
    if (on_second) HIPCHK(c, hipEventRecord(c->ev_upload, c->stream));
    hipLaunchKernelGGL(k_validate_blob, dim3(nblocks(work, 256, 2048)), dim3(256), 0, c->stream, bc, pub);
    CHK(check_launch(c, "k_validate_blob"));
    // The 6 A grids of the ring and amide centres and the candidate lists of the ring / amide loops depend on what was uploaded
    // only (centres, their boxes, the atoms as they came): they are built beside the validation kernel, on the SECOND stream, as
    // soon as the copy has landed, while the host waits for the verdict and turns round, and the first pass joins them in front of
    // its grid build (ARP_UPLOAD_ASIDE=0: behind the validation kernel on the main stream, where the static order of the first pass
    // then waits for them: 25 us of latency chains at 100 k atoms).  A structure that fails the validation has them thrown away
    // below: centres outside their box are clamped into it, nothing is written out of bounds.
    // A failure behind the validation kernel (an allocation, a launch): its verdict is still on its way and kernels of the second
    // stream may be reading the blob — wait for both streams, consume what was published, leave no structure resident.
    auto abandon = [&](int rc) -> int {
        const std::string why = c->err;
        if (c->stream2) (void)hipStreamSynchronize(c->stream2);
        if (polled) (void)collect_counters(c); else (void)hipStreamSynchronize(c->stream);
        c->err = why;
        c->uplists_pending = false;
        c->n = c->nres = c->nring = c->namide = 0;
        c->blob_bytes = 0;
        c->static_dirty = true; c->sp_radius = 0;
        c->lists_dirty = true; c->lists_from_upload = false;
        c->ring_grid.valid = false; c->amide_grid.valid = false;
        c->sel_prefilled = false; c->sp_cnt_zeroed = 0;
        c->ctr_zero_ok = false;
        return rc;
    };
    bool grids_made = false;
    if (with_upload) {
        c->ring_grid.valid = false; c->amide_grid.valid = false;
        c->lists_dirty = true;
        batch_reset(c);
        if (on_second) {
            HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_upload, 0));
            std::swap(c->stream, c->stream2);
        }
        int rc = ensure_center_grids(c);
        static const int lists_with_upload = env_int("ARP_LISTS_WITH_UPLOAD", 1);
        if (rc == ARP_OK && lists_with_upload) rc = ensure_plane_lists(c);      // (the four candidate lists need nothing else: see ar_enumerate)
        if (on_second) {
            std::swap(c->stream, c->stream2);
            if (rc == ARP_OK) {
                if (hipEventRecord(c->ev_uplists, c->stream2) != hipSuccess) { c->err = "hipEventRecord failed (upload lists)"; rc = ARP_E_HIP; }
                else c->uplists_pending = true;
            }
        }
        if (rc != ARP_OK) return abandon(rc);
        grids_made = true;
    }
    const bool ring_grid_made = grids_made && c->ring_grid.valid, amide_grid_made = grids_made && c->amide_grid.valid;
    const bool lists_made = grids_made && !c->lists_dirty;
    // The static columns of the structure and their spatial order — the first three launches of its first pass — go out NOW, for the
    // cell edge of the context's last pass: they run while the host waits for the verdict and turns round (13 us between the end of
    // the validation kernel and the first launch of the pass, at 100 k atoms), and leave at once when the check has failed
    // (upload_bad: the arrays they would index are what the check is about).  A pass with another cell edge re-orders the columns
    // as it does for any resident structure.
    static const int static_ahead = env_int("ARP_STATIC_WITH_UPLOAD", 1);
    if (static_ahead && polled && h.n > 0 && c->last_cutoff > 0 && !after_batch) {
        c->ahead_seq = bc.seq;
        const int rc = ensure_static(c, c->last_cutoff);
        c->ahead_seq = 0;
        if (rc != ARP_OK) return abandon(rc);
    }
    int h_err[2] = {0, 0};
    if (polled) {
        CHK(collect_counters(c));
        h_err[0] = (int)(uint32_t)c->h_ctr[C_ERR];
    } else {
        HIPCHK(c, hipMemcpyAsync(&h_err[0], d_err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        if (also) HIPCHK(c, hipMemcpyAsync(&h_err[1], also, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    c->lists_dirty = !lists_made;
    c->lists_from_upload = lists_made;
    const bool verdict_ok = h_err[0] == 0 && h_err[1] == 0;
    if (!verdict_ok) { c->static_dirty = true; c->sp_radius = 0; }      // (whatever was composed ahead of the verdict is void)
    c->ring_grid.valid = verdict_ok && ring_grid_made; c->amide_grid.valid = verdict_ok && amide_grid_made;
    c->bag_ap.valid = c->bag_pp.valid = c->bag_gg.valid = c->bag_gp.valid = false;
    if (h_err[0] != 0 || h_err[1] != 0) {
        c->n = c->nres = c->nring = c->namide = 0;   // nothing usable is resident
        c->blob_bytes = 0;
        c->err = std::string(who) + ": the structure failed validation (non-finite value, point outside its box, index out of range, "
                                    "offsets that are not a CSR, or an item that occurs twice)";
        return ARP_E_ARG;
    }
    c->ctr_zero_ok = polled ? true : counters_were_zero;      // (polled: the publishing block returned every counter to zero)
    return ARP_OK;
}
}  // namespace

int arp_set_blob(arp_ctx* c, const void* blob, uint64_t bytes) {
    if (!c || !blob || bytes < sizeof(arp_blob_header)) return ARP_E_ARG;
    CHK(join_upload_lists(c));
    arp_blob_header h;
    memcpy(&h, blob, sizeof(h));
    CHK(check_blob_header(c, h, bytes));
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, c->blob_dev.reserve((size_t)h.bytes));
    // The per-atom radii (16 of a structure's ~87 bytes per atom) are redundant when every atom's {vdw, cov} pair is in the
    // blob's table — always, for real structures: a handful of elements —: they stay on the host and the validation kernel
    // writes them on the device from the table.  (From 32 768 atoms on: below that the second copy costs more than the bytes.)
    bool rad_from_table = h.n >= 32768 && h.n_rad > 0 && h.n_rad <= RAD_TABLE;
    if (rad_from_table) {
        const uint16_t* ridx = (const uint16_t*)((const uint8_t*)blob + h.off[19]);
        unsigned any_none = 0;
        for (int64_t i = 0; i < h.n; ++i) any_none |= (unsigned)(ridx[i] == RAD_NONE);
        rad_from_table = any_none == 0;
    }
    if (rad_from_table) {
        HIPCHK(c, hipMemcpyAsync(c->blob_dev.p, blob, (size_t)h.off[1], hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->blob_dev.p + h.off[2], (const uint8_t*)blob + h.off[2], (size_t)(h.bytes - h.off[2]), hipMemcpyHostToDevice, c->stream));
    } else {
        HIPCHK(c, hipMemcpyAsync(c->blob_dev.p, blob, (size_t)h.bytes, hipMemcpyHostToDevice, c->stream));
    }
    borrow_blob_views(c, h);
    return validate_resident_blob(c, h, "arp_set_blob", nullptr, /*gather_sb=*/true, rad_from_table);   // (one launch: checks + single-bond neighbour coordinates)
}

int arp_get_blob(arp_ctx* c, void* host, uint64_t cap, uint64_t* bytes) {
    if (!c || !bytes) return ARP_E_ARG;
    if (c->blob_bytes == 0 || !c->blob_dev.p) FAIL(c, ARP_E_ARG, "arp_get_blob: the resident structure did not come from a blob");
    *bytes = c->blob_bytes;
    if (!host) return ARP_OK;
    if (cap < c->blob_bytes) FAIL(c, ARP_E_CAPACITY, "arp_get_blob: buffer too small");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(host, c->blob_dev.p, (size_t)c->blob_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ARP_OK;
}

// ---- sharded structures assembled on the device ---------------------------------------------------------
namespace {
const uint64_t REC_ESIZE[5] = {sizeof(arp_rec_atom), 3 * sizeof(double), sizeof(int32_t), sizeof(arp_rec_ring), sizeof(arp_rec_amide)};
bool rec_counts_ok(int64_t na, int64_t nh, int64_t nb, int64_t nring, int64_t namide) {
    const int64_t lim = 0x7FFFFFF0LL / 3;
    return na >= 0 && nh >= 0 && nb >= 0 && nring >= 0 && namide >= 0 && na <= lim && nh <= lim && nb <= 0x7FFFFFF0LL && nring <= lim &&
           namide <= lim;
}
int check_rec_header(arp_ctx* c, const arp_rec_header& h, uint64_t bytes, const char* who) {
    const std::string w(who);
    if (h.magic != ARP_REC_MAGIC) FAIL(c, ARP_E_ARG, w + ": bad magic");
    if (!rec_counts_ok(h.na, h.nh, h.nb, h.nring, h.namide)) FAIL(c, ARP_E_ARG, w + ": counts out of range");
    if (h.bytes != arp_records_size(h.na, h.nh, h.nb, h.nring, h.namide) || h.bytes > bytes) FAIL(c, ARP_E_ARG, w + ": size does not match the counts");
    const int64_t cnt[5] = {h.na, h.nh, h.nb, h.nring, h.namide};
    uint64_t off = align16(sizeof(arp_rec_header));
    for (int k = 0; k < 5; ++k) {
        if (h.off[k] != off) FAIL(c, ARP_E_ARG, w + ": unexpected section offset");
        off = align16(off + REC_ESIZE[k] * (uint64_t)cnt[k]);
    }
    if (h.n_rad < 0 || h.n_rad > RAD_TABLE) FAIL(c, ARP_E_ARG, w + ": n_rad out of range");
    return ARP_OK;
}
RecList rec_list(const uint8_t* base, const arp_rec_header& h) {
    RecList l;
    l.a = (const arp_rec_atom*)(base + h.off[0]); l.h = (const double*)(base + h.off[1]); l.b = (const int*)(base + h.off[2]);
    l.r = (const arp_rec_ring*)(base + h.off[3]); l.m = (const arp_rec_amide*)(base + h.off[4]);
    l.na = (int)h.na; l.nh = (int)h.nh; l.nb = (int)h.nb; l.nr = (int)h.nring; l.nm = (int)h.namide;
    return l;
}
RecList empty_rec_list() {
    RecList l;
    l.a = nullptr; l.h = nullptr; l.b = nullptr; l.r = nullptr; l.m = nullptr;
    l.na = l.nh = l.nb = l.nr = l.nm = 0;
    return l;
}
}  // namespace

uint64_t arp_records_size(int64_t na, int64_t nh, int64_t nb, int64_t nring, int64_t namide) {
    if (!rec_counts_ok(na, nh, nb, nring, namide)) return 0;
    const int64_t cnt[5] = {na, nh, nb, nring, namide};
    uint64_t off = align16(sizeof(arp_rec_header));
    for (int k = 0; k < 5; ++k) off = align16(off + REC_ESIZE[k] * (uint64_t)cnt[k]);
    return off;
}

int arp_records_layout(void* buf, uint64_t bytes, int64_t na, int64_t nh, int64_t nb, int64_t nring, int64_t namide) {
    const uint64_t need = arp_records_size(na, nh, nb, nring, namide);
    if (!buf || need == 0 || bytes < need) return ARP_E_ARG;
    arp_rec_header h;
    memset(&h, 0, sizeof(h));
    h.magic = ARP_REC_MAGIC;
    h.bytes = need;
    h.na = na; h.nh = nh; h.nb = nb; h.nring = nring; h.namide = namide;
    const int64_t cnt[5] = {na, nh, nb, nring, namide};
    uint64_t off = align16(sizeof(arp_rec_header));
    for (int k = 0; k < 5; ++k) {
        h.off[k] = off;
        off = align16(off + REC_ESIZE[k] * (uint64_t)cnt[k]);
    }
    memcpy(buf, &h, sizeof(h));
    return ARP_OK;
}

int arp_records_fill(void* buf, uint64_t bytes, int64_t n_total, int64_t nres_total, int64_t nring_total, int64_t namide_total, const float* xyz, const double* vdw, const double* cov,
                     const uint16_t* type_mask, const uint16_t* flags, const int32_t* res_id, const uint8_t* res_flags,
                     const int32_t* res_prev, const int32_t* res_next, const int32_t* bond_off, const int32_t* bond_idx,
                     const int32_t* h_off, const double* h_xyz, const int32_t* sb_nbr, const double* ring_center,
                     const double* ring_normal, const int32_t* ring_res, const float* amide_center, const float* amide_normal,
                     const int32_t* amide_res, const uint8_t* sel, const int64_t* atom_ids, const int64_t* ring_ids,
                     const int64_t* amide_ids) {
    if (!buf || bytes < sizeof(arp_rec_header)) return ARP_E_ARG;
    arp_rec_header h;
    memcpy(&h, buf, sizeof(h));
    if (h.magic != ARP_REC_MAGIC || h.bytes > bytes || h.bytes != arp_records_size(h.na, h.nh, h.nb, h.nring, h.namide)) return ARP_E_ARG;
    if ((h.na > 0 && (!atom_ids || !xyz || !vdw || !cov || !type_mask || !flags || !res_id || !res_flags || !res_prev || !res_next || !bond_off ||
                      !h_off || !sb_nbr)) ||
        (h.nring > 0 && (!ring_ids || !ring_center || !ring_normal || !ring_res)) ||
        (h.namide > 0 && (!amide_ids || !amide_center || !amide_normal || !amide_res)))
        return ARP_E_ARG;
    uint8_t* const b = (uint8_t*)buf;
    memset(b + sizeof(h), 0, (size_t)h.bytes - sizeof(h));
    arp_rec_atom* A = (arp_rec_atom*)(b + h.off[0]);
    double* H = (double*)(b + h.off[1]);
    int32_t* B = (int32_t*)(b + h.off[2]);
    arp_rec_ring* R = (arp_rec_ring*)(b + h.off[3]);
    arp_rec_amide* M = (arp_rec_amide*)(b + h.off[4]);
    int64_t hs = 0, bs = 0;
    struct Pair { uint64_t a, b; };
    std::vector<Pair> uniq;
    auto bits = [](double d) { uint64_t u; memcpy(&u, &d, 8); return u; };
    for (int64_t k = 0; k < h.na; ++k) {
        const int64_t i = atom_ids[k];
        if (i < 0 || i >= n_total || (k > 0 && atom_ids[k - 1] >= i)) return ARP_E_ARG;
        arp_rec_atom& r = A[k];
        r.x = xyz[3 * i]; r.y = xyz[3 * i + 1]; r.z = xyz[3 * i + 2]; r.gid = (int32_t)i;
        r.vdw = vdw[i]; r.cov = cov[i];
        const int32_t nb = sb_nbr[i];
        if (nb < -1 || nb >= n_total) return ARP_E_ARG;
        if (nb >= 0) { r.sb_x = xyz[3 * (int64_t)nb]; r.sb_y = xyz[3 * (int64_t)nb + 1]; r.sb_z = xyz[3 * (int64_t)nb + 2]; r.sb_has = 1; }
        const int32_t res = res_id[i];
        if (res < 0 || res >= nres_total) return ARP_E_ARG;
        r.res_gid = res; r.res_prev = res_prev[res]; r.res_next = res_next[res]; r.res_flags = res_flags[res];
        r.tmask = type_mask[i]; r.flags = flags[i];
        r.sel = sel ? sel[i] : (uint8_t)1;
        r.h_start = (int32_t)hs; r.h_cnt = h_off[i + 1] - h_off[i];
        r.bond_start = (int32_t)bs; r.bond_cnt = bond_off[i + 1] - bond_off[i];
        if (r.h_cnt < 0 || r.bond_cnt < 0 || hs + r.h_cnt > h.nh || bs + r.bond_cnt > h.nb) return ARP_E_ARG;
        if (r.h_cnt) memcpy(H + 3 * hs, h_xyz + 3 * (int64_t)h_off[i], (size_t)r.h_cnt * 24);
        if (r.bond_cnt) memcpy(B + bs, bond_idx + bond_off[i], (size_t)r.bond_cnt * 4);
        hs += r.h_cnt; bs += r.bond_cnt;
        const Pair key{bits(r.vdw), bits(r.cov)};
        bool seen = false;
        for (const Pair& u : uniq) if (u.a == key.a && u.b == key.b) { seen = true; break; }
        if (!seen && uniq.size() < 4096) uniq.push_back(key);     // (a handful of element values in practice)
    }
    if (hs != h.nh || bs != h.nb) return ARP_E_ARG;
    for (int64_t k = 0; k < h.nring; ++k) {
        const int64_t i = ring_ids[k];
        if (i < 0 || i >= nring_total || (k > 0 && ring_ids[k - 1] >= i) || ring_res[i] < -1 || ring_res[i] >= nres_total) return ARP_E_ARG;
        for (int q = 0; q < 3; ++q) { R[k].c[q] = ring_center[3 * i + q]; R[k].n[q] = ring_normal[3 * i + q]; }
        R[k].gid = (int32_t)i; R[k].res = ring_res[i];
    }
    for (int64_t k = 0; k < h.namide; ++k) {
        const int64_t i = amide_ids[k];
        if (i < 0 || i >= namide_total || (k > 0 && amide_ids[k - 1] >= i) || amide_res[i] < -1 || amide_res[i] >= nres_total) return ARP_E_ARG;
        for (int q = 0; q < 3; ++q) { M[k].c[q] = amide_center[3 * i + q]; M[k].n[q] = amide_normal[3 * i + q]; }
        M[k].gid = (int32_t)i; M[k].res = amide_res[i];
    }
    std::sort(uniq.begin(), uniq.end(), [](const Pair& p, const Pair& q) { return p.a != q.a ? p.a < q.a : p.b < q.b; });
    h.n_rad = (int64_t)std::min<size_t>(uniq.size(), RAD_TABLE);
    memset(h.rad_tab, 0, sizeof(h.rad_tab));
    for (int64_t k = 0; k < h.n_rad; ++k) { memcpy(&h.rad_tab[2 * k], &uniq[(size_t)k].a, 8); memcpy(&h.rad_tab[2 * k + 1], &uniq[(size_t)k].b, 8); }
    auto box = [](double* lo, double* hi, int64_t cnt, auto coord) {
        for (int q = 0; q < 3; ++q) lo[q] = hi[q] = cnt > 0 ? coord(0, q) : 0.0;
        for (int64_t k = 1; k < cnt; ++k)
            for (int q = 0; q < 3; ++q) { const double v = coord(k, q); lo[q] = std::min(lo[q], v); hi[q] = std::max(hi[q], v); }
    };
    box(h.lo, h.hi, h.na, [&](int64_t k, int q) { return (double)(&A[k].x)[q]; });
    box(h.ring_lo, h.ring_hi, h.nring, [&](int64_t k, int q) { return R[k].c[q]; });
    box(h.amide_lo, h.amide_hi, h.namide, [&](int64_t k, int q) { return (double)M[k].c[q]; });
    memcpy(buf, &h, sizeof(h));
    return ARP_OK;
}

int arp_shard_set_home(arp_ctx* c, const void* records, uint64_t bytes) {
    if (!c || !records || bytes < sizeof(arp_rec_header)) return ARP_E_ARG;
    arp_rec_header h;
    memcpy(&h, records, sizeof(h));
    CHK(check_rec_header(c, h, bytes, "arp_shard_set_home"));
    {   // the CSR runs of the home records are read by the face kernels: check them here (the merge checks received buffers)
        const arp_rec_atom* a = (const arp_rec_atom*)((const uint8_t*)records + h.off[0]);
        for (int64_t i = 0; i < h.na; ++i) {
            const bool ok = a[i].h_cnt >= 0 && a[i].bond_cnt >= 0 && a[i].h_start >= 0 && a[i].bond_start >= 0 &&
                            (int64_t)a[i].h_start + a[i].h_cnt <= h.nh && (int64_t)a[i].bond_start + a[i].bond_cnt <= h.nb &&
                            (i == 0 || a[i - 1].gid < a[i].gid);
            if (!ok) FAIL(c, ARP_E_ARG, "arp_shard_set_home: atom records must ascend by gid and their runs must lie inside the sections");
        }
    }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, c->rec_home.reserve((size_t)h.bytes));
    HIPCHK(c, hipMemcpyAsync(c->rec_home.p, records, (size_t)h.bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->rec_home_hdr = h;
    c->has_rec_home = true;
    return ARP_OK;
}

int arp_shard_pack_face(arp_ctx* c, int slot, double x_lo, double x_hi, uint64_t* device_ptr, uint64_t* out_bytes) {
    if (!c || slot < 0 || slot > 1 || !device_ptr || !out_bytes || std::isnan(x_lo) || std::isnan(x_hi)) return ARP_E_ARG;
    if (!c->has_rec_home) FAIL(c, ARP_E_ARG, "arp_shard_pack_face: call arp_shard_set_home first");
    HIPCHK(c, hipSetDevice(c->device));
    const arp_rec_header& H = c->rec_home_hdr;
    const RecList home = rec_list(c->rec_home.p, H);
    // five flag / count arrays, each one longer than its set (the scans leave the totals there)
    const size_t na = (size_t)H.na, nr = (size_t)H.nring, nm = (size_t)H.namide;
    HIPCHK(c, c->sh_scan.reserve(3 * (na + 1) + (nr + 1) + (nm + 1)));
    int* fa = c->sh_scan.p; int* fh = fa + na + 1; int* fb = fh + na + 1; int* fr = fb + na + 1; int* fm = fr + nr + 1;
    hipLaunchKernelGGL(k_face_flags, dim3(nblocks((int64_t)std::max({na, nr, nm, (size_t)1}), 256)), dim3(256), 0, c->stream, home, x_lo, x_hi,
                       fa, fh, fb, fr, fm);
    ScanSegs S;
    S.p[0] = fa; S.p[1] = fh; S.p[2] = fb; S.p[3] = fr; S.p[4] = fm;
    S.n[0] = S.n[1] = S.n[2] = (int)na; S.n[3] = (int)nr; S.n[4] = (int)nm;
    hipLaunchKernelGGL(k_scan_segments, dim3(5), dim3(1024), 0, c->stream, S);
    CHK(check_launch(c, "k_face_flags / k_scan_segments"));
    int tot[5];
    const int* last[5] = {fa + na, fh + na, fb + na, fr + nr, fm + nm};
    for (int k = 0; k < 5; ++k) HIPCHK(c, hipMemcpyAsync(&tot[k], last[k], sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const uint64_t bytes = arp_records_size(tot[0], tot[1], tot[2], tot[3], tot[4]);
    if (bytes == 0) FAIL(c, ARP_E_ARG, "arp_shard_pack_face: counts out of range");
    std::vector<uint8_t> hb(sizeof(arp_rec_header));
    arp_records_layout(hb.data(), bytes, tot[0], tot[1], tot[2], tot[3], tot[4]);
    arp_rec_header F;
    memcpy(&F, hb.data(), sizeof(F));
    // any box that contains the points will do: the home boxes, clipped to the x range that was asked for
    for (int k = 0; k < 3; ++k) {
        F.lo[k] = H.lo[k]; F.hi[k] = H.hi[k]; F.ring_lo[k] = H.ring_lo[k]; F.ring_hi[k] = H.ring_hi[k];
        F.amide_lo[k] = H.amide_lo[k]; F.amide_hi[k] = H.amide_hi[k];
    }
    auto clip = [&](double& lo, double& hi) {
        const double l = std::max(lo, x_lo), u = std::min(hi, x_hi);
        if (l <= u) { lo = l; hi = u; }    // (an empty face keeps the home box; its points are none)
    };
    clip(F.lo[0], F.hi[0]); clip(F.ring_lo[0], F.ring_hi[0]); clip(F.amide_lo[0], F.amide_hi[0]);
    F.n_rad = H.n_rad;
    memcpy(F.rad_tab, H.rad_tab, sizeof(F.rad_tab));
    DevBuf<uint8_t>& out = c->rec_face[slot];
    HIPCHK(c, out.reserve((size_t)bytes));
    HIPCHK(c, hipMemsetAsync(out.p, 0, (size_t)bytes, c->stream));         // alignment gaps travel too: keep them defined
    HIPCHK(c, hipMemcpyAsync(out.p, &F, sizeof(F), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_face_write, dim3(nblocks((int64_t)std::max({na, nr, nm, (size_t)1}), 256)), dim3(256), 0, c->stream, home, fa, fh, fb, fr,
                       fm, (arp_rec_atom*)(out.p + F.off[0]), (double*)(out.p + F.off[1]), (int*)(out.p + F.off[2]),
                       (arp_rec_ring*)(out.p + F.off[3]), (arp_rec_amide*)(out.p + F.off[4]));
    CHK(check_launch(c, "k_face_write"));
    HIPCHK(c, hipStreamSynchronize(c->stream));     // the caller hands the buffer to another stream (RCCL)
    *device_ptr = (uint64_t)(uintptr_t)out.p;
    *out_bytes = bytes;
    return ARP_OK;
}

int arp_shard_assemble(arp_ctx* c, uint64_t dev_left, uint64_t bytes_left, uint64_t dev_right, uint64_t bytes_right, int64_t nres_global,
                       int64_t counts[3]) {
    if (!c || nres_global < 0) return ARP_E_ARG;
    CHK(join_upload_lists(c));
    if (!c->has_rec_home) FAIL(c, ARP_E_ARG, "arp_shard_assemble: call arp_shard_set_home first");
    HIPCHK(c, hipSetDevice(c->device));
    batch_reset(c);      // (a shard is one structure)
    arp_rec_header hd[3];
    hd[0] = c->rec_home_hdr;
    const uint8_t* base[3] = {c->rec_home.p, (const uint8_t*)(uintptr_t)dev_left, (const uint8_t*)(uintptr_t)dev_right};
    const uint64_t given[3] = {hd[0].bytes, bytes_left, bytes_right};
    RecLists L;
    L.l[0] = rec_list(base[0], hd[0]);
    for (int s = 1; s < 3; ++s) {
        if (!base[s]) { L.l[s] = empty_rec_list(); memset(&hd[s], 0, sizeof(hd[s])); continue; }
        if (given[s] < sizeof(arp_rec_header)) FAIL(c, ARP_E_ARG, "arp_shard_assemble: halo buffer shorter than its header");
        HIPCHK(c, hipMemcpyAsync(&hd[s], base[s], sizeof(arp_rec_header), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int s = 1; s < 3; ++s)
        if (base[s]) {
            CHK(check_rec_header(c, hd[s], given[s], "arp_shard_assemble"));
            L.l[s] = rec_list(base[s], hd[s]);
        }
    const int64_t n = hd[0].na + hd[1].na + hd[2].na, nh = hd[0].nh + hd[1].nh + hd[2].nh;
    const int64_t nring = hd[0].nring + hd[1].nring + hd[2].nring, namide = hd[0].namide + hd[1].namide + hd[2].namide;
    if (!rec_counts_ok(n, nh, 0, nring, namide)) FAIL(c, ARP_E_ARG, "arp_shard_assemble: merged counts out of range");
    if (n > 0 && nres_global <= 0) FAIL(c, ARP_E_ARG, "arp_shard_assemble: atoms without a residue table");
    // positions, hydrogen and local-bond counts, their scans
    HIPCHK(c, c->sh_scan.reserve(2 * ((size_t)n + 1) + 4));
    HIPCHK(c, c->sh_src.reserve((size_t)std::max<int64_t>(n, 1)));
    int* hoff = c->sh_scan.p; int* boff = hoff + n + 1; int* d_err = boff + n + 1;
    HIPCHK(c, hipMemsetAsync(d_err, 0, sizeof(int), c->stream));
    if (n > 0) {
        hipLaunchKernelGGL(k_merge_positions, dim3(nblocks(n, 256)), dim3(256), 0, c->stream, L, c->sh_src.p, hoff, boff, d_err);
        CHK(check_launch(c, "k_merge_positions"));
    }
    ScanSegs S;
    memset(&S, 0, sizeof(S));
    S.p[0] = hoff; S.p[1] = boff; S.n[0] = S.n[1] = (int)n;
    hipLaunchKernelGGL(k_scan_segments, dim3(2), dim3(1024), 0, c->stream, S);
    CHK(check_launch(c, "k_scan_segments"));
    int tot[3] = {0, 0, 0};
    HIPCHK(c, hipMemcpyAsync(&tot[0], hoff + n, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&tot[1], boff + n, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&tot[2], d_err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (tot[2] != 0 || tot[0] != nh)
        FAIL(c, ARP_E_ARG, "arp_shard_assemble: an atom occurs twice in home + halos, a list is not ascending, or a record's runs leave its sections");
    const int64_t nbond = tot[1];
    // the blob the merged structure lives in
    const uint64_t bytes = arp_blob_size(n, nres_global, nbond, nh, nring, namide);
    if (bytes == 0) FAIL(c, ARP_E_ARG, "arp_shard_assemble: merged counts out of range");
    std::vector<uint8_t> hb(sizeof(arp_blob_header));
    arp_blob_layout(hb.data(), bytes, n, nres_global, nbond, nh, nring, namide);
    arp_blob_header B;
    memcpy(&B, hb.data(), sizeof(B));
    auto merge_box = [&](double* lo, double* hi, int which) {
        bool any = false;
        for (int s = 0; s < 3; ++s) {
            const int64_t cnt = which == 0 ? hd[s].na : (which == 1 ? hd[s].nring : hd[s].namide);
            if (cnt == 0) continue;
            const double* l = which == 0 ? hd[s].lo : (which == 1 ? hd[s].ring_lo : hd[s].amide_lo);
            const double* u = which == 0 ? hd[s].hi : (which == 1 ? hd[s].ring_hi : hd[s].amide_hi);
            for (int k = 0; k < 3; ++k) {
                lo[k] = any ? std::min(lo[k], l[k]) : l[k];
                hi[k] = any ? std::max(hi[k], u[k]) : u[k];
            }
            any = true;
        }
        if (!any) for (int k = 0; k < 3; ++k) lo[k] = hi[k] = 0.0;
    };
    merge_box(B.lo, B.hi, 0); merge_box(B.ring_lo, B.ring_hi, 1); merge_box(B.amide_lo, B.amide_hi, 2);
    B.n_rad = hd[0].n_rad;
    CHK(check_blob_header(c, B, bytes));
    HIPCHK(c, c->blob_dev.reserve((size_t)bytes));
    HIPCHK(c, hipMemsetAsync(c->blob_dev.p, 0, (size_t)bytes, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->blob_dev.p, &B, sizeof(B), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->blob_dev.p + B.off[20], hd[0].rad_tab, sizeof(double) * 2 * RAD_TABLE, hipMemcpyHostToDevice, c->stream));
    borrow_blob_views(c, B);
    const size_t n1 = (size_t)std::max<int64_t>(n, 1), r1 = (size_t)std::max<int64_t>(nring, 1), m1 = (size_t)std::max<int64_t>(namide, 1);
    HIPCHK(c, c->sb.reserve(n1)); HIPCHK(c, c->gid.reserve(n1)); HIPCHK(c, c->home.reserve(n1)); HIPCHK(c, c->origin.reserve(n1));
    HIPCHK(c, c->sh_sel.reserve(n1));
    HIPCHK(c, c->ring_gid.reserve(r1)); HIPCHK(c, c->ring_home.reserve(r1)); HIPCHK(c, c->ring_origin.reserve(r1));
    HIPCHK(c, c->am_gid.reserve(m1)); HIPCHK(c, c->am_home.reserve(m1)); HIPCHK(c, c->am_origin.reserve(m1));
    if (nres_global > 0) {
        hipLaunchKernelGGL(k_init_residues, dim3(nblocks(nres_global, 256)), dim3(256), 0, c->stream, (int)nres_global, c->res_flags.p,
                           c->res_prev.p, c->res_next.p);
        CHK(check_launch(c, "k_init_residues"));
    }
    MergeOut O;
    O.n = (int)n; O.nres = (int)nres_global; O.n_rad = (int)B.n_rad;
    O.xyz = c->xyz.p; O.rad = c->rad.p; O.tmask = c->tmask.p; O.flags = c->flags.p; O.res_id = c->res_id.p;
    O.res_flags = c->res_flags.p; O.res_prev = c->res_prev.p; O.res_next = c->res_next.p;
    O.bond_off = c->bond_off.p; O.bond_idx = c->bond_idx.p; O.h_off = c->h_off.p; O.h_xyz = c->h_xyz_d.p; O.sb_nbr = c->blob_sb_nbr.p;
    O.rad_idx = c->rad_idx.p; O.rad_tab = c->rad_tab.p;
    O.sb = c->sb.p; O.gid = c->gid.p; O.home = c->home.p; O.origin = c->origin.p; O.sel = c->sh_sel.p;
    hipLaunchKernelGGL(k_merge_fill, dim3(nblocks(std::max<int64_t>(n, 1), 256)), dim3(256), 0, c->stream, L, O, c->sh_src.p, hoff, boff, d_err);
    CHK(check_launch(c, "k_merge_fill"));
    GroupOut G;
    G.ring_c = c->ring_c.p; G.ring_n = c->ring_n.p; G.ring_res = c->ring_res.p; G.ring_gid = c->ring_gid.p; G.ring_home = c->ring_home.p;
    G.ring_origin = c->ring_origin.p;
    G.am_c = c->am_c.p; G.am_n = c->am_n.p; G.am_res = c->am_res.p; G.am_gid = c->am_gid.p; G.am_home = c->am_home.p; G.am_origin = c->am_origin.p;
    if (nring + namide > 0) {
        hipLaunchKernelGGL(k_merge_groups, dim3(nblocks(std::max(nring, namide), 256)), dim3(256), 0, c->stream, L, G, d_err);
        CHK(check_launch(c, "k_merge_groups"));
    }
    CHK(validate_resident_blob(c, B, "arp_shard_assemble", d_err));     // waits; resets selection and ownership state
    c->has_gid = c->has_home = true;
    c->gid_max = -1;      // (the merged ids live on the device only: the sort takes the 31-bit key width)
    c->has_group_owner = true;
    c->shard_resident = true;
    if (counts) { counts[0] = n; counts[1] = nring; counts[2] = namide; }
    return ARP_OK;
}

int arp_shard_layout(arp_ctx* c, int32_t* global_id, int8_t* origin, uint8_t* sel, int32_t* ring_gid, int8_t* ring_origin,
                     int32_t* amide_gid, int8_t* amide_origin) {
    if (!c) return ARP_E_ARG;
    if (!c->shard_resident) FAIL(c, ARP_E_ARG, "arp_shard_layout: no structure assembled by arp_shard_assemble is resident");
    HIPCHK(c, hipSetDevice(c->device));
    CHK(download_async(c, global_id, c->gid.p, (size_t)c->n));
    CHK(download_async(c, origin, c->origin.p, (size_t)c->n));
    CHK(download_async(c, sel, c->sh_sel.p, (size_t)c->n));
    CHK(download_async(c, ring_gid, c->ring_gid.p, (size_t)c->nring));
    CHK(download_async(c, ring_origin, c->ring_origin.p, (size_t)c->nring));
    CHK(download_async(c, amide_gid, c->am_gid.p, (size_t)c->namide));
    CHK(download_async(c, amide_origin, c->am_origin.p, (size_t)c->namide));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ARP_OK;
}

int arp_set_ownership(arp_ctx* c, const uint8_t* is_home, const int32_t* global_id) {
    if (!c) return ARP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    batch_reset(c);      // (the partition of a batch was declared for the arrays that were resident then: arp_set_batch again after this call)
    c->static_dirty = true; c->lists_from_upload = false;
    if (is_home) { CHK(upload(c, c->home, is_home, (size_t)c->n)); c->has_home = true; }
    else c->has_home = false;
    if (global_id) {
        if (c->n > 0 && global_id[0] < 0) FAIL(c, ARP_E_ARG, "arp_set_ownership: global_id must not be negative");
        for (int64_t i = 1; i < c->n; ++i)
            if (global_id[i] <= global_id[i - 1]) FAIL(c, ARP_E_ARG, "arp_set_ownership: global_id must be strictly increasing");
        CHK(upload(c, c->gid, global_id, (size_t)c->n));
        c->has_gid = true;
        c->gid_max = c->n > 0 ? (int64_t)global_id[c->n - 1] : -1;
    } else { c->has_gid = false; c->gid_max = -1; }
    c->atom_grid.valid = false;   // M_HOME is part of the sorted records
    c->all_grid_current = false;
    c->contacts_valid = false;
    return ARP_OK;
}

int arp_set_group_ownership(arp_ctx* c, const uint8_t* ring_home, const int32_t* ring_gid, const uint8_t* amide_home,
                            const int32_t* amide_gid) {
    if (!c) return ARP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    batch_reset(c);      // (a shard is one structure)
    c->static_dirty = true; c->lists_from_upload = false;
    if (!ring_home && !ring_gid && !amide_home && !amide_gid) { c->has_group_owner = false; return ARP_OK; }
    if ((c->nring > 0 && (!ring_home || !ring_gid)) || (c->namide > 0 && (!amide_home || !amide_gid)))
        FAIL(c, ARP_E_ARG, "arp_set_group_ownership: all four arrays are required");
    for (int64_t i = 1; i < c->nring; ++i)
        if (ring_gid[i] <= ring_gid[i - 1]) FAIL(c, ARP_E_ARG, "arp_set_group_ownership: ring_gid must be strictly increasing");
    for (int64_t i = 1; i < c->namide; ++i)
        if (amide_gid[i] <= amide_gid[i - 1]) FAIL(c, ARP_E_ARG, "arp_set_group_ownership: amide_gid must be strictly increasing");
    CHK(upload(c, c->ring_home, ring_home, (size_t)c->nring)); CHK(upload(c, c->ring_gid, ring_gid, (size_t)c->nring));
    CHK(upload(c, c->am_home, amide_home, (size_t)c->namide)); CHK(upload(c, c->am_gid, amide_gid, (size_t)c->namide));
    c->has_group_owner = true;
    return ARP_OK;
}

int arp_set_single_bond_neighbour_coords(arp_ctx* c, const float* sb_xyz, const uint8_t* sb_present) {
    if (!c || (c->n > 0 && (!sb_xyz || !sb_present))) return ARP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    c->static_dirty = true; c->lists_from_upload = false;
    std::vector<float4> sb((size_t)c->n);
    for (int64_t i = 0; i < c->n; ++i)
        sb[i] = sb_present[i] ? make_float4(sb_xyz[3 * i], sb_xyz[3 * i + 1], sb_xyz[3 * i + 2], 1.0f) : make_float4(0, 0, 0, 0);
    CHK(upload(c, c->sb, sb.data(), (size_t)c->n));
    c->contacts_valid = false;
    return ARP_OK;
}

int arp_set_selection_state(arp_ctx* c, const uint8_t* in_selection, const uint8_t* in_plus, const uint8_t* ring_sel,
                            const uint8_t* ring_plus, const uint8_t* amide_sel, const uint8_t* amide_plus) {
    if (!c) return ARP_E_ARG;
    if ((c->n > 0 && (!in_selection || !in_plus)) || (c->nring > 0 && (!ring_sel || !ring_plus)) ||
        (c->namide > 0 && (!amide_sel || !amide_plus)))
        FAIL(c, ARP_E_ARG, "arp_set_selection_state: null input");
    HIPCHK(c, hipSetDevice(c->device));
    c->static_dirty = true; c->lists_from_upload = false;
    CHK(upload(c, c->sel, in_selection, (size_t)c->n));
    CHK(upload(c, c->plus, in_plus, (size_t)c->n));
    c->sel_uploaded = true;
    c->nsel = -1;
    c->sel_all = false;   // the caller's masks are taken as they are
    ++c->sel_epoch;
    CHK(upload(c, c->ring_sel, ring_sel, (size_t)c->nring)); CHK(upload(c, c->ring_plus, ring_plus, (size_t)c->nring));
    CHK(upload(c, c->am_sel, amide_sel, (size_t)c->namide)); CHK(upload(c, c->am_plus, amide_plus, (size_t)c->namide));
    c->sel_made = true;
    c->atom_grid.valid = false;
    c->all_grid_current = false;
    c->contacts_valid = false;
    c->bag_ap.valid = c->bag_pp.valid = c->bag_gg.valid = c->bag_gp.valid = false;
    return ARP_OK;
}

int arp_set_selection(arp_ctx* c, const uint8_t* in_selection) {
    if (!c || (c->n > 0 && !in_selection)) return ARP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    CHK(upload(c, c->sel, in_selection, (size_t)c->n));
    c->sel_uploaded = true;
    // how many atoms are selected decides how selection_plus is computed (arp_run_launch): everything -> nothing to do,
    // a handful -> direct test of every atom against the list (k_expand_small), otherwise the 6 A grid search
    int64_t nsel = 0;
    std::vector<int> list;
    for (int64_t i = 0; i < c->n; ++i)
        if (in_selection[i]) {
            if (nsel < SMALL_SEL_MAX) list.push_back((int)i);
            ++nsel;
        }
    c->nsel = nsel;
    c->sel_all = (nsel == c->n);
    if (nsel > 0 && nsel <= SMALL_SEL_MAX) CHK(upload(c, c->sel_list, list.data(), (size_t)nsel));
    c->sel_made = false;  // expansion pending
    ++c->sel_epoch;
    c->all_grid_current = false;
    c->contacts_valid = false;
    return ARP_OK;
}

// ---- NeighborSearch.search_all -----------------------------------------------------------
int arp_search_all(arp_ctx* c, double radius, const uint8_t* active, int64_t cap, int32_t* out_i, int32_t* out_j, int64_t* count) {
    if (!c || !count || cap < 0 || !(radius > 0)) return ARP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const uint8_t* d_active = nullptr;
    if (active) {
        CHK(upload(c, c->tmp_u8, active, (size_t)c->n));
        d_active = c->tmp_u8.p;
    }
    CHK(build_contact_grid(c, radius, 0, 0, d_active));
    c->atom_grid.valid = false;  // not the contact grid
    c->contacts_valid = false;
    HIPCHK(c, c->pairs.reserve((size_t)std::max<int64_t>(cap, 1)));
    CHK(zero_counter(c, C_SEARCH_PAIRS, 1));
    CHK(zero_counter(c, C_STAT_MCAND, 2 * STAT_SLOTS));
    if (c->n > 0) {
        hipLaunchKernelGGL((k_search<MODE_PAIRS>), dim3(search_blocks(c->atom_grid.d)), dim3(64 * SEARCH_WAVES), 0, c->stream,
                           c->atom_grid.d, c->atom_grid.start.p, c->s_xyzm.p, c->s_aux.p, radius * radius, 1, 0, c->pairs.p,
                           (u64)cap, c->d_ctr + ctr_dev(C_SEARCH_PAIRS), c->d_ctr + ctr_dev(C_STAT_MCAND), c->d_ctr + ctr_dev(C_STAT_MACC), (uint8_t*)nullptr, GroupMasks{}, (const int*)nullptr, (const int*)nullptr);
        CHK(check_launch(c, "k_search<PAIRS>"));
    }
    CHK(read_counters(c));
    collect_events(c);
    *count = (int64_t)c->h_ctr[C_SEARCH_PAIRS];
    if (*count > cap) FAIL(c, ARP_E_CAPACITY, "arp_search_all: output buffer too small");
    if (*count > 0) {
        std::vector<int2> tmp((size_t)*count);
        CHK(download(c, tmp.data(), c->pairs.p, (size_t)*count));
        for (int64_t k = 0; k < *count; ++k) { out_i[k] = tmp[k].x; out_j[k] = tmp[k].y; }
    }
    return ARP_OK;
}

// ---- _make_selection --------------------------------------------------------------------
int arp_make_selection(arp_ctx* c, const uint8_t* in_selection, double expand_radius, uint8_t* out_plus, uint8_t* out_ring_sel,
                       uint8_t* out_ring_plus, uint8_t* out_amide_sel, uint8_t* out_amide_plus) {
    if (!c || !(expand_radius > 0)) return ARP_E_ARG;
    if (in_selection) CHK(arp_set_selection(c, in_selection));
    else {
        HIPCHK(c, hipSetDevice(c->device));
        CHK(default_selection(c));  // nothing uploaded for this structure: whole structure (I:1395)
    }
    CHK(enqueue_selection(c, expand_radius));
    CHK(read_counters(c));
    collect_events(c);
    c->stats[5] = (int64_t)c->h_ctr[C_MARK_CAND];
    c->stats[6] = (int64_t)c->h_ctr[C_MARK_ACC];
    CHK(download(c, out_plus, c->plus.p, out_plus ? (size_t)c->n : 0));
    CHK(download(c, out_ring_sel, c->ring_sel.p, out_ring_sel ? (size_t)c->nring : 0));
    CHK(download(c, out_ring_plus, c->ring_plus.p, out_ring_plus ? (size_t)c->nring : 0));
    CHK(download(c, out_amide_sel, c->am_sel.p, out_amide_sel ? (size_t)c->namide : 0));
    CHK(download(c, out_amide_plus, c->am_plus.p, out_amide_plus ? (size_t)c->namide : 0));
    return ARP_OK;
}

int arp_get_selection(arp_ctx* c, uint8_t* out_plus, uint8_t* out_ring_sel, uint8_t* out_ring_plus, uint8_t* out_amide_sel,
                      uint8_t* out_amide_plus) {
    if (!c) return ARP_E_ARG;
    if (!c->sel_made) FAIL(c, ARP_E_ARG, "arp_get_selection: no selection has been expanded yet");
    HIPCHK(c, hipSetDevice(c->device));
    CHK(download(c, out_plus, c->plus.p, out_plus ? (size_t)c->n : 0));
    CHK(download(c, out_ring_sel, c->ring_sel.p, out_ring_sel ? (size_t)c->nring : 0));
    CHK(download(c, out_ring_plus, c->ring_plus.p, out_ring_plus ? (size_t)c->nring : 0));
    CHK(download(c, out_amide_sel, c->am_sel.p, out_amide_sel ? (size_t)c->namide : 0));
    CHK(download(c, out_amide_plus, c->am_plus.p, out_amide_plus ? (size_t)c->namide : 0));
    return ARP_OK;
}

// ---- _calculate_atom_contacts -------------------------------------------------------------
int arp_atom_contacts_launch(arp_ctx* c, double cutoff, double vdw_comp, int include_sequence_adjacent, int64_t* count) {
    if (!c || !(cutoff > 0)) return ARP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    CHK(ensure_default_selection(c));
    for (int attempt = 0;; ++attempt) {
        CHK(enqueue_contacts(c, cutoff, vdw_comp, include_sequence_adjacent, false));
        CHK(read_counters(c));
        collect_events(c);
        if (!finish_contacts(c)) break;
        if (attempt == 2) FAIL(c, ARP_E_CAPACITY, "arp_atom_contacts_launch: pair buffer could not be sized");
        CHK(grow_pairs(c));
    }
    if (count) *count = c->n_contacts;
    return device_error(c);
}

int arp_atom_contacts_fetch(arp_ctx* c, int64_t cap, int32_t* out_i, int32_t* out_j, float* out_dist, uint16_t* out_sift,
                            uint8_t* out_ctype, int64_t* count) {
    if (!c || !count) return ARP_E_ARG;
    if (!c->contacts_valid) FAIL(c, ARP_E_ARG, "arp_atom_contacts_fetch: no launch results");
    HIPCHK(c, hipSetDevice(c->device));
    *count = c->n_contacts;
    if (c->n_contacts > cap) FAIL(c, ARP_E_CAPACITY, "arp_atom_contacts_fetch: output buffer too small");
    const size_t k = (size_t)c->n_contacts;
    // five copies in flight, one synchronisation (into buffers from arp_host_alloc they run at PCIe speed); the records come
    // in the canonical (i, j) order once arp_atom_contacts_sort has run on them, in the order of the pair list before
    if (c->contacts_sorted && c->sorted_is_csr && out_i) {      // (this call hands out bgn ids: the sorted columns once more, as records)
        const bool was = c->packed_csr;
        c->packed_csr = false;
        const int rc = sort_contacts(c);
        c->packed_csr = was;
        CHK(rc);
    }
    const uint8_t* sl = c->sorted_slab.p;
    const bool srt = c->contacts_sorted;
    if (k) {
        if (out_i) HIPCHK(c, hipMemcpyAsync(out_i, srt ? (const void*)(sl + c->srt_off[0]) : (const void*)c->out_i.p, k * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        if (out_j) HIPCHK(c, hipMemcpyAsync(out_j, srt ? (const void*)(sl + c->srt_off[1]) : (const void*)c->out_j.p, k * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        if (out_dist) HIPCHK(c, hipMemcpyAsync(out_dist, srt ? (const void*)(sl + c->srt_off[2]) : (const void*)c->out_d.p, k * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        if (out_sift) HIPCHK(c, hipMemcpyAsync(out_sift, srt ? (const void*)(sl + c->srt_off[3]) : (const void*)c->out_s.p, k * sizeof(uint16_t), hipMemcpyDeviceToHost, c->stream));
        if (out_ctype) HIPCHK(c, hipMemcpyAsync(out_ctype, srt ? (const void*)(sl + c->srt_off[4]) : (const void*)c->out_ct.p, k * sizeof(uint8_t), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ARP_OK;
}

int arp_atom_contacts_sort(arp_ctx* c) {
    if (!c) return ARP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    return sort_contacts(c);
}

// Canonical order of a ring / amide bag of MORE than BAG_SORT_MAX records (config 5: 42 k plane-plane records): the radix
// passes of the atom-atom bag (arp_sort.h) on {first id, second id} with the record's index riding as the payload — its bits in
// the float32 column — so that the sorted "distance" column IS the permutation k_pack_segments follows.
__global__ __launch_bounds__(256) void k_iota_bits(long long n, float* __restrict__ out, uint16_t* __restrict__ zs, uint8_t* __restrict__ zc) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        out[i] = __uint_as_float((uint32_t)i);
        zs[i] = 0;
        zc[i] = 0;
    }
}
int bag_order_large(arp_ctx* c, const int* first, const int* second, size_t k, int64_t idmax, DevBuf<uint32_t>& perm) {
    HIPCHK(c, perm.reserve(2 * k));                       // [0, k): the permutation; [k, 2 k): the indices as the sort's input column
    HIPCHK(c, c->bagsort_i.reserve(k)); HIPCHK(c, c->bagsort_j.reserve(k));
    HIPCHK(c, c->bagsort_s.reserve(2 * k)); HIPCHK(c, c->bagsort_ct.reserve(2 * k));
    int idbits = 1;
    while (((int64_t)1 << idbits) <= std::max<int64_t>(idmax, 1)) ++idbits;
    const int passes = (idbits + SORT_MAX_BITS - 1) / SORT_MAX_BITS;
    for (int q = 0; q < 2; ++q) { HIPCHK(c, c->bagsort_key[q].reserve(k)); HIPCHK(c, c->bagsort_val[q].reserve(k)); }
    const long long tiles = ((long long)k + SORT_TILE - 1) / SORT_TILE;
    const int tstride = (int)((tiles + 3) & ~3ll);
    HIPCHK(c, c->bagsort_table.reserve((size_t)SORT_BINS * (size_t)tstride));
    HIPCHK(c, c->bagsort_total.reserve(SORT_BINS));
    float* const idx_in = reinterpret_cast<float*>(perm.p + k);
    hipLaunchKernelGGL(k_iota_bits, dim3(nblocks((int64_t)k, 256, 1024)), dim3(256), 0, c->stream, (long long)k, idx_in, c->bagsort_s.p + k, c->bagsort_ct.p + k);
    SortArgs A{};
    A.ci = first; A.cj = second;
    A.d_in = idx_in; A.s_in = c->bagsort_s.p + k; A.ct_in = c->bagsort_ct.p + k;
    A.i_out = c->bagsort_i.p; A.j_out = c->bagsort_j.p; A.d_out = reinterpret_cast<float*>(perm.p);
    A.s_out = c->bagsort_s.p; A.ct_out = c->bagsort_ct.p;
    A.n = (long long)k; A.T = (int)tiles; A.tstride = tstride; A.jbits = idbits;
    A.table = c->bagsort_table.p; A.total = c->bagsort_total.p;
    int shift = idbits;
    for (int ps = 0; ps < passes; ++ps) {
        A.first = ps == 0; A.last = 0;
        A.shift = shift;
        A.bits = idbits / passes + (ps < idbits % passes ? 1 : 0);
        shift += A.bits;
        A.key_in = ps > 0 ? c->bagsort_key[(ps - 1) & 1].p : nullptr;
        A.val_in = ps > 0 ? c->bagsort_val[(ps - 1) & 1].p : nullptr;
        A.key_out = c->bagsort_key[ps & 1].p;
        A.val_out = c->bagsort_val[ps & 1].p;
        hipLaunchKernelGGL(k_sort_hist, dim3(A.T), dim3(SORT_THREADS), 0, c->stream, A);
        hipLaunchKernelGGL(k_sort_scan, dim3(1 << A.bits), dim3(SORT_THREADS), 0, c->stream, A);
        hipLaunchKernelGGL(k_sort_scatter, dim3(A.T), dim3(SORT_THREADS), 0, c->stream, A);
    }
    A.key_in = c->bagsort_key[(passes - 1) & 1].p;
    A.val_in = c->bagsort_val[(passes - 1) & 1].p;
    hipLaunchKernelGGL(k_sort_runs, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, c->stream, A);
    return check_launch(c, "bag_order_large");
}

// Every result of the last pass with ONE copy: the atom-atom bag in canonical order (sorted on the device if it is not yet)
// and the used prefixes of the four ring / amide bags behind it, gathered in HBM (k_pack_segments) and copied in one piece.
int arp_fetch_packed(arp_ctx* c, void* host, uint64_t host_bytes, int64_t counts[5], uint64_t offsets[ARP_PACKED_OFFSETS], uint64_t* bytes_used) {
    if (!c || !counts || !offsets || !bytes_used) return ARP_E_ARG;
    if (!c->contacts_valid) FAIL(c, ARP_E_ARG, "arp_fetch_packed: no launch results");
    HIPCHK(c, hipSetDevice(c->device));
    Bag* bags[4] = {&c->bag_pp, &c->bag_ap, &c->bag_gg, &c->bag_gp};      // (the order of get_contacts, I:183-210)
    static const size_t es[12] = {4, 4, 8, 8, 8, 8, 4, 4, 4, 1, 1, 1};
    size_t off[5], cbytes;
    const bool csr = c->packed_csr && !c->has_gid;
    if (c->packed_csr && c->has_gid) FAIL(c, ARP_E_ARG, "arp_fetch_packed: the row-offset layout needs packed atom ids (this context holds a shard with global ids)");
    sorted_layout((size_t)c->n_contacts, off, &cbytes, csr ? (size_t)std::max<int64_t>(c->n, 0) + 1 : (size_t)c->n_contacts);
    size_t total = cbytes;
    PackTable t;
    t.n = 0;
    for (int q = 0; q < ARP_PACKED_OFFSETS; ++q) offsets[q] = 0;
    for (int q = 0; q < 5; ++q) offsets[q] = off[q];
    counts[0] = c->n_contacts;
    // Sizes first: where every array of every bag goes in the one piece, and whether it fits — before anything is launched (a
    // caller with too small a buffer, or a result beyond what one piece can address, leaves with bytes_used and no work done)
    *bytes_used = 0;
    int seg_of[4][12];
    for (int b = 0; b < 4; ++b) {
        Bag& g = *bags[b];
        counts[1 + b] = g.valid ? g.count : 0;
        for (int q = 0; q < 12; ++q) seg_of[b][q] = -1;
        if (!g.valid || g.count == 0) continue;
        if ((uint64_t)g.count >= ((uint64_t)1 << 31)) FAIL(c, ARP_E_CAPACITY, "arp_fetch_packed: a ring / amide bag of 2^31 records or more (fetch the bags one by one)");
        const uint8_t* ptr[12] = {(const uint8_t*)g.a.p, (const uint8_t*)g.b.p, (const uint8_t*)g.d0.p, (const uint8_t*)g.d1.p,
                                  (const uint8_t*)g.d2.p, (const uint8_t*)g.d3.p, (const uint8_t*)g.f0.p, (const uint8_t*)g.f1.p,
                                  (const uint8_t*)g.f2.p, g.u0.p, g.u1.p, g.u2.p};
        for (int q = 0; q < 12; ++q) {
            if (!ptr[q]) continue;
            const size_t bytes = (size_t)g.count * es[q];
            if (t.n >= 48 || bytes >= ((size_t)1 << 32) || total >= ((size_t)1 << 32)) FAIL(c, ARP_E_CAPACITY, "arp_fetch_packed: ring / amide bags too large for one piece (fetch them one by one)");
            offsets[5 + 12 * b + q] = total;
            seg_of[b][q] = t.n;
            t.s[t.n++] = PackSeg{ptr[q], (uint32_t)total, (uint32_t)bytes, nullptr, (uint32_t)es[q]};
            total = (total + bytes + 15) & ~(size_t)15;
        }
    }
    *bytes_used = total;
    if (!host || host_bytes < total) FAIL(c, ARP_E_CAPACITY, "arp_fetch_packed: host buffer too small (bytes_used holds the size needed)");
    // canonical order of the small bags, made on the device: plane-plane, group-group, group-plane by (first id, second id),
    // atom-plane by (ring, atom) — the order the reference's loops create them in
    BagOrderArgs bo{};
    bool any_order = false;
    HIPCHK(c, c->bag_perm.reserve(4 * (size_t)BAG_SORT_MAX));
    for (int b = 0; b < 4; ++b) {
        Bag& g = *bags[b];
        const bool small = g.valid && g.count > 0 && g.count <= BAG_SORT_MAX;
        bo.first[b] = (b == 1) ? g.b.p : g.a.p;          // (atom-plane: a = atom, b = ring)
        bo.second[b] = (b == 1) ? g.a.p : g.b.p;
        bo.n[b] = small ? (int)g.count : 0;
        bo.perm[b] = c->bag_perm.p + (size_t)b * BAG_SORT_MAX;
        any_order = any_order || small;
    }
    const uint32_t* big_perm[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int b = 0; b < 4; ++b) {
        Bag& g = *bags[b];
        if (!(g.valid && g.count > BAG_SORT_MAX)) continue;
        const int64_t idmax = (c->has_gid || c->has_group_owner) ? (((int64_t)1 << 31) - 1) : std::max<int64_t>({c->n, c->nring, c->namide, 2}) - 1;      // (a shard's records carry global ids)
        CHK(bag_order_large(c, bo.first[b], bo.second[b], (size_t)g.count, idmax, c->bag_perm_big[b]));
        big_perm[b] = c->bag_perm_big[b].p;
    }
    // (on the second stream, beside the radix passes of the atom-atom bag: one block per bag, 80 us for a bag of 4096)
    const bool order_aside = any_order && c->stream2 && !c->external_stream;      // (beside the sort, which may be under way already: arp_set_sort_after_pass)
    for (int b = 0; b < 4; ++b)
        for (int q = 0; q < 12; ++q)
            if (seg_of[b][q] >= 0) t.s[seg_of[b][q]].perm = bo.n[b] > 0 ? bo.perm[b] : big_perm[b];
    c->contacts_sorted = c->contacts_sorted && c->sorted_slab.cap >= total && c->sorted_is_csr == csr;
    // (aside: the sort's launches go out first — they are the critical path, the host needs ~5 us per launch —, the one block per
    // small bag on the second stream behind them)
    if (any_order && !order_aside) {
        hipLaunchKernelGGL(k_bag_order, dim3(4), dim3(1024), 0, c->stream, bo);
        CHK(check_launch(c, "k_bag_order"));
    }
    CHK(sort_contacts(c, total - cbytes));
    if (any_order && order_aside) {
        hipLaunchKernelGGL(k_bag_order, dim3(4), dim3(1024), 0, c->stream2, bo);
        CHK(check_launch(c, "k_bag_order"));
        HIPCHK(c, hipEventRecord(c->ev_planes, c->stream2));
    }
    if (order_aside) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_planes, 0));
    if (t.n > 0) {
        hipLaunchKernelGGL(k_pack_segments, pack_grid(t), dim3(256), 0, c->stream, t, c->sorted_slab.p);
        CHK(check_launch(c, "k_pack_segments"));
    }
    if (total) {
        // A small piece (a protein's bags: a few hundred kilobytes) into a page-locked, device-visible buffer is written by a kernel:
        // the copy engine needs ~10 us to get going, which is as long as such a copy takes (stand-in end to end 0.155 -> see
        // profiles/README.md).  Anything larger, or a pageable buffer: the copy engine.
        static const size_t direct_max = (size_t)std::max(0, env_int("ARP_FETCH_DIRECT_MAX_KB", 1024)) << 10;
        void* host_dev = nullptr;
        if (total <= direct_max && host_bytes >= ((total + 15) & ~(uint64_t)15) && c->sorted_slab.cap >= ((total + 15) & ~(size_t)15)) {      // (whole quads on both sides)
            hipPointerAttribute_t at{};
            if (hipPointerGetAttributes(&at, host) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer) host_dev = at.devicePointer;
            else (void)hipGetLastError();
        }
        if (host_dev && ((uintptr_t)host_dev & 15) == 0) {
            const unsigned nq = (unsigned)((total + 15) / 16);
            hipLaunchKernelGGL(k_copy_quads, dim3(nblocks(nq, 256, 1024)), dim3(256), 0, c->stream, (const int4*)c->sorted_slab.p, (int4*)host_dev, nq);
            CHK(check_launch(c, "k_copy_quads"));
        } else {
            HIPCHK(c, hipMemcpyAsync(host, c->sorted_slab.p, total, hipMemcpyDeviceToHost, c->stream));
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ARP_OK;
}

int arp_atom_contacts(arp_ctx* c, double cutoff, double vdw_comp, int include_sequence_adjacent, int64_t cap, int32_t* out_i,
                      int32_t* out_j, float* out_dist, uint16_t* out_sift, uint8_t* out_ctype, int64_t* count) {
    if (!c || !count) return ARP_E_ARG;
    CHK(arp_atom_contacts_launch(c, cutoff, vdw_comp, include_sequence_adjacent, count));
    return arp_atom_contacts_fetch(c, cap, out_i, out_j, out_dist, out_sift, out_ctype, count);
}

// ---- per-atom accumulators of the contact loop (I:821-852, 923-934; U:182-221) ------------------------------
int arp_atom_accumulators(arp_ctx* c, uint16_t* out_sift4, int32_t* out_counts8) {
    if (!c || !out_sift4 || !out_counts8) return ARP_E_ARG;
    if (!c->contacts_valid) FAIL(c, ARP_E_ARG, "arp_atom_accumulators: no atom-contact results (call a launch first)");
    if (c->has_gid) FAIL(c, ARP_E_ARG, "arp_atom_accumulators: not available on a shard (contacts carry global ids)");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)std::max<int64_t>(c->n, 1);
    DevBuf<unsigned int> acc_s;
    DevBuf<int> acc_c;
    HIPCHK(c, acc_s.reserve(2 * n));
    HIPCHK(c, acc_c.reserve(8 * n));
    HIPCHK(c, hipMemsetAsync(acc_s.p, 0, 2 * n * sizeof(unsigned int), c->stream));
    HIPCHK(c, hipMemsetAsync(acc_c.p, 0, 8 * n * sizeof(int), c->stream));
    if (c->n_contacts > 0) {
        hipLaunchKernelGGL(k_accumulate, dim3(nblocks(c->n_contacts, 256, 4096)), dim3(256), 0, c->stream, (long long)c->n_contacts,
                           c->out_i.p, c->out_j.p, c->out_s.p, c->out_ct.p, acc_s.p, acc_c.p);
        CHK(check_launch(c, "k_accumulate"));
    }
    std::vector<unsigned int> hs(2 * n);
    int rc = download(c, hs.data(), acc_s.p, 2 * (size_t)c->n);
    if (rc == ARP_OK) rc = download(c, out_counts8, acc_c.p, 8 * (size_t)c->n);
    acc_s.release();
    acc_c.release();
    CHK(rc);
    for (int64_t a = 0; a < c->n; ++a) {   // {all, inter_only, intra_only, water_only}
        out_sift4[4 * a] = (uint16_t)(hs[2 * a] & 0x7FFF);
        out_sift4[4 * a + 1] = (uint16_t)((hs[2 * a] >> 16) & 0x7FFF);
        out_sift4[4 * a + 2] = (uint16_t)(hs[2 * a + 1] & 0x7FFF);
        out_sift4[4 * a + 3] = (uint16_t)((hs[2 * a + 1] >> 16) & 0x7FFF);
    }
    return ARP_OK;
}

int arp_atom_integer_sifts(arp_ctx* c, uint8_t* out_isift) {
    if (!c || !out_isift) return ARP_E_ARG;
    if (!c->contacts_valid) FAIL(c, ARP_E_ARG, "arp_atom_integer_sifts: no atom-contact results (call a launch first)");
    if (c->has_gid) FAIL(c, ARP_E_ARG, "arp_atom_integer_sifts: not available on a shard (contacts carry global ids)");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n4 = 4 * (size_t)std::max<int64_t>(c->n, 1);
    DevBuf<u64> last_rank;
    DevBuf<unsigned int> before, last_sift;
    DevBuf<uint8_t> out;
    HIPCHK(c, last_rank.reserve(n4));
    HIPCHK(c, before.reserve(n4));
    HIPCHK(c, last_sift.reserve(n4));
    HIPCHK(c, out.reserve(15 * n4));
    HIPCHK(c, hipMemsetAsync(last_rank.p, 0, n4 * sizeof(u64), c->stream));
    HIPCHK(c, hipMemsetAsync(before.p, 0, n4 * sizeof(unsigned int), c->stream));
    HIPCHK(c, hipMemsetAsync(last_sift.p, 0, n4 * sizeof(unsigned int), c->stream));
    if (c->n_contacts > 0) {
        const dim3 grid(nblocks(c->n_contacts, 256, 4096));
        hipLaunchKernelGGL(k_isift_last, grid, dim3(256), 0, c->stream, (long long)c->n_contacts, c->out_i.p, c->out_j.p,
                           c->out_ct.p, last_rank.p);
        hipLaunchKernelGGL(k_isift_fill, grid, dim3(256), 0, c->stream, (long long)c->n_contacts, c->out_i.p, c->out_j.p,
                           c->out_s.p, c->out_ct.p, last_rank.p, before.p, last_sift.p);
        CHK(check_launch(c, "k_isift_fill"));
    }
    hipLaunchKernelGGL(k_isift_compose, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, c->stream, (long long)n4, before.p,
                       last_sift.p, out.p);
    CHK(check_launch(c, "k_isift_compose"));
    int rc = download(c, out_isift, out.p, 60 * (size_t)c->n);
    last_rank.release();
    before.release();
    last_sift.release();
    out.release();
    return rc;
}

// ---- ring / amide contacts: launch (results stay in HBM) + fetch ----------------------------------
// Which loops take the list path when called alone (measured on BASELINE configs[4], 10 k rings + 10 k amides, HIP events):
// plane-plane 22.5 -> 18.6 us from the list; atom-plane 37 -> 63 us (its list has 60 candidates per ring: the walk, which tests the
// atoms of a ring's cells as it meets them, does less); the two amide loops 19 - 20 us either way (a chain of five dependent round
// trips whatever evaluates 1.5 k records).  ARP_BAG_LISTS: bit k = loop k from its list (default 2: plane-plane), 0 = every loop by
// its grid walk as through round 5, 15 = all four from the lists (the way a whole pass evaluates them).
static bool bag_walk(int kind = 1) { static const int v = env_int("ARP_BAG_LISTS", 2); return ((v >> kind) & 1) == 0; }
int arp_atom_plane_launch(arp_ctx* c, int64_t* count) {
    if (!c) return ARP_E_ARG;
    if (!bag_walk(0)) return bag_launch_lists(c, c->bag_ap, C_AP, true, false, 0, count);
    return bag_launch(c, c->bag_ap, C_AP, true, false, enqueue_atom_plane, count);
}
int arp_plane_plane_launch(arp_ctx* c, int64_t* count) {
    if (!c) return ARP_E_ARG;
    if (!bag_walk(1)) return bag_launch_lists(c, c->bag_pp, C_PP, true, false, 1, count);
    return bag_launch(c, c->bag_pp, C_PP, true, false, enqueue_plane_plane, count);
}
int arp_group_group_launch(arp_ctx* c, int64_t* count) {
    if (!c) return ARP_E_ARG;
    if (!bag_walk(2)) return bag_launch_lists(c, c->bag_gg, C_GG, false, true, 2, count);
    return bag_launch(c, c->bag_gg, C_GG, false, true, enqueue_group_group, count);
}
int arp_group_plane_launch(arp_ctx* c, int64_t* count) {
    if (!c) return ARP_E_ARG;
    if (!bag_walk(3)) return bag_launch_lists(c, c->bag_gp, C_GP, true, false, 3, count);
    return bag_launch(c, c->bag_gp, C_GP, true, false, enqueue_group_plane, count);
}

#define FETCH_PROLOGUE(bag, what)                                                                  \
    if (!c || !count) return ARP_E_ARG;                                                           \
    if (!(bag).valid) FAIL(c, ARP_E_ARG, what ": no launch results");                             \
    HIPCHK(c, hipSetDevice(c->device));                                                           \
    *count = (bag).count;                                                                         \
    if ((bag).count > cap) FAIL(c, ARP_E_CAPACITY, what ": output buffer too small");             \
    const size_t m = (size_t)(bag).count;

namespace {
int stage_bags(arp_ctx* c) {
    Bag* bags[4] = {&c->bag_ap, &c->bag_pp, &c->bag_gg, &c->bag_gp};
    bool stale = false;
    for (Bag* b : bags) stale = stale || (b->valid && b->staged_version != b->version);
    if (!stale) return ARP_OK;
    PackTable t;
    t.n = 0;
    size_t total = 0;
    for (Bag* b : bags) {
        if (!b->valid) continue;
        const uint8_t* ptr[12] = {(const uint8_t*)b->a.p, (const uint8_t*)b->b.p, (const uint8_t*)b->d0.p, (const uint8_t*)b->d1.p,
                                  (const uint8_t*)b->d2.p, (const uint8_t*)b->d3.p, (const uint8_t*)b->f0.p, (const uint8_t*)b->f1.p,
                                  (const uint8_t*)b->f2.p, b->u0.p, b->u1.p, b->u2.p};
        const size_t es[12] = {4, 4, 8, 8, 8, 8, 4, 4, 4, 1, 1, 1};
        for (int k = 0; k < 12; ++k) {
            b->stage_off[k] = (uint32_t)total;
            if (!ptr[k] || b->count == 0) continue;
            const size_t bytes = (size_t)b->count * es[k];
            t.s[t.n++] = PackSeg{ptr[k], (uint32_t)total, (uint32_t)bytes, nullptr, 0u};
            total = (total + bytes + 15) & ~(size_t)15;
        }
    }
    if (total > BAG_STAGE_MAX) {   // big bags (configs[4]): array by array, as before
        for (Bag* b : bags) b->staged_version = 0;
        return ARP_OK;
    }
    if (total > 0) {
        if (c->bag_stage_cap < total) {
            if (c->bag_stage) (void)hipHostFree(c->bag_stage);
            c->bag_stage = nullptr;
            c->bag_stage_cap = 0;
            const size_t want = total + total / 2 + 4096;
            if (hipHostMalloc((void**)&c->bag_stage, want, hipHostMallocDefault) != hipSuccess) { c->bag_stage = nullptr; return ARP_OK; }
            c->bag_stage_cap = want;
        }
        HIPCHK(c, c->bag_pack.reserve(total));
        hipLaunchKernelGGL(k_pack_segments, pack_grid(t), dim3(256), 0, c->stream, t, c->bag_pack.p);
        CHK(check_launch(c, "k_pack_segments"));
        HIPCHK(c, hipMemcpyAsync(c->bag_stage, c->bag_pack.p, total, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    } else if (!c->bag_stage) {    // nothing to copy, but "staged" needs a buffer to point into
        if (hipHostMalloc((void**)&c->bag_stage, 4096, hipHostMallocDefault) != hipSuccess) { c->bag_stage = nullptr; return ARP_OK; }
        c->bag_stage_cap = 4096;
    }
    for (Bag* b : bags)
        if (b->valid) b->staged_version = b->version;
    return ARP_OK;
}
}  // namespace

int arp_atom_plane_fetch(arp_ctx* c, int64_t cap, int32_t* out_atom, int32_t* out_ring, double* out_dist, double* out_theta,
                         uint8_t* out_mask, uint8_t* out_ctype, int64_t* count) {
    FETCH_PROLOGUE(c->bag_ap, "arp_atom_plane_fetch")
    Bag& b = c->bag_ap;
    CHK(stage_bags(c));
    CHK(bag_download(c, b, out_atom, b.a, 0, m)); CHK(bag_download(c, b, out_ring, b.b, 1, m)); CHK(bag_download(c, b, out_dist, b.d0, 2, m));
    CHK(bag_download(c, b, out_theta, b.d1, 3, m)); CHK(bag_download(c, b, out_mask, b.u0, 9, m)); CHK(bag_download(c, b, out_ctype, b.u1, 10, m));
    if (!bag_is_staged(c, b)) HIPCHK(c, hipStreamSynchronize(c->stream));
    return ARP_OK;
}
int arp_plane_plane_fetch(arp_ctx* c, int64_t cap, int32_t* out_bgn, int32_t* out_end, double* out_dist, double* out_dihedral,
                          double* out_theta_bgn, double* out_theta_end, uint8_t* out_type1, uint8_t* out_type2,
                          uint8_t* out_ctype, int64_t* count) {
    FETCH_PROLOGUE(c->bag_pp, "arp_plane_plane_fetch")
    Bag& b = c->bag_pp;
    CHK(stage_bags(c));
    CHK(bag_download(c, b, out_bgn, b.a, 0, m)); CHK(bag_download(c, b, out_end, b.b, 1, m)); CHK(bag_download(c, b, out_dist, b.d0, 2, m));
    CHK(bag_download(c, b, out_dihedral, b.d1, 3, m)); CHK(bag_download(c, b, out_theta_bgn, b.d2, 4, m)); CHK(bag_download(c, b, out_theta_end, b.d3, 5, m));
    CHK(bag_download(c, b, out_type1, b.u0, 9, m)); CHK(bag_download(c, b, out_type2, b.u1, 10, m)); CHK(bag_download(c, b, out_ctype, b.u2, 11, m));
    if (!bag_is_staged(c, b)) HIPCHK(c, hipStreamSynchronize(c->stream));
    return ARP_OK;
}
int arp_group_group_fetch(arp_ctx* c, int64_t cap, int32_t* out_bgn, int32_t* out_end, float* out_dist, float* out_dihedral,
                          float* out_theta, uint8_t* out_ctype, int64_t* count) {
    FETCH_PROLOGUE(c->bag_gg, "arp_group_group_fetch")
    Bag& b = c->bag_gg;
    CHK(stage_bags(c));
    CHK(bag_download(c, b, out_bgn, b.a, 0, m)); CHK(bag_download(c, b, out_end, b.b, 1, m)); CHK(bag_download(c, b, out_dist, b.f0, 6, m));
    CHK(bag_download(c, b, out_dihedral, b.f1, 7, m)); CHK(bag_download(c, b, out_theta, b.f2, 8, m)); CHK(bag_download(c, b, out_ctype, b.u0, 9, m));
    if (!bag_is_staged(c, b)) HIPCHK(c, hipStreamSynchronize(c->stream));
    return ARP_OK;
}
int arp_group_plane_fetch(arp_ctx* c, int64_t cap, int32_t* out_amide, int32_t* out_ring, double* out_dist, double* out_dihedral,
                          double* out_theta, uint8_t* out_ctype, int64_t* count) {
    FETCH_PROLOGUE(c->bag_gp, "arp_group_plane_fetch")
    Bag& b = c->bag_gp;
    CHK(stage_bags(c));
    CHK(bag_download(c, b, out_amide, b.a, 0, m)); CHK(bag_download(c, b, out_ring, b.b, 1, m)); CHK(bag_download(c, b, out_dist, b.d0, 2, m));
    CHK(bag_download(c, b, out_dihedral, b.d1, 3, m)); CHK(bag_download(c, b, out_theta, b.d2, 4, m)); CHK(bag_download(c, b, out_ctype, b.u0, 9, m));
    if (!bag_is_staged(c, b)) HIPCHK(c, hipStreamSynchronize(c->stream));
    return ARP_OK;
}

// launch + fetch.  A too-small caller buffer returns ARP_E_CAPACITY with the required count.
int arp_atom_plane(arp_ctx* c, int64_t cap, int32_t* out_atom, int32_t* out_ring, double* out_dist, double* out_theta,
                   uint8_t* out_mask, uint8_t* out_ctype, int64_t* count) {
    if (!c || !count || cap < 0) return ARP_E_ARG;
    CHK(arp_atom_plane_launch(c, count));
    return arp_atom_plane_fetch(c, cap, out_atom, out_ring, out_dist, out_theta, out_mask, out_ctype, count);
}
int arp_plane_plane(arp_ctx* c, int64_t cap, int32_t* out_bgn, int32_t* out_end, double* out_dist, double* out_dihedral,
                    double* out_theta_bgn, double* out_theta_end, uint8_t* out_type1, uint8_t* out_type2, uint8_t* out_ctype,
                    int64_t* count) {
    if (!c || !count || cap < 0) return ARP_E_ARG;
    CHK(arp_plane_plane_launch(c, count));
    return arp_plane_plane_fetch(c, cap, out_bgn, out_end, out_dist, out_dihedral, out_theta_bgn, out_theta_end, out_type1,
                                 out_type2, out_ctype, count);
}
int arp_group_group(arp_ctx* c, int64_t cap, int32_t* out_bgn, int32_t* out_end, float* out_dist, float* out_dihedral,
                    float* out_theta, uint8_t* out_ctype, int64_t* count) {
    if (!c || !count || cap < 0) return ARP_E_ARG;
    CHK(arp_group_group_launch(c, count));
    return arp_group_group_fetch(c, cap, out_bgn, out_end, out_dist, out_dihedral, out_theta, out_ctype, count);
}
int arp_group_plane(arp_ctx* c, int64_t cap, int32_t* out_amide, int32_t* out_ring, double* out_dist, double* out_dihedral,
                    double* out_theta, uint8_t* out_ctype, int64_t* count) {
    if (!c || !count || cap < 0) return ARP_E_ARG;
    CHK(arp_group_plane_launch(c, count));
    return arp_group_plane_fetch(c, cap, out_amide, out_ring, out_dist, out_dihedral, out_theta, out_ctype, count);
}

// ---- run_arpeggio (I:329-347): every stage enqueued back to back, ONE host synchronisation ---------
namespace {
// One pass = enqueue (everything back to back on the context's stream, no host synchronisation) + wait (the one host wait,
// capacity checks, a re-run if a buffer was too small).  arp_run_launch is the two in a row; arp_run_enqueue / arp_run_wait
// give them to the caller separately, so that ONE host thread keeps several contexts busy.
int run_pass_enqueue(arp_ctx* c, double cutoff, double vdw_comp, int include_sequence_adjacent, double expand_radius) {
    HIPCHK(c, hipSetDevice(c->device));
    CHK(default_selection(c));  // no selection uploaded for this structure: whole structure (I:1395)
    if (c->whole_structure && !c->sel_all)
        FAIL(c, ARP_E_ARG, "arp_run_launch: arp_set_whole_structure is on but the uploaded selection is partial");
    // every stage enqueued back to back (no host synchronisation, no allocation once the buffers are sized)
    // the counter block must be zero when a pass starts; a pass leaves it zeroed (k_publish_counters)
    auto ensure_zero = [&]() -> int {
        if (!c->ctr_zero_ok) HIPCHK(c, hipMemsetAsync(c->d_ctr, 0, sizeof(u64) * C_DEV_WORDS, c->stream));
        c->ctr_zero_ok = false;
        return ARP_OK;
    };
    auto enqueue_all = [&]() -> int {
        c->ctr_clean = true;
        struct Unclean { arp_ctx* c; ~Unclean() { c->ctr_clean = false; c->pub.expected = 0; c->fuse_sets = false; c->init_plus_in_bin = false; } } unclean{c};
        CHK(ensure_static(c, cutoff));       // (the spatial order of the columns is the one of this pass's cells)
        c->last_cutoff = cutoff;
        // The pass ends inside its last kernel (k_sift_planes): the last block to finish publishes the counters.
        static const int inkernel_publish = env_int("ARP_INKERNEL_PUBLISH", 1);
        c->pub = PublishArgs{c->d_ctr, c->h_ctr_pinned, 0, 0};
        if (inkernel_publish && !c->external_stream) {
            c->pub.expected = 1;
            c->pub.seq = ++c->publish_seq;
        }
        // _make_selection (I:1384-1424).  Whole-structure selection (the reference's default, I:1395 with no
        // selectors): selection_plus is the selection, nothing to search.  A small selection (ligand, binding site:
        // nsel <= SMALL_SEL_MAX, worth it while the N x S direct tests stay below ~1.7e7) gets selection_plus from a
        // direct test of every atom against the selected ones — one short kernel.  Anything else: the all-atom 6 A
        // grid and the expansion search.
        // (not for several structures in one pass: their coordinates overlap, only the grid keeps them apart)
        const bool small_sel = !c->sel_all && c->nsel > 0 && c->nsel <= SMALL_SEL_MAX && c->n > 0 && !c->has_home &&
                               c->n * c->nsel <= (int64_t)1 << 24 && c->batch_n == 0;
        if (c->sel_all || small_sel) {
            c->sel_made = true;
            HIPCHK(c, c->plus.reserve((size_t)std::max<int64_t>(c->n, 1)));
            c->contacts_valid = false;
            c->bag_ap.valid = c->bag_pp.valid = c->bag_gg.valid = c->bag_gp.valid = false;
            c->init_plus_in_bin = c->sel_all;      // selection_plus = selection: written by the contact grid's binning kernel
        } else {
            CHK(enqueue_expansion(c, expand_radius));                                // I:342 (I:1384-1424)
        }
        if (small_sel) {
            Prof p(c, SLOT_MARK);
            hipLaunchKernelGGL(k_expand_small, dim3(nblocks(c->n, 256, 1 << 22)), dim3(256), 0, c->stream, (int)c->n, c->xyz.p, c->sel_list.p,
                               (int)c->nsel, c->sel.p, expand_radius * expand_radius, c->plus.p, c->d_ctr + ctr_dev(C_STAT_MCAND));
            CHK(check_launch(c, "k_expand_small"));
        }
        // (ring / amide grids, built once per structure: with the candidate lists, in enqueue_contacts)
        // I:1413-1437 (residue / ring / amide sets) ride on the contact grid build, I:345-347 in three launches
        c->fuse_sets = true;
        CHK(enqueue_contacts(c, cutoff, vdw_comp, include_sequence_adjacent, true));
        if (c->pub.expected) return ARP_OK;                                         // the last kernel publishes (pass_end)
        return enqueue_counter_copy(c, 1);
    };
    const auto t0 = std::chrono::steady_clock::now();
    CHK(ensure_zero());
    CHK(enqueue_all());
    c->host_enqueue_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    return ARP_OK;
}

int run_pass_wait(arp_ctx* c, int64_t counts[5]) {
    HIPCHK(c, hipSetDevice(c->device));
    auto any_overflow = [&](bool grow) -> int {   // returns 1 when a buffer was too small (and regrows it if asked)
        int again = 0;
        if (finish_contacts(c)) { if (grow) CHK(grow_pairs(c)); again = 1; }
        if (finish_bag(c, c->bag_ap, C_AP)) { if (grow) CHK(grow_bag(c, c->bag_ap, C_AP, true, false)); again = 1; }
        if (finish_bag(c, c->bag_pp, C_PP)) { if (grow) CHK(grow_bag(c, c->bag_pp, C_PP, true, false)); again = 1; }
        if (finish_bag(c, c->bag_gg, C_GG)) { if (grow) CHK(grow_bag(c, c->bag_gg, C_GG, false, true)); again = 1; }
        if (finish_bag(c, c->bag_gp, C_GP)) { if (grow) CHK(grow_bag(c, c->bag_gp, C_GP, true, false)); again = 1; }
        const int lists = finish_plane_lists(c, grow);
        if (lists < 0) return lists;
        return again | lists;
    };
    for (int attempt = 0;; ++attempt) {
        const auto t1 = std::chrono::steady_clock::now();
        CHK(collect_counters(c));
        c->ctr_zero_ok = true;
        c->host_wait_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
        ++c->host_passes;
        collect_events(c);
        const int again = any_overflow(true);
        if (again < 0) return again;
        if (!again) break;
        if (attempt == 2) FAIL(c, ARP_E_CAPACITY, "arp_run_launch: result buffers could not be sized");
        CHK(run_pass_enqueue(c, c->pending_cutoff, c->pending_comp, c->pending_seq_adj, c->pending_expand));   // a buffer was too small: once more
    }
    c->stats[5] = (int64_t)c->h_ctr[C_MARK_CAND];
    c->stats[6] = (int64_t)c->h_ctr[C_MARK_ACC];
    if (counts) {
        counts[0] = c->n_contacts; counts[1] = c->bag_pp.count; counts[2] = c->bag_ap.count;
        counts[3] = c->bag_gg.count; counts[4] = c->bag_gp.count;
    }
    const int rc_dev = device_error(c);
    if (rc_dev == ARP_OK && c->sort_after_pass && c->contacts_valid && c->n_contacts > 0 && c->n_contacts < ((int64_t)1 << 31)) {
        // (room for the ring / amide bags behind the sorted columns, as arp_fetch_packed lays them out: at most 52 bytes a record
        // and twelve arrays a bag, each rounded up to 16 bytes)
        const size_t extra = 52 * (size_t)(c->bag_pp.count + c->bag_ap.count + c->bag_gg.count + c->bag_gp.count) + 4 * 12 * 16;
        CHK(sort_contacts(c, extra));
    }
    return rc_dev;
}
}  // namespace

int arp_run_enqueue(arp_ctx* c, double cutoff, double vdw_comp, int include_sequence_adjacent, double expand_radius) {
    if (!c || !(cutoff > 0) || !(expand_radius > 0)) return ARP_E_ARG;
    if (c->pass_pending) FAIL(c, ARP_E_ARG, "arp_run_enqueue: the previous pass has not been waited for (arp_run_wait)");
    c->pending_cutoff = cutoff; c->pending_comp = vdw_comp; c->pending_seq_adj = include_sequence_adjacent; c->pending_expand = expand_radius;
    CHK(run_pass_enqueue(c, cutoff, vdw_comp, include_sequence_adjacent, expand_radius));
    c->pass_pending = true;
    return ARP_OK;
}

int arp_run_wait(arp_ctx* c, int64_t counts[5]) {
    if (!c) return ARP_E_ARG;
    if (!c->pass_pending) FAIL(c, ARP_E_ARG, "arp_run_wait: no pass was enqueued (arp_run_enqueue)");
    c->pass_pending = false;
    return run_pass_wait(c, counts);
}

int arp_run_launch(arp_ctx* c, double cutoff, double vdw_comp, int include_sequence_adjacent, double expand_radius,
                   int64_t counts[5]) {
    CHK(arp_run_enqueue(c, cutoff, vdw_comp, include_sequence_adjacent, expand_radius));
    return arp_run_wait(c, counts);
}

// ---- staged run_arpeggio for sharded runs ------------------------------------------------------------
int arp_device_buffer(arp_ctx* c, int which, uint64_t* device_ptr, int64_t* bytes) {
    if (!c || !device_ptr || !bytes) return ARP_E_ARG;
    if (which == ARP_BUF_PLUS) {
        if (!c->plus.p) FAIL(c, ARP_E_ARG, "arp_device_buffer: selection_plus does not exist yet (run stage 0 first)");
        *device_ptr = (uint64_t)(uintptr_t)c->plus.p;
        *bytes = c->n;
    } else if (which == ARP_BUF_RES_SETS) {
        if (!c->res_sel.p) FAIL(c, ARP_E_ARG, "arp_device_buffer: residue sets do not exist yet (run stage 1 first)");
        *device_ptr = (uint64_t)(uintptr_t)c->res_sel.p;
        *bytes = 2 * std::max<int64_t>(c->nres, 1);
    } else FAIL(c, ARP_E_ARG, "arp_device_buffer: unknown buffer");
    return ARP_OK;
}

int arp_run_stage(arp_ctx* c, int stage, double cutoff, double vdw_comp, int include_sequence_adjacent, double expand_radius,
                  int64_t counts[5]) {
    if (!c || stage < 0 || stage > 2) return ARP_E_ARG;
    c->ctr_zero_ok = false;
    HIPCHK(c, hipSetDevice(c->device));
    if (stage == 0) {          // I:1384-1424 on the local atoms; exact for the atoms this rank owns
        if (!(expand_radius > 0)) return ARP_E_ARG;
        CHK(default_selection(c));
        HIPCHK(c, hipMemsetAsync(c->d_ctr, 0, sizeof(u64) * C_DEV_WORDS, c->stream));
        c->ctr_clean = true;
        int rc = enqueue_expansion(c, expand_radius);
        c->ctr_clean = false;
        CHK(rc);
        if (!c->external_stream && !c->comm) HIPCHK(c, hipStreamSynchronize(c->stream));   // (with a communicator the exchange follows on the same stream)
        return ARP_OK;
    }
    if (!c->sel_made) FAIL(c, ARP_E_ARG, "arp_run_stage: stage 0 has not run");
    if (stage == 1) {          // I:1413, 1431 residue sets from the (now globally correct) selection_plus bits
        const int n = (int)c->n;
        const size_t nres = (size_t)std::max<int64_t>(c->nres, 1);
        CHK(check_residue_ranges(c));
        HIPCHK(c, c->res_sel.reserve(2 * nres));
        HIPCHK(c, hipMemsetAsync(c->res_sel.p, 0, 2 * nres, c->stream));
        if (n > 0)
            hipLaunchKernelGGL(k_res_mark, dim3(nblocks(n, 256)), dim3(256), 0, c->stream, n, c->res_id.p, c->sel.p, c->plus.p,
                               c->res_sel.p, c->res_sel.p + nres);
        CHK(check_launch(c, "k_res_mark"));
        if (!c->external_stream && !c->comm) HIPCHK(c, hipStreamSynchronize(c->stream));   // (with a communicator the exchange follows on the same stream)
        return ARP_OK;
    }
    // stage 2: ring / amide sets from the (now globally reduced) residue sets, then every contact bag
    if (!(cutoff > 0)) return ARP_E_ARG;
    if (!c->res_sel.p) FAIL(c, ARP_E_ARG, "arp_run_stage: stage 1 has not run");
    for (int attempt = 0;; ++attempt) {
        const size_t nres = (size_t)std::max<int64_t>(c->nres, 1);
        // the expansion statistics of stage 0 live in the counter block: keep them, clear the rest
        HIPCHK(c, hipMemsetAsync(c->d_ctr, 0, sizeof(u64) * 5 * CTR_LINE, c->stream));                      // scalars, the four bags
        HIPCHK(c, hipMemset2DAsync(c->d_ctr + ctr_dev(C_STAT_CAND), sizeof(u64) * CTR_LINE, 0, 2 * sizeof(u64), STAT_SLOTS, c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_ctr + ctr_dev(C_SEG_PAIRS), 0, sizeof(u64) * PAIR_SEGS * CTR_LINE, c->stream));
        c->ctr_clean = true;
        struct Unclean { arp_ctx* c; ~Unclean() { c->ctr_clean = false; } } unclean{c};
        if (c->nring + c->namide > 0)
            hipLaunchKernelGGL(k_group_mask, dim3(nblocks(c->nring + c->namide, 256)), dim3(256), 0, c->stream, (int)c->nring,
                               (int)c->namide, c->ring_res.p, c->am_res.p, c->res_sel.p, c->res_sel.p + nres, c->ring_sel.p,
                               c->ring_plus.p, c->am_sel.p, c->am_plus.p);
        CHK(check_launch(c, "k_group_mask"));
        CHK(ensure_center_grids(c));
        CHK(enqueue_contacts(c, cutoff, vdw_comp, include_sequence_adjacent, true));
        CHK(enqueue_counter_copy(c));
        CHK(collect_counters(c));
        collect_events(c);
        int again = 0;
        if (finish_contacts(c)) { CHK(grow_pairs(c)); again = 1; }
        if (finish_bag(c, c->bag_ap, C_AP)) { CHK(grow_bag(c, c->bag_ap, C_AP, true, false)); again = 1; }
        if (finish_bag(c, c->bag_pp, C_PP)) { CHK(grow_bag(c, c->bag_pp, C_PP, true, false)); again = 1; }
        if (finish_bag(c, c->bag_gg, C_GG)) { CHK(grow_bag(c, c->bag_gg, C_GG, false, true)); again = 1; }
        if (finish_bag(c, c->bag_gp, C_GP)) { CHK(grow_bag(c, c->bag_gp, C_GP, true, false)); again = 1; }
        {
            const int lists = finish_plane_lists(c, true);
            if (lists < 0) return lists;
            again |= lists;
        }
        if (!again) break;
        if (attempt == 2) FAIL(c, ARP_E_CAPACITY, "arp_run_stage: result buffers could not be sized");
    }
    c->stats[5] = (int64_t)c->h_ctr[C_MARK_CAND];
    c->stats[6] = (int64_t)c->h_ctr[C_MARK_ACC];
    if (counts) {
        counts[0] = c->n_contacts; counts[1] = c->bag_pp.count; counts[2] = c->bag_ap.count;
        counts[3] = c->bag_gg.count; counts[4] = c->bag_gp.count;
    }
    return device_error(c);
}

// ---- measurement -------------------------------------------------------------------------------
int arp_get_stats(arp_ctx* c, int64_t stats[8]) {
    if (!c || !stats) return ARP_E_ARG;
    for (int k = 0; k < 8; ++k) stats[k] = c->stats[k];
    return ARP_OK;
}

int arp_set_profiling(arp_ctx* c, int enabled) {
    if (!c) return ARP_E_ARG;
    c->profiling = enabled != 0;   // launches bracketed by HIP events on their own streams
    return ARP_OK;
}

int arp_get_kernel_times(arp_ctx* c, double ms[8], int64_t launches[8], int reset) {
    if (!c || !ms || !launches) return ARP_E_ARG;
    for (int k = 0; k < NSLOT; ++k) { ms[k] = c->k_ms[k]; launches[k] = c->k_launches[k]; }
    if (reset) for (int k = 0; k < NSLOT; ++k) { c->k_ms[k] = 0; c->k_launches[k] = 0; }
    return ARP_OK;
}

// ---- geometric part of initialize() --------------------------------------------------
int arp_ring_geometry(arp_ctx* c, int64_t nring, const int32_t* ring_off, const int32_t* ring_idx, double* out_center,
                      double* out_normal) {
    if (!c || nring < 0 || (nring > 0 && (!ring_off || !out_center || !out_normal))) return ARP_E_ARG;
    if (nring == 0) return ARP_OK;
    if (!csr_ok(ring_off, nring)) FAIL(c, ARP_E_ARG, "arp_ring_geometry: offsets must start at 0 and never decrease");
    const int64_t m = ring_off[nring];
    if (m > 0 && !ring_idx) return ARP_E_ARG;
    for (int64_t r = 0; r < nring; ++r)
        if (ring_off[r + 1] - ring_off[r] < 3) FAIL(c, ARP_E_ARG, "arp_ring_geometry: a ring needs at least three atoms");
    for (int64_t k = 0; k < m; ++k)
        if (ring_idx[k] < 0 || ring_idx[k] >= c->n) FAIL(c, ARP_E_ARG, "arp_ring_geometry: atom index out of range");
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf<int> d_off, d_idx;
    DevBuf<double> d_c, d_n;
    int rc = upload(c, d_off, ring_off, (size_t)nring + 1);
    if (rc == ARP_OK) rc = upload(c, d_idx, ring_idx, (size_t)m);
    hipError_t e = d_c.reserve((size_t)nring * 3);
    if (e == hipSuccess) e = d_n.reserve((size_t)nring * 3);
    if (rc == ARP_OK && e == hipSuccess) {
        hipLaunchKernelGGL(k_ring_geometry, dim3(nblocks(nring, 256)), dim3(256), 0, c->stream, (int)nring, d_off.p, d_idx.p,
                           c->xyz.p, d_c.p, d_n.p);
        rc = check_launch(c, "k_ring_geometry");
        if (rc == ARP_OK) rc = download(c, out_center, d_c.p, (size_t)nring * 3);
        if (rc == ARP_OK) rc = download(c, out_normal, d_n.p, (size_t)nring * 3);
    }
    d_off.release(); d_idx.release(); d_c.release(); d_n.release();
    if (e != hipSuccess) FAIL(c, ARP_E_NOMEM, "arp_ring_geometry: out of device memory");
    return rc;
}

int arp_amide_geometry(arp_ctx* c, int64_t namide, const int32_t* amide_atoms, float* out_center, float* out_normal) {
    if (!c || namide < 0 || (namide > 0 && (!amide_atoms || !out_center || !out_normal))) return ARP_E_ARG;
    if (namide == 0) return ARP_OK;
    for (int64_t k = 0; k < 4 * namide; ++k)
        if ((k & 3) != 3 && (amide_atoms[k] < 0 || amide_atoms[k] >= c->n))   // N, C, O are read; the fourth atom is not
            FAIL(c, ARP_E_ARG, "arp_amide_geometry: atom index out of range");
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf<int> d_at;
    DevBuf<float> d_c, d_n;
    int rc = upload(c, d_at, amide_atoms, (size_t)namide * 4);
    hipError_t e = d_c.reserve((size_t)namide * 3);
    if (e == hipSuccess) e = d_n.reserve((size_t)namide * 3);
    if (rc == ARP_OK && e == hipSuccess) {
        hipLaunchKernelGGL(k_amide_geometry, dim3(nblocks(namide, 256)), dim3(256), 0, c->stream, (int)namide, d_at.p, c->xyz.p,
                           d_c.p, d_n.p);
        rc = check_launch(c, "k_amide_geometry");
        if (rc == ARP_OK) rc = download(c, out_center, d_c.p, (size_t)namide * 3);
        if (rc == ARP_OK) rc = download(c, out_normal, d_n.p, (size_t)namide * 3);
    }
    d_at.release(); d_c.release(); d_n.release();
    if (e != hipSuccess) FAIL(c, ARP_E_NOMEM, "arp_amide_geometry: out of device memory");
    return rc;
}

int arp_ring_residues(arp_ctx* c, int64_t nring, const double* center, int32_t* out_ring_res, double* out_shortest) {
    if (!c || nring < 0 || (nring > 0 && (!center || !out_ring_res))) return ARP_E_ARG;
    if (nring == 0) return ARP_OK;
    if (!all_finite(center, 3 * nring)) FAIL(c, ARP_E_ARG, "arp_ring_residues: non-finite ring centre");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n == 0) {
        for (int64_t r = 0; r < nring; ++r) { out_ring_res[r] = -1; if (out_shortest) out_shortest[r] = -1.0; }
        return ARP_OK;
    }
    // the all-atom grid (hydrogens included, I:1455); its cells (>= 6 A) cover the 3 A query with the 27-cell stencil
    if (!(c->all_grid_current && c->all_grid.valid && c->all_grid.radius == 6.0)) CHK(build_all_grid(c, 6.0));
    DevBuf<double> d_c, d_d;
    DevBuf<int> d_r;
    int rc = upload(c, d_c, center, (size_t)nring * 3);
    hipError_t e = d_r.reserve((size_t)nring);
    if (e == hipSuccess) e = d_d.reserve((size_t)nring);
    if (rc == ARP_OK && e == hipSuccess) {
        hipLaunchKernelGGL(k_ring_residue, dim3(nblocks(nring * 64, 256, 4096)), dim3(256), 0, c->stream, c->all_grid.d,
                           c->all_grid.start.p, c->a_xyzm.p, c->a_aux.p, (int)nring, d_c.p, d_r.p, d_d.p);
        rc = check_launch(c, "k_ring_residue");
        if (rc == ARP_OK) rc = download(c, out_ring_res, d_r.p, (size_t)nring);
        if (rc == ARP_OK && out_shortest) rc = download(c, out_shortest, d_d.p, (size_t)nring);
    }
    d_c.release(); d_r.release(); d_d.release();
    if (e != hipSuccess) FAIL(c, ARP_E_NOMEM, "arp_ring_residues: out of device memory");
    return rc;
}

int arp_search(arp_ctx* c, double radius, int64_t ncenters, const double* centers, int64_t cap, int32_t* out_center, int32_t* out_atom,
               int64_t* count) {
    if (!c || !count || ncenters < 0 || cap < 0 || !(radius > 0) || (ncenters > 0 && !centers)) return ARP_E_ARG;
    *count = 0;
    if (ncenters == 0 || c->n == 0) return ARP_OK;
    if (ncenters > 0x7FFFFFF0LL) FAIL(c, ARP_E_ARG, "arp_search: too many centres");
    if (!all_finite(centers, 3 * ncenters)) FAIL(c, ARP_E_ARG, "arp_search: non-finite centre");
    HIPCHK(c, hipSetDevice(c->device));
    if (!(c->all_grid_current && c->all_grid.valid && c->all_grid.radius >= radius)) CHK(build_all_grid(c, std::max(radius, 6.0)));
    DevBuf<double> d_c;
    DevBuf<int2> d_out;
    DevBuf<u64> d_n;
    int rc = upload(c, d_c, centers, (size_t)ncenters * 3);
    hipError_t e = d_out.reserve((size_t)std::max<int64_t>(cap, 1));
    if (e == hipSuccess) e = d_n.reserve(1);
    u64 found = 0;
    std::vector<int2> tmp;
    if (rc == ARP_OK && e == hipSuccess) {
        e = hipMemsetAsync(d_n.p, 0, sizeof(u64), c->stream);
        hipLaunchKernelGGL(k_center_search, dim3(nblocks(ncenters * 64, 256, 4096)), dim3(256), 0, c->stream, c->all_grid.d, c->all_grid.start.p,
                           c->a_xyzm.p, c->a_aux.p, (int)ncenters, d_c.p, radius * radius, d_out.p, (u64)cap, d_n.p);
        rc = check_launch(c, "k_center_search");
        if (rc == ARP_OK) rc = download(c, &found, d_n.p, 1);
        if (rc == ARP_OK && found <= (u64)cap && found > 0) {
            tmp.resize((size_t)found);
            rc = download(c, tmp.data(), d_out.p, (size_t)found);
        }
    }
    d_c.release(); d_out.release(); d_n.release();
    if (e != hipSuccess) FAIL(c, ARP_E_NOMEM, "arp_search: out of device memory");
    if (rc != ARP_OK) return rc;
    *count = (int64_t)found;
    if (found > (u64)cap) FAIL(c, ARP_E_CAPACITY, "arp_search: output buffer too small");
    std::sort(tmp.begin(), tmp.end(), [](const int2& a, const int2& b) { return a.x != b.x ? a.x < b.x : a.y < b.y; });
    for (size_t k = 0; k < tmp.size(); ++k) {
        if (out_center) out_center[k] = tmp[k].x;
        if (out_atom) out_atom[k] = tmp[k].y;
    }
    return ARP_OK;
}

// Page-locked host memory for result buffers (and inputs): copies to / from it are DMA transfers at PCIe speed,
// pageable memory goes through the runtime's staging buffer at a fraction of that.
int arp_host_alloc(uint64_t bytes, void** out) {
    if (!out) return ARP_E_ARG;
    *out = nullptr;
    if (bytes == 0) return ARP_OK;
    return hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault) == hipSuccess ? ARP_OK : ARP_E_NOMEM;
}
int arp_host_free(void* p) {
    if (!p) return ARP_OK;
    return hipHostFree(p) == hipSuccess ? ARP_OK : ARP_E_HIP;
}

int arp_set_batch(arp_ctx* c, int64_t nstruct, const int64_t* atom_off, const int64_t* ring_off, const int64_t* amide_off,
                  const double* boxes) {
    if (!c) return ARP_E_ARG;
    CHK(join_upload_lists(c));
    if (nstruct == 0) {   // back to one structure
        batch_reset(c);
        c->static_dirty = true; c->lists_from_upload = false; c->lists_dirty = true;
        c->atom_grid.valid = false; c->all_grid_current = false; c->ring_grid.valid = false; c->amide_grid.valid = false;
        return ARP_OK;
    }
    if (nstruct < 0 || !atom_off || !ring_off || !amide_off || !boxes) FAIL(c, ARP_E_ARG, "arp_set_batch: bad input");
    if (c->has_home || c->has_gid || c->shard_resident) FAIL(c, ARP_E_ARG, "arp_set_batch: not for a shard of a distributed structure");
    auto check = [&](const int64_t* off, int64_t total) {
        if (off[0] != 0 || off[nstruct] != total) return false;
        for (int64_t s_ = 0; s_ < nstruct; ++s_)
            if (off[s_ + 1] < off[s_]) return false;
        return true;
    };
    if (!check(atom_off, c->n) || !check(ring_off, c->nring) || !check(amide_off, c->namide))
        FAIL(c, ARP_E_ARG, "arp_set_batch: offsets must start at 0, never decrease and end at the resident atom / ring / amide counts");
    for (int64_t k = 0; k < nstruct; ++k)
        for (int a = 0; a < 3; ++a) {
            const double lo = boxes[6 * k + a], hi = boxes[6 * k + 3 + a];
            if (!std::isfinite(lo) || !std::isfinite(hi) || hi < lo) FAIL(c, ARP_E_ARG, "arp_set_batch: a box is not finite or has hi < lo");
        }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    batch_reset(c);
    c->batch_atom_off.assign(atom_off, atom_off + nstruct + 1);
    c->batch_ring_off.assign(ring_off, ring_off + nstruct + 1);
    c->batch_amide_off.assign(amide_off, amide_off + nstruct + 1);
    c->batch_box.assign(boxes, boxes + 6 * nstruct);
    // structure of every atom / ring / amide: made on the device from the three offset tables (building and uploading three
    // arrays of that length on the host was 160 us of a 64-structure batch)
    {
        const size_t m = (size_t)nstruct + 1;
        std::vector<long long> h(3 * m);
        for (size_t k = 0; k < m; ++k) { h[k] = atom_off[k]; h[m + k] = ring_off[k]; h[2 * m + k] = amide_off[k]; }
        CHK(upload(c, c->batch_off_dev, h.data(), h.size()));        // (waits: the staging vector ends here)
        const int64_t tot[3] = {c->n, c->nring, c->namide};
        DevBuf<int>* dst[3] = {&c->sid_atom, &c->sid_ring, &c->sid_amide};
        for (int k = 0; k < 3; ++k) {
            HIPCHK(c, dst[k]->reserve((size_t)std::max<int64_t>(tot[k], 1)));
            if (tot[k] > 0)
                hipLaunchKernelGGL(k_fill_sid, dim3(nblocks(tot[k], 256)), dim3(256), 0, c->stream, c->batch_off_dev.p + k * m, (int)nstruct, dst[k]->p, (long long)tot[k]);
        }
        CHK(check_launch(c, "k_fill_sid"));
    }
    c->batch_n = nstruct;
    c->static_dirty = true; c->lists_from_upload = false; c->lists_dirty = true;
    c->contacts_valid = false;
    c->atom_grid.valid = false; c->all_grid_current = false; c->ring_grid.valid = false; c->amide_grid.valid = false;
    c->bag_ap.valid = c->bag_pp.valid = c->bag_gg.valid = c->bag_gp.valid = false;
    return ARP_OK;
}

// ---- exchange between the shards of a distributed structure (RCCL behind the C ABI) -----------------------------------
#define NCCLCHK(ctx, expr)                                                                              \
    do {                                                                                                \
        ncclResult_t r_ = (expr);                                                                       \
        if (r_ != ncclSuccess) {                                                                        \
            (ctx)->err = std::string(#expr) + ": " + rccl().GetErrorString(r_);                         \
            return ARP_E_HIP;                                                                           \
        }                                                                                               \
    } while (0)

int arp_comm_unique_id(uint8_t* out, uint64_t cap) {
    if (!out || cap < NCCL_UNIQUE_ID_BYTES) return ARP_E_ARG;
    if (!rccl().load()) { set_create_error(rccl().error); return ARP_E_HIP; }
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) { set_create_error("ncclGetUniqueId failed"); return ARP_E_HIP; }
    memcpy(out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return ARP_OK;
}

int arp_comm_init(arp_ctx* c, int rank, int world, const uint8_t* unique_id) {
    if (!c || !unique_id || world < 1 || rank < 0 || rank >= world) return ARP_E_ARG;
    if (c->comm) FAIL(c, ARP_E_ARG, "arp_comm_init: the context has a communicator already (arp_comm_destroy first)");
    if (!rccl().load()) FAIL(c, ARP_E_HIP, rccl().error);
    HIPCHK(c, hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
    NCCLCHK(c, rccl().CommInitRank(&c->comm, world, id, rank));
    c->comm_rank = rank;
    c->comm_world = world;
    HIPCHK(c, c->comm_words.reserve(4));
    return ARP_OK;
}

int arp_comm_destroy(arp_ctx* c) {
    if (!c) return ARP_E_ARG;
    if (c->comm) {
        (void)hipStreamSynchronize(c->stream);
        (void)rccl().CommDestroy(c->comm);
        c->comm = nullptr;
    }
    c->comm_rank = 0;
    c->comm_world = 1;
    return ARP_OK;
}

int arp_comm_info(arp_ctx* c, int* rank, int* world) {
    if (!c) return ARP_E_ARG;
    if (rank) *rank = c->comm_rank;
    if (world) *world = c->comm_world;
    if (!c->comm) return ARP_E_ARG;
    // what RCCL itself says about the communicator (the ranks it connected), where the library exports the queries
    if (world && rccl().CommCount) NCCLCHK(c, rccl().CommCount(c->comm, world));
    if (rank && rccl().CommUserRank) NCCLCHK(c, rccl().CommUserRank(c->comm, rank));
    return ARP_OK;
}

int arp_shard_exchange_faces(arp_ctx* c, uint64_t left_ptr, uint64_t left_bytes, uint64_t right_ptr, uint64_t right_bytes, uint64_t received[4]) {
    if (!c || !received) return ARP_E_ARG;
    if (!c->comm) FAIL(c, ARP_E_ARG, "arp_shard_exchange_faces: no communicator (arp_comm_init)");
    HIPCHK(c, hipSetDevice(c->device));
    const int peer[2] = {c->comm_rank - 1, c->comm_rank + 1};
    const bool has[2] = {peer[0] >= 0, peer[1] < c->comm_world};
    const void* out_ptr[2] = {(const void*)(uintptr_t)left_ptr, (const void*)(uintptr_t)right_ptr};
    const unsigned long long out_bytes[2] = {has[0] ? left_bytes : 0ull, has[1] ? right_bytes : 0ull};
    // sizes first: one 64-bit word each way (device words, filled / read through the stream)
    unsigned long long words[4] = {out_bytes[0], out_bytes[1], 0ull, 0ull};
    HIPCHK(c, hipMemcpyAsync(c->comm_words.p, words, sizeof(words), hipMemcpyHostToDevice, c->stream));
    NCCLCHK(c, rccl().GroupStart());
    for (int s_ = 0; s_ < 2; ++s_)
        if (has[s_]) {
            NCCLCHK(c, rccl().Send(c->comm_words.p + s_, 8, ncclUint8, peer[s_], c->comm, c->stream));
            NCCLCHK(c, rccl().Recv(c->comm_words.p + 2 + s_, 8, ncclUint8, peer[s_], c->comm, c->stream));
        }
    NCCLCHK(c, rccl().GroupEnd());
    HIPCHK(c, hipMemcpyAsync(words, c->comm_words.p, sizeof(words), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int s_ = 0; s_ < 2; ++s_) HIPCHK(c, c->comm_recv[s_].reserve((size_t)std::max<unsigned long long>(words[2 + s_], 1)));
    NCCLCHK(c, rccl().GroupStart());
    for (int s_ = 0; s_ < 2; ++s_)
        if (has[s_]) {
            if (out_bytes[s_]) NCCLCHK(c, rccl().Send(out_ptr[s_], (size_t)out_bytes[s_], ncclUint8, peer[s_], c->comm, c->stream));
            if (words[2 + s_]) NCCLCHK(c, rccl().Recv(c->comm_recv[s_].p, (size_t)words[2 + s_], ncclUint8, peer[s_], c->comm, c->stream));
        }
    NCCLCHK(c, rccl().GroupEnd());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int s_ = 0; s_ < 2; ++s_) {
        received[2 * s_] = (has[s_] && words[2 + s_]) ? (uint64_t)(uintptr_t)c->comm_recv[s_].p : 0;
        received[2 * s_ + 1] = has[s_] ? words[2 + s_] : 0;
    }
    return ARP_OK;
}

int arp_shard_set_exchange_lists(arp_ctx* c, const int32_t* send_left, int64_t n_send_left, const int32_t* send_right, int64_t n_send_right,
                                 const int32_t* recv_left, int64_t n_recv_left, const int32_t* recv_right, int64_t n_recv_right) {
    if (!c || n_send_left < 0 || n_send_right < 0 || n_recv_left < 0 || n_recv_right < 0) return ARP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int32_t* src[4] = {send_left, send_right, recv_left, recv_right};
    const int64_t cnt[4] = {n_send_left, n_send_right, n_recv_left, n_recv_right};
    for (int k = 0; k < 4; ++k) {
        if (cnt[k] > 0 && !src[k]) return ARP_E_ARG;
        for (int64_t i = 0; i < cnt[k]; ++i)
            if (src[k][i] < 0 || src[k][i] >= c->n) FAIL(c, ARP_E_ARG, "arp_shard_set_exchange_lists: atom index out of range");
    }
    for (int s_ = 0; s_ < 2; ++s_) {
        CHK(upload_async(c, c->xl_send[s_], src[s_], (size_t)cnt[s_]));
        CHK(upload_async(c, c->xl_recv[s_], src[2 + s_], (size_t)cnt[2 + s_]));
        HIPCHK(c, c->xl_send_buf[s_].reserve((size_t)std::max<int64_t>(cnt[s_], 1)));
        HIPCHK(c, c->xl_recv_buf[s_].reserve((size_t)std::max<int64_t>(cnt[2 + s_], 1)));
        c->xl_nsend[s_] = cnt[s_];
        c->xl_nrecv[s_] = cnt[2 + s_];
    }
    return upload_done(c);
}

int arp_shard_exchange_plus(arp_ctx* c) {
    if (!c) return ARP_E_ARG;
    if (!c->comm) FAIL(c, ARP_E_ARG, "arp_shard_exchange_plus: no communicator (arp_comm_init)");
    if (!c->plus.p) FAIL(c, ARP_E_ARG, "arp_shard_exchange_plus: selection_plus does not exist yet (run stage 0 first)");
    HIPCHK(c, hipSetDevice(c->device));
    const int peer[2] = {c->comm_rank - 1, c->comm_rank + 1};
    const bool has[2] = {peer[0] >= 0, peer[1] < c->comm_world};
    for (int s_ = 0; s_ < 2; ++s_)
        if (has[s_] && c->xl_nsend[s_] > 0)
            hipLaunchKernelGGL(k_gather_u8, dim3(nblocks(c->xl_nsend[s_], 256)), dim3(256), 0, c->stream, (int)c->xl_nsend[s_], c->xl_send[s_].p, c->plus.p, c->xl_send_buf[s_].p);
    CHK(check_launch(c, "k_gather_u8"));
    NCCLCHK(c, rccl().GroupStart());
    for (int s_ = 0; s_ < 2; ++s_)
        if (has[s_]) {
            if (c->xl_nsend[s_] > 0) NCCLCHK(c, rccl().Send(c->xl_send_buf[s_].p, (size_t)c->xl_nsend[s_], ncclUint8, peer[s_], c->comm, c->stream));
            if (c->xl_nrecv[s_] > 0) NCCLCHK(c, rccl().Recv(c->xl_recv_buf[s_].p, (size_t)c->xl_nrecv[s_], ncclUint8, peer[s_], c->comm, c->stream));
        }
    NCCLCHK(c, rccl().GroupEnd());
    for (int s_ = 0; s_ < 2; ++s_)
        if (has[s_] && c->xl_nrecv[s_] > 0)
            hipLaunchKernelGGL(k_scatter_u8, dim3(nblocks(c->xl_nrecv[s_], 256)), dim3(256), 0, c->stream, (int)c->xl_nrecv[s_], c->xl_recv[s_].p, c->xl_recv_buf[s_].p, c->plus.p);
    return check_launch(c, "k_scatter_u8");
}

int arp_shard_reduce_residue_sets(arp_ctx* c) {
    if (!c) return ARP_E_ARG;
    if (!c->comm) FAIL(c, ARP_E_ARG, "arp_shard_reduce_residue_sets: no communicator (arp_comm_init)");
    if (!c->res_sel.p) FAIL(c, ARP_E_ARG, "arp_shard_reduce_residue_sets: residue sets do not exist yet (run stage 1 first)");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nb = 2 * (size_t)std::max<int64_t>(c->nres, 1);
    NCCLCHK(c, rccl().AllReduce(c->res_sel.p, c->res_sel.p, nb, ncclUint8, ncclMax, c->comm, c->stream));
    return ARP_OK;
}

#ifdef ARP_SEARCH_TRACE
// developer builds only: where k_search<MODE_CONTACTS> leaves its per-wave trace (device pointer to 4 u64 per wave, 0 = off)
int arp_debug_search_trace(arp_ctx* c, uint64_t device_ptr) {
    if (!c) return ARP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    unsigned long long* p = (unsigned long long*)(uintptr_t)device_ptr;
    HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(g_search_trace), &p, sizeof(p)));
    return ARP_OK;
}
int arp_debug_alloc(uint64_t bytes, uint64_t* device_ptr) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return ARP_E_HIP;
    (void)hipMemset(p, 0, bytes);
    *device_ptr = (uint64_t)(uintptr_t)p;
    return ARP_OK;
}
int arp_debug_read(uint64_t device_ptr, void* host, uint64_t bytes) {
    return hipMemcpy(host, (const void*)(uintptr_t)device_ptr, bytes, hipMemcpyDeviceToHost) == hipSuccess ? ARP_OK : ARP_E_HIP;
}
#endif

int arp_set_sort_after_pass(arp_ctx* c, int enabled) {
    if (!c) return ARP_E_ARG;
    c->sort_after_pass = enabled != 0;
    return ARP_OK;
}

int arp_set_packed_layout(arp_ctx* c, int layout) {
    if (!c || (layout != ARP_LAYOUT_RECORDS && layout != ARP_LAYOUT_ROWS)) return ARP_E_ARG;
    c->packed_csr = layout == ARP_LAYOUT_ROWS;
    return ARP_OK;
}

int arp_set_grid_reuse(arp_ctx* c, int enabled) {
    if (!c) return ARP_E_ARG;
    c->grid_reuse = enabled != 0;
    c->cg_valid = false;
    return ARP_OK;
}

int arp_set_whole_structure(arp_ctx* c, int enabled) {
    if (!c) return ARP_E_ARG;
    c->whole_structure = enabled != 0;
    c->contacts_valid = false;
    c->bag_ap.valid = c->bag_pp.valid = c->bag_gg.valid = c->bag_gp.valid = false;
    return ARP_OK;
}

int arp_get_host_times(arp_ctx* c, double us[2], int64_t* passes, int reset) {
    if (!c || !us || !passes) return ARP_E_ARG;
    us[0] = c->host_enqueue_us; us[1] = c->host_wait_us; *passes = c->host_passes;
    if (reset) { c->host_enqueue_us = c->host_wait_us = 0; c->host_passes = 0; }
    return ARP_OK;
}

uint64_t arp_stream_handle(arp_ctx* c) { return c ? (uint64_t)(uintptr_t)c->stream : 0; }

int arp_use_stream(arp_ctx* c, uint64_t stream) {
    if (!c) return ARP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (stream == 0) {
        c->stream = c->own_stream;
        c->external_stream = false;
    } else {
        c->stream = (hipStream_t)(uintptr_t)stream;
        c->external_stream = true;
    }
    return ARP_OK;
}

}  // extern "C"


// ---- mmCIF category reader (host only): arp_cif_api.h, shared with the g++-built libarpeggio_host.so ----------------------------
#include "arp_cif_api.h"
