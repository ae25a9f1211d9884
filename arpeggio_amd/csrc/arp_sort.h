// arp_sort.h — canonical order of the atom-atom bag on the device.
//
// The reference emits its atom-atom records in the order NeighborSearch.search_all delivers the pairs (interactions.py:707)
// and exports them in that order (interactions.py:183-190); that order is a property of a KD-tree over a hashed `set`, so the
// boundary defines a canonical one instead: ascending (bgn, end) = (i, j) packed atom index.  k_sift_planes
// leaves the records in the order of the pair list (eight per-XCD segments, cell by cell); this header puts them into the
// canonical order in HBM: least-significant-digit radix passes on the key i << jbits | j with the rest of the record
// (distance, SIFt, contact type: 56 bits) as payload, and a last launch that writes the five result columns themselves into
// one slab, so that the host gets all five columns with ONE copy.
//
// The radix passes sort by i ONLY (its significant bits, up to 9 per pass: two passes below 262 144 atoms); three launches
// per pass, no atomics on global memory, every launch fills the chip:
//   k_sort_hist     block t counts the digits of tile t (4096 records)                              -> table[digit][t]
//   k_sort_scan     block d turns row d of the table into exclusive prefixes over the tiles, total[d] = its sum
//   k_sort_scatter  block t: base of digit d = (exclusive scan of total[])[d] + table[d][t]; ranks its records stably —
//                   wave-level match by ballots, one LDS counter per (wave, digit), the sixteen rounds' counter updates
//                   issued back to back as returning LDS adds —, puts them into tile order in LDS and writes them out
//                   with consecutive lanes on consecutive addresses of a digit's run
// The records of one bgn atom are then one short run (a dozen records, a few hundred in a clump), in any order:
//   k_sort_runs     one thread per record: the run it is in (neighbours with the same i, read through L1), its rank by j among
//                   them — the pairs of a bag are distinct, so there are no ties — and the five columns written at
//                   run start + rank.  Two radix passes on j (two thirds of the launches of a four-pass sort) become one
//                   launch that moves no keys.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SORT_THREADS 256
#define SORT_ITEMS 16
#define SORT_TILE (SORT_THREADS * SORT_ITEMS)
#define SORT_MAX_BITS 9
#define SORT_BINS (1 << SORT_MAX_BITS)
#define SORT_WAVES (SORT_THREADS / 64)

struct SortArgs {
    // input of this pass: keys + record indices of the pass before, or (first pass) the bag's own i / j columns
    const unsigned long long* key_in;
    const unsigned long long* val_in;      // the record's payload: distance bits | SIFt << 32 | contact type << 48
    const int* ci;
    const int* cj;
    // output of this pass: keys + indices, or (last pass) the five columns of the sorted bag
    unsigned long long* key_out;
    unsigned long long* val_out;
    const float* d_in;
    const uint16_t* s_in;
    const uint8_t* ct_in;
    int* i_out;
    int* j_out;
    float* d_out;
    uint16_t* s_out;
    uint8_t* ct_out;
    long long n;       // records
    int T;             // tiles = blocks of k_sort_hist / k_sort_scatter
    int tstride;       // entries per row of the table (>= T, a multiple of 4)
    int first, last;   // first / last pass
    int shift, bits;   // the digit of this pass: (key >> shift) & ((1 << bits) - 1)
    int jbits;         // key = i << jbits | j
    int* table;        // [SORT_BINS][tstride]: records of tile t with digit d, then their exclusive prefix over the tiles
    long long* total;  // [SORT_BINS]: records with digit d
    // CSR layout of the sorted bag (arp_set_packed_layout): not null -> k_sort_runs writes no i column but the row offsets —
    // row_out[a] = first record of bgn atom a, row_out[nrows] = n; the records of atom a are [row_out[a], row_out[a + 1])
    int* row_out;
    int nrows;
};

__device__ __forceinline__ unsigned long long sort_key_at(const SortArgs& A, long long p) {
    if (A.first) return ((unsigned long long)(uint32_t)A.ci[p] << A.jbits) | (unsigned long long)(uint32_t)A.cj[p];
    return A.key_in[p];
}

// the payload travels with the key (a gather through a record index at the end was three random sector reads per record)
__device__ __forceinline__ unsigned long long sort_val_at(const SortArgs& A, long long p) {
    if (A.first) return (unsigned long long)__float_as_uint(A.d_in[p]) | ((unsigned long long)A.s_in[p] << 32) | ((unsigned long long)A.ct_in[p] << 48);
    return A.val_in[p];
}

__global__ __launch_bounds__(SORT_THREADS) void k_sort_hist(SortArgs A) {
    __shared__ int s_hist[SORT_BINS];
    for (int d = threadIdx.x; d < SORT_BINS; d += SORT_THREADS) s_hist[d] = 0;
    __syncthreads();
    const long long lo = (long long)blockIdx.x * SORT_TILE, hi = min(lo + SORT_TILE, A.n);
    const uint32_t mask = (1u << A.bits) - 1u;
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const long long p = lo + r * SORT_THREADS + threadIdx.x;
        // (the digits are bits of i: the first pass reads the i column alone, not the j column it would only shift away)
        if (p < hi) atomicAdd(&s_hist[A.first ? (((uint32_t)A.ci[p] >> (A.shift - A.jbits)) & mask) : ((uint32_t)(A.key_in[p] >> A.shift) & mask)], 1);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < (1 << A.bits); d += SORT_THREADS) A.table[(size_t)d * A.tstride + blockIdx.x] = s_hist[d];
}

// exclusive scan of one value per thread over the block (SORT_THREADS threads); *sum = the block's total
__device__ __forceinline__ long long sort_block_scan(long long v, long long* s_w, long long* sum) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    long long incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const long long u = __shfl_up(incl, off);
        if (lane >= off) incl += u;
    }
    __syncthreads();                 // (s_w may still be read by the scan before)
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    long long woff = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < SORT_WAVES; ++k) { const long long x = s_w[k]; if (k < w) woff += x; tot += x; }
    if (sum) *sum = tot;
    return woff + incl - v;
}

__global__ __launch_bounds__(SORT_THREADS) void k_sort_scan(SortArgs A) {
    __shared__ long long s_w[SORT_WAVES];
    int* row = A.table + (size_t)blockIdx.x * A.tstride;
    long long run = 0;
    for (int t0 = 0; t0 < A.T; t0 += SORT_THREADS) {      // (block-uniform trip count)
        const int t = t0 + threadIdx.x;
        const int v = t < A.T ? row[t] : 0;
        long long sum;
        const long long e = sort_block_scan((long long)v, s_w, &sum);
        if (t < A.T) row[t] = (int)(run + e);              // (a tile's prefix inside one digit is below 2^31: the index payload is 32 bits)
        run += sum;
    }
    if (threadIdx.x == 0) A.total[blockIdx.x] = run;
}

__global__ __launch_bounds__(SORT_THREADS) void k_sort_scatter(SortArgs A) {
    __shared__ unsigned long long s_key[SORT_TILE];
    __shared__ unsigned long long s_val[SORT_TILE];
    __shared__ int s_whist[SORT_WAVES][SORT_BINS];   // per wave and digit: records seen so far, then the wave's offset in the digit's run
    __shared__ int s_texcl[SORT_BINS];               // first slot of digit d in the tile's sorted order
    __shared__ long long s_gbase[SORT_BINS];         // global position of that slot
    __shared__ long long s_w[SORT_WAVES];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int t = blockIdx.x;
    const int nb = 1 << A.bits;
    const uint32_t mask = (uint32_t)nb - 1u;
    // ---- global base of my two digits: records of smaller digits in all tiles + records of the digit in the tiles before
    const int d0 = 2 * threadIdx.x, d1 = d0 + 1;
    const long long tot0 = d0 < nb ? A.total[d0] : 0, tot1 = d1 < nb ? A.total[d1] : 0;
    const int pre0 = d0 < nb ? A.table[(size_t)d0 * A.tstride + t] : 0, pre1 = d1 < nb ? A.table[(size_t)d1 * A.tstride + t] : 0;
    // ---- the tile's records
    const long long lo = (long long)t * SORT_TILE, hi = min(lo + SORT_TILE, A.n);
    const long long wbase = lo + (long long)w * (64 * SORT_ITEMS);
    unsigned long long key[SORT_ITEMS];
    unsigned long long val[SORT_ITEMS];
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const long long p = wbase + r * 64 + lane;
        const bool valid = p < hi;
        key[r] = valid ? sort_key_at(A, p) : ~0ull;
        val[r] = valid ? sort_val_at(A, p) : 0ull;
    }
    for (int d = threadIdx.x; d < SORT_WAVES * SORT_BINS; d += SORT_THREADS) (&s_whist[0][0])[d] = 0;
    {
        const long long e = sort_block_scan(tot0 + tot1, s_w, nullptr);      // (its barriers also publish the zeroed counters)
        if (d0 < SORT_BINS) { s_gbase[d0] = e + pre0; s_gbase[d1] = e + tot0 + pre1; }
    }
    // ---- stable rank of every record among the records of its digit in this wave's part of the tile
    const unsigned long long below = (1ull << lane) - 1ull;
    int rk[SORT_ITEMS];      // peers below me | leader << 8
    int old[SORT_ITEMS];
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const bool valid = wbase + r * 64 + lane < hi;
        const uint32_t d = (uint32_t)(key[r] >> A.shift) & mask;
        unsigned long long peers = __ballot(valid);
        for (int k = 0; k < A.bits; ++k) {
            const bool bit = (d >> k) & 1u;
            const unsigned long long b = __ballot(bit);
            peers &= bit ? b : ~b;
        }
        const int leader = valid ? (__ffsll((long long)peers) - 1) : lane;
        rk[r] = __popcll(peers & below) | (leader << 8);
        old[r] = 0;
        // one returning LDS add per digit and round, by its first lane; the sixteen rounds' adds leave back to back (LDS
        // operations of a wave execute in order, so a digit's counter sees the rounds in order) and are waited for once
        if (valid && lane == leader) old[r] = atomicAdd(&s_whist[w][d], __popcll(peers));
    }
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) rk[r] = __shfl(old[r], rk[r] >> 8) + (rk[r] & 255);
    __syncthreads();
    // ---- per digit: exclusive offsets of the waves, size of the digit's run in the tile, first slot of the run
    {
        int c0 = 0, c1 = 0;
#pragma unroll
        for (int k = 0; k < SORT_WAVES; ++k) {
            const int a = s_whist[k][d0], b = s_whist[k][d1];
            s_whist[k][d0] = c0; s_whist[k][d1] = c1;
            c0 += a; c1 += b;
        }
        const int e = (int)sort_block_scan((long long)(c0 + c1), s_w, nullptr);
        s_texcl[d0] = e; s_texcl[d1] = e + c0;
        s_gbase[d0] -= e; s_gbase[d1] -= e + c0;       // position of slot s of digit d = s_gbase[d] + s
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        if (wbase + r * 64 + lane >= hi) continue;
        const uint32_t d = (uint32_t)(key[r] >> A.shift) & mask;
        const int slot = s_texcl[d] + s_whist[w][d] + rk[r];
        s_key[slot] = key[r];
        s_val[slot] = val[r];
    }
    __syncthreads();
    const int cnt = (int)(hi - lo);
#pragma unroll 4
    for (int s = threadIdx.x; s < cnt; s += SORT_THREADS) {
        const unsigned long long k = s_key[s];
        const unsigned long long q = s_val[s];
        const long long pos = s_gbase[(uint32_t)(k >> A.shift) & mask] + s;
        A.key_out[pos] = k;
        A.val_out[pos] = q;
    }
}

// A small bag (a protein has ~2 x 10^4 contacts) in ONE launch instead of the six of two radix passes, which are all launch
// latency at that size: one block counts the records of every bgn atom in LDS (a counter per atom id), scans the counters, and
// moves key + payload to the start of their atom's run plus a ticket — in ANY order inside the run: k_sort_runs ranks a record
// by j among the run whatever order the run arrives in.  (The first version kept the counters in global memory: a thread's
// twenty returning atomics in a row, 2 - 4 us each, made it slower than the radix passes.)
// One CU's scattered 8-byte stores bound it (1.4 us per 1000 records): 21 k records 46 against 65 us, even at 38 k.
#define SORT_SMALL_THREADS 1024
#define SORT_SMALL_MAX_RECORDS 32768
#define SORT_SMALL_MAX_BINS 12288
#define SORT_SMALL_BLOCKS 32
// Since round 5 the one block is SORT_SMALL_BLOCKS blocks that do not talk to each other: block b owns the atom ids
// [b, b + 1) * nbin / blocks; every block reads ALL the i column (coalesced, a few trips), counts the records of its own ids in LDS
// and the records of smaller ids in registers (= where its part of the output begins), and places its own records.  The scattered
// stores, which bound the kernel, are shared by as many CUs: 16.5 k records 27.1 us with one block, 15.6 with 8, 14.5 with 16, 11.8 with 32.
__global__ __launch_bounds__(SORT_SMALL_THREADS) void k_sort_small(SortArgs A, int nbin) {
    __shared__ int s_cnt[SORT_SMALL_MAX_BINS];
    __shared__ int s_wsum[SORT_SMALL_THREADS / 64];
    __shared__ int s_below[SORT_SMALL_THREADS / 64];
    const int tid = (int)threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n = (int)A.n;
    const int lo = (int)((long long)blockIdx.x * nbin / gridDim.x), hi = (int)((long long)(blockIdx.x + 1) * nbin / gridDim.x);
    const int mine_bins = hi - lo;
    for (int b = tid; b < mine_bins; b += SORT_SMALL_THREADS) s_cnt[b] = 0;
    __syncthreads();
    // (four records per thread and step: 16-byte loads, a quarter of the trips — a trip is a memory latency)
    const int n4 = n >> 2;
    int below = 0;
    auto count = [&](int i) {
        if (i < lo) ++below;
        else if (i < hi) atomicAdd(&s_cnt[i - lo], 1);
    };
#pragma unroll 2
    for (int q = tid; q < n4; q += SORT_SMALL_THREADS) {
        const int4 i4 = reinterpret_cast<const int4*>(A.ci)[q];
        count(i4.x); count(i4.y); count(i4.z); count(i4.w);
    }
    for (int p = (n4 << 2) + tid; p < n; p += SORT_SMALL_THREADS) count(A.ci[p]);
    for (int o = 32; o > 0; o >>= 1) below += __shfl_xor(below, o);
    if (lane == 0) s_below[w] = below;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < SORT_SMALL_THREADS / 64; ++k) base += s_below[k];
    // exclusive scan in place: a contiguous chunk of bins per thread, the chunks' sums scanned over the block
    const int per = (mine_bins + SORT_SMALL_THREADS - 1) / SORT_SMALL_THREADS;
    const int b_lo = min(tid * per, mine_bins), b_hi = min(b_lo + per, mine_bins);
    int mine = 0;
    for (int b = b_lo; b < b_hi; ++b) mine += s_cnt[b];
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int u = __shfl_up(incl, off);
        if (lane >= off) incl += u;
    }
    if (lane == 63) s_wsum[w] = incl;
    __syncthreads();
    int run = base + incl - mine;
    for (int k = 0; k < w; ++k) run += s_wsum[k];
    for (int b = b_lo; b < b_hi; ++b) {
        const int c = s_cnt[b];
        s_cnt[b] = run;
        run += c;
    }
    __syncthreads();
    auto place = [&](uint32_t i, uint32_t j, float d, uint32_t sf, uint32_t ct) {
        if ((int)i < lo || (int)i >= hi) return;
        const int pos = atomicAdd(&s_cnt[i - lo], 1);
        A.key_out[pos] = ((unsigned long long)i << A.jbits) | (unsigned long long)j;
        A.val_out[pos] = (unsigned long long)__float_as_uint(d) | ((unsigned long long)sf << 32) | ((unsigned long long)ct << 48);
    };
#pragma unroll 2
    for (int q = tid; q < n4; q += SORT_SMALL_THREADS) {
        const int4 i4 = reinterpret_cast<const int4*>(A.ci)[q];
        const bool any = (i4.x >= lo && i4.x < hi) || (i4.y >= lo && i4.y < hi) || (i4.z >= lo && i4.z < hi) || (i4.w >= lo && i4.w < hi);
        if (!any) continue;
        const int4 j4 = reinterpret_cast<const int4*>(A.cj)[q];
        const float4 d4 = reinterpret_cast<const float4*>(A.d_in)[q];
        const uint2 s4 = reinterpret_cast<const uint2*>(A.s_in)[q];
        const uint32_t c4 = reinterpret_cast<const uint32_t*>(A.ct_in)[q];
        place((uint32_t)i4.x, (uint32_t)j4.x, d4.x, s4.x & 0xFFFFu, c4 & 0xFFu);
        place((uint32_t)i4.y, (uint32_t)j4.y, d4.y, s4.x >> 16, (c4 >> 8) & 0xFFu);
        place((uint32_t)i4.z, (uint32_t)j4.z, d4.z, s4.y & 0xFFFFu, (c4 >> 16) & 0xFFu);
        place((uint32_t)i4.w, (uint32_t)j4.w, d4.w, s4.y >> 16, c4 >> 24);
    }
    for (int p = (n4 << 2) + tid; p < n; p += SORT_SMALL_THREADS)
        place((uint32_t)A.ci[p], (uint32_t)A.cj[p], A.d_in[p], A.s_in[p], A.ct_in[p]);
}

// After the radix passes: key_in / idx_in sorted by i.  Thread p: the run of records with its i, its rank by j inside the
// run, the five columns of the sorted bag at run start + rank.
#define RUN_HALO 64
__global__ __launch_bounds__(256) void k_sort_runs(SortArgs A) {
    // i and j of the block's 256 positions and of RUN_HALO on either side, in LDS: a record looks at its neighbours without a
    // chain of dependent loads (the run of a bgn atom is a dozen records; what is longer than the halo goes on through global
    // memory).  Atom indices are below 2^31: both halves of a key are 32-bit numbers, 0xFFFFFFFF marks "no record".
    __shared__ uint32_t s_i[256 + 2 * RUN_HALO], s_j[256 + 2 * RUN_HALO];
    const long long b0 = (long long)blockIdx.x * 256;
    const unsigned long long jmask = (1ull << A.jbits) - 1ull;
    for (int t = threadIdx.x; t < 256 + 2 * RUN_HALO; t += 256) {
        const long long q = b0 - RUN_HALO + t;
        const bool in = q >= 0 && q < A.n;
        const unsigned long long k = in ? A.key_in[q] : 0ull;
        s_i[t] = in ? (uint32_t)(k >> A.jbits) : 0xFFFFFFFFu;
        s_j[t] = (uint32_t)(k & jmask);
    }
    __syncthreads();
    const long long p = b0 + threadIdx.x;
    if (p >= A.n) return;
    const int s = (int)threadIdx.x + RUN_HALO;
    const uint32_t i = s_i[s], j = s_j[s];
    int rank = 0, left = 0, right = 0;
    for (int k0 = 1; k0 <= RUN_HALO; k0 += 8) {      // (a run is contiguous: `same` is 1, 1, ..., 1, 0, 0, ...; the wave stops when all its runs have ended)
        int any = 0;
#pragma unroll
        for (int k = k0; k < k0 + 8; ++k) {
            const int sl = (s_i[s - k] == i) ? 1 : 0, sr = (s_i[s + k] == i) ? 1 : 0;
            left += sl; right += sr;
            rank += (sl & ((s_j[s - k] < j) ? 1 : 0)) + (sr & ((s_j[s + k] < j) ? 1 : 0));
            any = sl | sr;
        }
        if (!__any(any)) break;
    }
    long long a = p - left;
    if (left == RUN_HALO)
        for (long long q = p - RUN_HALO - 1; q >= 0; --q) {
            const unsigned long long kq = A.key_in[q];
            if ((uint32_t)(kq >> A.jbits) != i) break;
            rank += ((uint32_t)(kq & jmask) < j) ? 1 : 0;
            a = q;
        }
    if (right == RUN_HALO)
        for (long long q = p + RUN_HALO + 1; q < A.n; ++q) {
            const unsigned long long kq = A.key_in[q];
            if ((uint32_t)(kq >> A.jbits) != i) break;
            rank += ((uint32_t)(kq & jmask) < j) ? 1 : 0;
        }
    const long long pos = a + rank;
    const unsigned long long v = A.val_in[p];
    if (A.row_out) {
        // the first record of a run knows where the rows of its atom — and of the record-less atoms since the run before it — begin
        // (hydrogens present as atoms have no records: gaps of a few ids); the last record of the bag closes the table
        if (p == a) {
            const uint32_t ip = s_i[s - 1];                       // (0xFFFFFFFF in front of the first record)
            for (long long id = (ip == 0xFFFFFFFFu) ? 0 : (long long)ip + 1; id <= (long long)i && id <= (long long)A.nrows; ++id) A.row_out[id] = (int)a;
        }
        if (p == A.n - 1)
            for (long long id = (long long)i + 1; id <= (long long)A.nrows; ++id) A.row_out[id] = (int)A.n;
    } else
    A.i_out[pos] = (int)i;
    A.j_out[pos] = (int)j;
    A.d_out[pos] = __uint_as_float((uint32_t)v);
    A.s_out[pos] = (uint16_t)(v >> 32);
    A.ct_out[pos] = (uint8_t)(v >> 48);
}
