// arp_sort.h — canonical order of the atom-atom bag on the device.
//
// The reference emits its atom-atom records in the order NeighborSearch.search_all delivers the pairs (interactions.py:707)
// and exports them in that order (interactions.py:183-190); that order is a property of a KD-tree over a hashed `set`, so the
// boundary defines a canonical one instead: ascending (bgn, end) = (i, j) packed atom index.  k_sift_planes
// leaves the records in the order of the pair list (eight per-XCD segments, cell by cell); this header puts them into the
// canonical order in HBM: a least-significant-digit radix sort on the key i << jbits | j with a 32-bit record index as
// payload, whose LAST pass writes the five result columns themselves — i and j from the key, distance / SIFt / contact type
// gathered through the index — into one slab, so that the host gets all five columns with ONE copy.
//
// Two launches per digit pass, no atomics on global memory, no data-dependent launch sizes:
//   k_sort_hist     block t counts the digits of its contiguous range of the input   -> table[digit][t]
//   k_sort_scatter  block t sums row `digit` of the table itself (T <= 256 entries per row, all rows in flight together: the
//                   "every block scans the small histogram out of L2" trick of k_scan_scatter_atoms), ranks its items
//                   stably — wave-level match by ballots, one LDS counter per (wave, digit) — and writes them
// A block's range is a whole number of tiles of 16 384 items which it walks in order, so T stays <= 256 whatever the
// count (25.6 M records of the 2 M-atom config: 7 tiles per block).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SORT_THREADS 1024
#define SORT_ITEMS 16
#define SORT_TILE (SORT_THREADS * SORT_ITEMS)
#define SORT_MAX_BITS 9
#define SORT_BINS (1 << SORT_MAX_BITS)
#define SORT_MAXT 256
#define SORT_WAVES (SORT_THREADS / 64)

struct SortArgs {
    // input of this pass: keys + record indices of the pass before, or (first pass) the bag's own i / j columns
    const unsigned long long* key_in;
    const uint32_t* idx_in;
    const int* ci;
    const int* cj;
    // output of this pass: keys + indices, or (last pass) the five columns of the sorted bag
    unsigned long long* key_out;
    uint32_t* idx_out;
    const float* d_in;
    const uint16_t* s_in;
    const uint8_t* ct_in;
    int* i_out;
    int* j_out;
    float* d_out;
    uint16_t* s_out;
    uint8_t* ct_out;
    long long n;       // records
    long long range;   // records per block (a multiple of SORT_TILE)
    int T;             // blocks
    int first, last;   // first / last pass
    int shift, bits;   // the digit of this pass: (key >> shift) & ((1 << bits) - 1)
    int jbits;         // key = i << jbits | j
    int* table;        // [SORT_BINS][SORT_MAXT]: items of block t with digit d
};

__device__ __forceinline__ unsigned long long sort_key_at(const SortArgs& A, long long p) {
    if (A.first) return ((unsigned long long)(uint32_t)A.ci[p] << A.jbits) | (unsigned long long)(uint32_t)A.cj[p];
    return A.key_in[p];
}

__global__ __launch_bounds__(SORT_THREADS) void k_sort_hist(SortArgs A) {
    __shared__ int s_hist[SORT_BINS];
    for (int d = threadIdx.x; d < SORT_BINS; d += SORT_THREADS) s_hist[d] = 0;
    __syncthreads();
    const long long lo = (long long)blockIdx.x * A.range, hi = min(lo + A.range, A.n);
    const uint32_t mask = (1u << A.bits) - 1u;
    for (long long p = lo + threadIdx.x; p < hi; p += SORT_THREADS)
        atomicAdd(&s_hist[(uint32_t)(sort_key_at(A, p) >> A.shift) & mask], 1);
    __syncthreads();
    for (int d = threadIdx.x; d < SORT_BINS; d += SORT_THREADS) A.table[d * SORT_MAXT + blockIdx.x] = s_hist[d];
}

__global__ __launch_bounds__(SORT_THREADS) void k_sort_scatter(SortArgs A) {
    __shared__ int s_whist[SORT_WAVES][SORT_BINS];   // per wave and digit: items seen so far in this tile, then the wave's offset
    __shared__ long long s_base[SORT_BINS];          // where the block's next item of a digit goes
    __shared__ int s_tot[SORT_BINS];
    __shared__ long long s_wsum[SORT_WAVES];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int t = blockIdx.x;
    const uint32_t mask = (1u << A.bits) - 1u;
    // ---- global base of every digit for this block: items of smaller digits in ALL blocks + items of this digit in the blocks before
    {
        long long before = 0, total = 0;
        if (threadIdx.x < SORT_BINS) {
            const int4* row = reinterpret_cast<const int4*>(A.table + threadIdx.x * SORT_MAXT);
            for (int q = 0; q * 4 < A.T; ++q) {
                const int4 v = row[q];
                const int vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int tt = q * 4 + k;
                    if (tt < A.T) { total += vv[k]; if (tt < t) before += vv[k]; }
                }
            }
        }
        // exclusive scan of `total` over the SORT_BINS digits (threads 0 .. SORT_BINS - 1 = waves 0 .. 7)
        long long incl = total;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const long long u = __shfl_up(incl, off);
            if (lane >= off) incl += u;
        }
        if (lane == 63) s_wsum[w] = incl;
        __syncthreads();
        long long woff = 0;
        for (int k = 0; k < w; ++k) woff += s_wsum[k];
        if (threadIdx.x < SORT_BINS) s_base[threadIdx.x] = woff + incl - total + before;
    }
    const long long lo = (long long)t * A.range, hi = min(lo + A.range, A.n);
    const unsigned long long below = (1ull << lane) - 1ull;
    for (long long tile = lo; tile < hi; tile += SORT_TILE) {
        for (int d = threadIdx.x; d < SORT_WAVES * SORT_BINS; d += SORT_THREADS) (&s_whist[0][0])[d] = 0;
        __syncthreads();     // (also orders s_base of the tile before / of the prologue)
        unsigned long long key[SORT_ITEMS];
        uint32_t idx[SORT_ITEMS];
        int rank[SORT_ITEMS];
        const long long wbase = tile + (long long)w * (64 * SORT_ITEMS);
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r) {
            const long long p = wbase + r * 64 + lane;
            const bool valid = p < hi;
            key[r] = valid ? sort_key_at(A, p) : 0ull;
            idx[r] = valid ? (A.first ? (uint32_t)p : A.idx_in[p]) : 0u;
        }
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r) {
            const long long p = wbase + r * 64 + lane;
            const bool valid = p < hi;
            const uint32_t d = (uint32_t)(key[r] >> A.shift) & mask;
            // lanes of this round with my digit (stable: lower lanes = earlier items)
            unsigned long long peers = __ballot(valid);
            for (int k = 0; k < A.bits; ++k) {
                const bool bit = (d >> k) & 1u;
                const unsigned long long b = __ballot(bit);
                peers &= bit ? b : ~b;
            }
            int old = 0;
            const int leader = valid ? (__ffsll((long long)peers) - 1) : lane;
            if (valid && lane == leader) {
                old = s_whist[w][d];
                s_whist[w][d] = old + __popcll(peers);
            }
            old = __shfl(old, leader);
            rank[r] = old + __popcll(peers & below);
        }
        __syncthreads();
        // per digit: counts of the waves -> exclusive offsets of the waves, total of the tile
        if (threadIdx.x < SORT_BINS) {
            int acc = 0;
#pragma unroll
            for (int k = 0; k < SORT_WAVES; ++k) {
                const int c = s_whist[k][threadIdx.x];
                s_whist[k][threadIdx.x] = acc;
                acc += c;
            }
            s_tot[threadIdx.x] = acc;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r) {
            const long long p = wbase + r * 64 + lane;
            if (p >= hi) continue;
            const uint32_t d = (uint32_t)(key[r] >> A.shift) & mask;
            const long long pos = s_base[d] + s_whist[w][d] + rank[r];
            if (!A.last) {
                A.key_out[pos] = key[r];
                A.idx_out[pos] = idx[r];
            } else {
                const uint32_t q = idx[r];
                A.i_out[pos] = (int)(key[r] >> A.jbits);
                A.j_out[pos] = (int)(key[r] & ((1ull << A.jbits) - 1ull));
                A.d_out[pos] = A.d_in[q];
                A.s_out[pos] = A.s_in[q];
                A.ct_out[pos] = A.ct_in[q];
            }
        }
        __syncthreads();
        if (threadIdx.x < SORT_BINS) s_base[threadIdx.x] += s_tot[threadIdx.x];
    }
}
