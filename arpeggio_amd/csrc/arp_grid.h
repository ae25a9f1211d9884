// arp_grid.h — uniform spatial-hash grid: binning, counting sort, cell ordering.
//
// Stands in for Bio.PDB.NeighborSearch's KD-tree build (interactions.py:1394,1442).
// Points are binned in float64 with a cell edge >= the search radius, so every pair
// within the radius lies in a 27-cell stencil.  The sort is a counting sort:
//   k_bin      cell id per point + histogram (atomicAdd on the cell counter)   [rings / amides;
//              atoms use k_bin_atoms / k_scatter_atoms of arp_pairs.h, which also build the records]
//   k_scan_*   exclusive prefix sum of the histogram (one block up to 65 536 cells, 3-phase above)
//   k_scatter  point -> slot inside its cell (atomicSub on the same counter, which ends at zero)
//   k_cellsort ascending point id inside each cell (optional: deterministic device-side order)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "arp_numerics.h"

// Several structures in one grid (arp_set_batch): every structure keeps its own coordinates and its own box; its cells sit
// at an integer offset inside a common grid, with at least one empty cell between two structures, so that no stencil ever
// reaches from one structure into another.  place[s] = origin of structure s's box, its cell offset and its cell counts;
// sid_* = structure of every atom (by local id), ring and amide.
struct BatchPlace {
    double ox, oy, oz;
    int cx, cy, cz, nx, ny, nz;
};
struct GridDesc {
    double ox, oy, oz, inv;
    int nx, ny, nz, ncell;
    const BatchPlace* place;      // null: one structure, the box of the grid is its box
    const int* sid_atom;
    const int* sid_ring;
    const int* sid_amide;
};

struct PtsF3 {  // packed float32[3] (amide centres)
    const float* p;
    __device__ __forceinline__ num::d3 get(int i) const {
        return {(double)p[3 * (size_t)i], (double)p[3 * (size_t)i + 1], (double)p[3 * (size_t)i + 2]};
    }
};
struct PtsD3 {  // packed float64[3] (ring centres)
    const double* p;
    __device__ __forceinline__ num::d3 get(int i) const {
        return {p[3 * (size_t)i], p[3 * (size_t)i + 1], p[3 * (size_t)i + 2]};
    }
};

// Unclamped integer cell coordinate (can be -1 or n for points outside the box).
__device__ __forceinline__ int cell_coord_raw(double v, double o, double inv, int n) {
    double t = floor((v - o) * inv);
    t = fmin(fmax(t, -2.0), (double)n + 1.0);
    return (int)t;
}
__device__ __forceinline__ int cell_coord(double v, double o, double inv, int n) {
    int c = cell_coord_raw(v, o, inv, n);
    return min(max(c, 0), n - 1);
}
__device__ __forceinline__ int cell_index(const GridDesc& g, num::d3 p) {
    int cx = cell_coord(p.x, g.ox, g.inv, g.nx);
    int cy = cell_coord(p.y, g.oy, g.inv, g.ny);
    int cz = cell_coord(p.z, g.oz, g.inv, g.nz);
    return (cz * g.ny + cy) * g.nx + cx;
}

// ... of a point of structure sid when the grid holds several structures (points outside their structure's box count
// as its border cells: clamping is monotone, so two points within the radius still land at most one cell apart)
__device__ __forceinline__ int cell_index(const GridDesc& g, num::d3 p, int sid) {
    if (!g.place) return cell_index(g, p);
    const BatchPlace b = g.place[sid];
    const int cx = b.cx + cell_coord(p.x, b.ox, g.inv, b.nx);
    const int cy = b.cy + cell_coord(p.y, b.oy, g.inv, b.ny);
    const int cz = b.cz + cell_coord(p.z, b.oz, g.inv, b.nz);
    return (cz * g.ny + cy) * g.nx + cx;
}

// cell id + histogram for every point (ring and amide centres); sid: structure of every point (batched grids)
template <class P>
__global__ __launch_bounds__(256) void k_bin(P pts, int n, GridDesc g, const int* __restrict__ sid, int* __restrict__ cell_of,
                                             int* __restrict__ cell_cnt) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int c = cell_index(g, pts.get(i), g.place ? sid[i] : 0);
        atomicAdd(&cell_cnt[c], 1);
        cell_of[i] = c;
    }
}


// Small grids (<= 65536 cells): the whole exclusive scan in ONE 1024-thread block (one launch
// instead of three).  Thread t owns ITEMS consecutive counters (int4 loads, kept in registers);
// out[n] = grand total.  Both arrays must be readable/writable up to ITEMS * 1024 elements.
// (in == out is allowed: a thread reads all of its own counters before anything is written)
template <int ITEMS>
__device__ __forceinline__ void scan_small_body(const int* in, int n, int* out, int* total_slot, unsigned long long* total_out) {
    __shared__ int sh[32];
    const int base = threadIdx.x * ITEMS;
    int v[ITEMS];
    int sum = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; k += 4) {
        const int4 q = *reinterpret_cast<const int4*>(in + base + k);
        v[k] = (base + k < n) ? q.x : 0;
        v[k + 1] = (base + k + 1 < n) ? q.y : 0;
        v[k + 2] = (base + k + 2 < n) ? q.z : 0;
        v[k + 3] = (base + k + 3 < n) ? q.w : 0;
        sum += v[k] + v[k + 1] + v[k + 2] + v[k + 3];
    }
    // inclusive scan of the 1024 thread sums: shuffles inside each wavefront (no barrier), then the 16
    // wave totals through LDS (two barriers in all)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) sh[wv] = incl;
    __syncthreads();
    if (wv == 0) {
        int w_incl = (lane < 16) ? sh[lane] : 0;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            const int t = __shfl_up(w_incl, off);
            if (lane >= off) w_incl += t;
        }
        if (lane < 16) sh[16 + lane] = w_incl;   // inclusive totals of waves 0..lane
    }
    __syncthreads();
    const int wave_base = (wv == 0) ? 0 : sh[16 + wv - 1];
    int run = wave_base + incl - sum;
#pragma unroll
    for (int k = 0; k < ITEMS; k += 4) {
        int4 q;
        q.x = run; run += v[k];
        q.y = run; run += v[k + 1];
        q.z = run; run += v[k + 2];
        q.w = run; run += v[k + 3];
        if (base + k < n) *reinterpret_cast<int4*>(out + base + k) = q;   // may spill past n inside the padded buffer
    }
    if (threadIdx.x == 1023) {
        *total_slot = sh[16 + 15];
        if (total_out) *total_out = (unsigned long long)sh[16 + 15];   // number of binned points, for the host statistics
    }
}
template <int ITEMS>
__global__ __launch_bounds__(1024) void k_scan_small(const int* __restrict__ in, int n, int* __restrict__ out,
                                                     unsigned long long* __restrict__ total_out) {
    scan_small_body<ITEMS>(in, n, out, out + n, total_out);
}
template <int ITEMS>
__global__ __launch_bounds__(1024) void k_scan_inplace(int* a, int n) {    // a[0..n) -> exclusive prefix, a[n] = total
    scan_small_body<ITEMS>(a, n, a, a + n, nullptr);
}

// Ring AND amide centre grids of a structure in ONE launch (block 0: rings, block 1: amides): bin, scan and scatter by a
// single 1024-thread block each — a few thousand points in a few thousand cells, where the six launches of the general path
// (k_bin / k_scan_small / k_scatter, twice) were 28 us of launch latency in every structure's first pass.  The histogram is
// zero on entry and on exit (the scatter counts it down again), as in the general path.
template <class P>
struct PointGridJob {
    P pts;
    int n;
    GridDesc g;
    const int* sid;
    int* cell_of;
    int* cnt;
    int* start;
    int* perm;
};
#define POINT_GRID_ITEMS 32          // cells per thread of the one-block scan: grids up to 32768 cells
#define POINT_GRID_LDS_CELLS 12288   // up to this many cells the histogram lives in LDS (48 KB)
template <class P>
__device__ __forceinline__ void point_grid_body(const PointGridJob<P>& J, int* s_cnt) {
    if (J.n <= 0) return;
    const int ncell = J.g.ncell;
    const bool in_lds = ncell <= POINT_GRID_LDS_CELLS;     // (block-uniform)
    int* const cnt = in_lds ? s_cnt : J.cnt;
    if (in_lds) {
        for (int k = threadIdx.x; k < POINT_GRID_LDS_CELLS; k += blockDim.x) s_cnt[k] = 0;
        __syncthreads();
    }
    for (int i = threadIdx.x; i < J.n; i += blockDim.x) {
        const int c = cell_index(J.g, J.pts.get(i), J.g.place ? J.sid[i] : 0);
        atomicAdd(&cnt[c], 1);
        J.cell_of[i] = c;
    }
    if (!in_lds) __threadfence();      // (counters updated on the memory side: the scan must not read them from this CU's L1)
    __syncthreads();
    if (in_lds) scan_small_body<POINT_GRID_LDS_CELLS / 1024>(cnt, ncell, J.start, J.start + ncell, nullptr);
    else scan_small_body<POINT_GRID_ITEMS>(cnt, ncell, J.start, J.start + ncell, nullptr);
    __threadfence_block();
    __syncthreads();
    for (int i = threadIdx.x; i < J.n; i += blockDim.x) {
        const int c = J.cell_of[i];
        const int slot = atomicSub(&cnt[c], 1) - 1;      // (the global histogram ends at zero again, as the general path leaves it)
        J.perm[J.start[c] + slot] = i;
    }
}
// zero4: four 64-bit words this launch clears on the way (the entry counts of the candidate lists that are built next)
__global__ __launch_bounds__(1024) void k_point_grids(PointGridJob<PtsD3> rings, PointGridJob<PtsF3> amides, unsigned long long* zero4) {
    __shared__ int s_cnt[POINT_GRID_LDS_CELLS];
    if (zero4 && blockIdx.x == 0 && threadIdx.x < 4) zero4[threadIdx.x] = 0ull;
    if (blockIdx.x == 0) point_grid_body(rings, s_cnt);
    else point_grid_body(amides, s_cnt);
}

// Larger histograms, two launches: every 1024-thread block scans one tile of TILE_ITEMS * 1024 counters with the
// same register / shuffle scan (tile totals to tile_sums), then k_scan_fix adds to every tile the sum of the totals
// before it (each block sums those few numbers itself: no third "scan of the totals" launch) and writes out[n].
#define TILE_ITEMS 16
#define TILE_CELLS (TILE_ITEMS * 1024)
__global__ __launch_bounds__(1024) void k_scan_tiles(const int* __restrict__ in, int n, int* __restrict__ out,
                                                     int* __restrict__ tile_sums) {
    const int t0 = blockIdx.x * TILE_CELLS;
    scan_small_body<TILE_ITEMS>(in + t0, min(n - t0, TILE_CELLS), out + t0, tile_sums + blockIdx.x, nullptr);
}
// The same in ONE launch for up to CHAIN_TILES tiles (every block of the launch is resident at once: two 1024-thread blocks
// per CU): a tile publishes its total as soon as it has it — one 64-bit word {launch number, total}, so nothing needs
// clearing between launches — and waits for the totals of the tiles before it (they are a handful of words), adds their
// sum to its prefixes before it writes them.  The wait is bounded: a tile that is not served within ~2^20 polls raises the
// device error flag (*err = ARP_E_HIP: the host fails the pass with a message) and goes on with what it has, instead of
// hanging the queue or killing the context.  The host uses this path only while every tile is resident at once
// (ntiles <= 2 blocks per CU) and clears the chain when the 32-bit launch number wraps.
#define CHAIN_TILES 128
__global__ __launch_bounds__(1024) void k_scan_tiles_chained(const int* __restrict__ in, int n, int* __restrict__ out,
                                                             unsigned long long* __restrict__ chain, unsigned int epoch,
                                                             unsigned long long* __restrict__ total_out, int* __restrict__ err) {
    __shared__ int sh[32];
    __shared__ int s_off;
    constexpr int ITEMS = TILE_ITEMS;
    const int tile = blockIdx.x, t0 = tile * TILE_CELLS, nt = min(n - t0, TILE_CELLS);
    const int base = threadIdx.x * ITEMS;
    int v[ITEMS];
    int sum = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; k += 4) {
        const int4 q = *reinterpret_cast<const int4*>(in + t0 + base + k);
        v[k] = (base + k < nt) ? q.x : 0;
        v[k + 1] = (base + k + 1 < nt) ? q.y : 0;
        v[k + 2] = (base + k + 2 < nt) ? q.z : 0;
        v[k + 3] = (base + k + 3 < nt) ? q.w : 0;
        sum += v[k] + v[k + 1] + v[k + 2] + v[k + 3];
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) sh[wv] = incl;
    __syncthreads();
    if (wv == 0) {
        int w_incl = (lane < 16) ? sh[lane] : 0;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            const int t = __shfl_up(w_incl, off);
            if (lane >= off) w_incl += t;
        }
        if (lane < 16) sh[16 + lane] = w_incl;   // inclusive totals of waves 0..lane
        const int total = __shfl(w_incl, 15);
        if (lane == 0)
            __hip_atomic_store(chain + tile, ((unsigned long long)epoch << 32) | (unsigned int)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the word is all a reader needs)
        // the totals of the tiles before this one (lane l takes tiles l, l + 64)
        int before = 0;
        for (int k = lane; k < tile; k += 64) {
            unsigned long long w = 0;
            int spins = 0;
            for (;;) {
                w = __hip_atomic_load(chain + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned int)(w >> 32) == epoch) break;
                if (++spins > (1 << 20)) { atomicExch(err, -2 /* ARP_E_HIP */); w = 0; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            before += (int)(unsigned int)w;
        }
        for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
        if (lane == 0) {
            s_off = before;
            if (tile == (int)gridDim.x - 1) {
                out[n] = before + total;
                if (total_out) *total_out = (unsigned long long)(before + total);
            }
        }
    }
    __syncthreads();
    const int wave_base = (wv == 0) ? 0 : sh[16 + wv - 1];
    int run = s_off + wave_base + incl - sum;
#pragma unroll
    for (int k = 0; k < ITEMS; k += 4) {
        int4 q;
        q.x = run; run += v[k];
        q.y = run; run += v[k + 1];
        q.z = run; run += v[k + 2];
        q.w = run; run += v[k + 3];
        // (padded buffers: the int4 may run past the tile's last cell — into the next tile's first prefixes, which that tile
        // writes itself, or past n, where out[n] must stay: write only the cells that are this tile's)
        if (base + k + 3 < nt) *reinterpret_cast<int4*>(out + t0 + base + k) = q;
        else {
            if (base + k < nt) out[t0 + base + k] = q.x;
            if (base + k + 1 < nt) out[t0 + base + k + 1] = q.y;
            if (base + k + 2 < nt) out[t0 + base + k + 2] = q.z;
        }
    }
}
__global__ __launch_bounds__(1024) void k_scan_fix(int* __restrict__ out, int n, const int* __restrict__ tile_sums, int ntiles,
                                                   unsigned long long* __restrict__ total_out) {
    __shared__ int sh[16];
    __shared__ int s_off;
    const int first = blockIdx.x * 4096;                 // this block: 4096 counters of one tile
    const int my_tile = first / TILE_CELLS;
    const bool last = blockIdx.x == gridDim.x - 1;
    const int upto = last ? ntiles : my_tile;            // the last block also needs the grand total
    int part = 0, part_before = 0;
    for (int k = threadIdx.x; k < upto; k += 1024) {
        const int v = tile_sums[k];
        part += v;
        if (k < my_tile) part_before += v;
    }
    // block sums of both partials (wave shuffles, then 16 wave totals)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int a = part_before, b = part;
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    __shared__ int sh2[16];
    if (lane == 0) { sh[wv] = a; sh2[wv] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int ta = 0, tb = 0;
        for (int k = 0; k < 16; ++k) { ta += sh[k]; tb += sh2[k]; }
        s_off = ta;
        if (last) {
            out[n] = tb;
            if (total_out) *total_out = (unsigned long long)tb;
        }
    }
    __syncthreads();
    const int off = s_off;
    const int i = first + threadIdx.x * 4;
    if (off != 0 && i < n) {   // (padded buffers: the int4 may run past n, where out[n] = the grand total must stay)
        int4 q = *reinterpret_cast<int4*>(out + i);
        q.x += off;
        if (i + 1 < n) q.y += off;
        if (i + 2 < n) q.z += off;
        if (i + 3 < n) q.w += off;
        *reinterpret_cast<int4*>(out + i) = q;
    }
}

__global__ __launch_bounds__(256) void k_scatter(int n, const int* __restrict__ cell_of, const int* __restrict__ start,
                                                 int* __restrict__ cell_cnt, int* __restrict__ perm) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int c = cell_of[i];
        if (c >= 0) {
            int slot = atomicSub(&cell_cnt[c], 1) - 1;
            perm[start[c] + slot] = i;
        }
    }
}

// ascending point id inside every cell (cells hold a handful of points)
__global__ __launch_bounds__(256) void k_cellsort(int ncell, const int* __restrict__ start, int* __restrict__ perm) {
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ncell; c += gridDim.x * blockDim.x) {
        int s = start[c], e = start[c + 1];
        for (int a = s + 1; a < e; ++a) {
            int v = perm[a];
            int b = a - 1;
            while (b >= s && perm[b] > v) {
                perm[b + 1] = perm[b];
                --b;
            }
            perm[b + 1] = v;
        }
    }
}
