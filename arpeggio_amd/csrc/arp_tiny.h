// arp_tiny.h — the front half of a pass for a SMALL SELECTION inside a small structure, in ONE launch of ONE block.
//
// The reference's own example is `-s /A/508/` (README.md:55-59; BASELINE configs[0]): a ligand of a few dozen atoms, whose
// selection_plus — everything within 6 A (interactions.py:1420-1424) — is a binding site of a few hundred.  Such a pass was four
// launches, every one a chain of dependent round trips with almost nothing to compute: k_expand_small (8.4 us), k_compact_atoms
// (10.0), k_search (12.0), k_sift_planes (12.7).  The first three have no parallelism to offer at this size, so one block does them
// back to back out of LDS and only the last launch remains:
//   1  _make_selection (interactions.py:1407-1424): every atom against the selected ones (k_expand_small's loop, its test count
//      included), selection and selection_plus kept as bitmaps in LDS;
//   2  the contact grid of the pass (k_compact_atoms' semantics: selection_plus without hydrogens, interactions.py:707-712, as an
//      ordered subset of the cell-ordered static columns; residue tags, cell starts, the columns the per-pair kernel gathers from),
//      tile by tile with the next tile's rows in flight, the kept records also in LDS;
//   3  the ring / amide sets (interactions.py:1433-1437) from the residue tags;
//   4  NeighborSearch.search_all over the kept atoms (interactions.py:1442, 705): all pairs out of LDS — a pair is a candidate when
//      its cells are neighbours (what the grid search tests and counts), a contact pair by the float64 test of Bio.PDB.kdtrees, and
//      enqueued after the residue filters of interactions.py:729-741 in the canonical orientation, exactly as k_search does.
// More kept atoms than TINY_KMAX: the block says so and leaves; the host repeats the pass with the three launches.
#pragma once

#define TINY_THREADS 1024
#define TINY_KMAX 512            // kept atoms (selection_plus without hydrogens) the all-pairs search takes
#define TINY_MAX_ROWS 8192       // atoms of the structure (hydrogens included): eight rows per thread, two bitmaps of 1 KB

struct TinyArgs {
    // 1: k_expand_small's arguments
    int n;
    const float4* xyz;           // uploaded coordinates, by local id
    const int* sel_list;
    int nsel;
    const uint8_t* sel;
    double r2_expand;
    uint8_t* plus;               // out: selection_plus, by local id
    u64* mark_stats;             // C_STAT_MCAND slot line: {tests, atoms gained}
    // 2: k_compact_atoms' arguments (the selection bits come from the bitmaps made in step 1)
    const float4* sp_xyzm;
    const int4* sp_aux;
    const int4* sp_qa;
    const int* sp_h;
    const int* sp_cell;
    GridDesc g;
    uint32_t req, forb;
    float4* s_xyzm;
    int4* s_aux;
    int4* s_qa;
    int* s_h;
    int* s_cell;
    int* start;
    u64* total_out;
    ResMarks rm;
    // 3
    GroupMasks gm;
    // 4: k_search<MODE_CONTACTS>'s arguments
    double r2;
    int include_seq_adj;
    int2* pairs;                 // segment 0 of the pair list
    u64 cap;                     // its capacity
    u64* ctr_pairs;              // head of segment 0
    u64* ctr_cand;
    u64* ctr_acc;
    u64* overflow;               // out: kept atoms when they are more than TINY_KMAX (0 otherwise: the counter block starts as zero)
};

__global__ __launch_bounds__(TINY_THREADS) void k_tiny_front(TinyArgs A) {
    __shared__ uint32_t s_selb[TINY_MAX_ROWS / 32], s_plusb[TINY_MAX_ROWS / 32];
    __shared__ float4 s_sx[SMALL_SEL_MAX];
    __shared__ float4 lx[TINY_KMAX];
    __shared__ int4 la[TINY_KMAX];
    __shared__ int4 lcc[TINY_KMAX];          // cell coordinates of the kept atoms
    __shared__ int s_wtot[TINY_THREADS / 64], s_woff[TINY_THREADS / 64];
    __shared__ int s_total, s_qn;
    __shared__ unsigned long long s_r0[TINY_THREADS / 64], s_r1[TINY_THREADS / 64];
    constexpr int NW = TINY_THREADS / 64;
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = A.n;
    auto block_sum2 = [&](unsigned long long a, unsigned long long b, unsigned long long& ra, unsigned long long& rb) {      // (every thread calls it)
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        __syncthreads();
        if (lane == 0) { s_r0[wv] = a; s_r1[wv] = b; }
        __syncthreads();
        ra = 0; rb = 0;
        for (int k = 0; k < NW; ++k) { ra += s_r0[k]; rb += s_r1[k]; }
    };

    // ---- 1: selection_plus = selection + every atom within the radius of a selected one (k_expand_small) ----
    for (int k = tid; k < TINY_MAX_ROWS / 32; k += TINY_THREADS) { s_selb[k] = 0u; s_plusb[k] = 0u; }
    for (int k = tid; k < A.nsel; k += TINY_THREADS) s_sx[k] = A.xyz[A.sel_list[k]];
    if (tid == 0) s_qn = 0;
    float4 xv[TINY_MAX_ROWS / TINY_THREADS];
    uint8_t sv[TINY_MAX_ROWS / TINY_THREADS];
#pragma unroll
    for (int r = 0; r < TINY_MAX_ROWS / TINY_THREADS; ++r) {      // (all of a thread's atoms asked for at once)
        const int i = r * TINY_THREADS + tid;
        xv[r] = (i < n) ? A.xyz[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        sv[r] = (i < n) ? A.sel[i] : (uint8_t)0;
    }
    __syncthreads();
    unsigned long long tests = 0, gained = 0;
#pragma unroll
    for (int r = 0; r < TINY_MAX_ROWS / TINY_THREADS; ++r) {
        const int i = r * TINY_THREADS + tid;
        if (i >= n) continue;
        const num::d3 x = {(double)xv[r].x, (double)xv[r].y, (double)xv[r].z};
        bool in = sv[r] != 0;
        const bool was = in;
        if (!in)
            for (int k = 0; k < A.nsel; ++k) {
                const float4 q = s_sx[k];
                ++tests;
                if (num::dist2_kd(x, num::d3{(double)q.x, (double)q.y, (double)q.z}) <= A.r2_expand) { in = true; break; }
            }
        A.plus[i] = in ? 1 : 0;
        if (was) atomicOr(&s_selb[i >> 5], 1u << (i & 31));
        if (in) atomicOr(&s_plusb[i >> 5], 1u << (i & 31));
        gained += (in && !was) ? 1u : 0u;
    }
    {
        unsigned long long t_all, g_all;
        block_sum2(tests, gained, t_all, g_all);      // (its barriers also publish the bitmaps)
        if (tid == 0 && A.mark_stats) { atomicAdd(A.mark_stats, t_all); atomicAdd(A.mark_stats + 1, g_all); }
    }

    // ---- 2: the contact grid of the pass (k_compact_atoms), tile by tile, the next tile's rows in flight ----
    struct Row { int4 aux; float4 xyzm; int cell, prev; bool valid; int i; };
    auto load_row = [&](int t) -> Row {
        Row R;
        R.i = t * TINY_THREADS + tid;
        R.valid = R.i < n;
        const int ii = R.valid ? R.i : n - 1;
        R.aux = A.sp_aux[ii];
        R.xyzm = A.sp_xyzm[ii];
        R.cell = A.sp_cell[ii];
        R.prev = (R.i > 0 && R.valid) ? A.sp_cell[R.i - 1] : -1;
        return R;
    };
    const int ntile = (n + TINY_THREADS - 1) / TINY_THREADS;
    int base = 0;
    Row nxt = load_row(0);
    for (int t = 0; t < ntile; ++t) {
        const Row R = nxt;
        if (t + 1 < ntile) nxt = load_row(t + 1);
        const int lid = R.aux.x;
        uint32_t m = __float_as_uint(R.xyzm.w);
        if (R.valid) {      // compose_xyzm with the selection of the moment
            if ((s_selb[lid >> 5] >> (lid & 31)) & 1u) m |= M_SEL;
            if ((s_plusb[lid >> 5] >> (lid & 31)) & 1u) m |= M_PLUS;
            if (A.rm.res_sel) {   // I:1413, 1431
                if (m & M_SEL) A.rm.res_sel[R.aux.y] = A.rm.tag;
                if (m & M_PLUS) A.rm.res_plus[R.aux.y] = A.rm.tag;
            }
        }
        const bool keep = R.valid && ((m & A.req) == A.req) && !(m & A.forb);
        const unsigned long long mk = __ballot(keep);
        const int rank_w = __popcll(mk & ((1ull << lane) - 1ull));
        if (lane == 0) s_wtot[wv] = __popcll(mk);
        __syncthreads();
        if (wv == 0) {
            const int tw = (lane < NW) ? s_wtot[lane] : 0;
            int incl = tw;
#pragma unroll
            for (int off = 1; off < NW; off <<= 1) {
                const int u = __shfl_up(incl, off);
                if (lane >= off) incl += u;
            }
            if (lane < NW) s_woff[lane] = incl - tw;
            if (lane == NW - 1) s_total = incl;
        }
        __syncthreads();
        const int kp = base + s_woff[wv] + rank_w;
        if (keep && kp < TINY_KMAX) {
            float4 v = R.xyzm;
            v.w = __uint_as_float(m);
            A.s_xyzm[kp] = v;
            A.s_aux[kp] = R.aux;
            A.s_qa[kp] = A.sp_qa[R.i];
            A.s_h[kp] = A.sp_h[R.i];
            A.s_cell[kp] = R.cell;
            lx[kp] = v;
            la[kp] = R.aux;
            const int row = R.cell / A.g.nx, cx = R.cell - row * A.g.nx, cz = row / A.g.ny, cy = row - cz * A.g.ny;
            lcc[kp] = make_int4(cx, cy, cz, 0);
        }
        // cell starts: the first row of a cell, the empty cells before it, the cells behind the last row (as k_compact_atoms)
        if (R.valid && R.cell != R.prev) A.start[R.cell] = kp;
        const bool last = R.valid && R.i == n - 1;
        unsigned long long mg = __ballot((R.valid && R.cell - R.prev > 1) || last);
        while (mg) {
            const int l = __ffsll((long long)mg) - 1;
            mg &= mg - 1ull;
            const int lo = __shfl(R.prev, l) + 1, hi = __shfl(R.cell, l), v = __shfl(kp, l);
            for (int c = lo + lane; c < hi; c += 64) A.start[c] = v;
            if (__shfl(last ? 1 : 0, l)) {
                const int tot = v + __shfl(keep ? 1 : 0, l);
                for (int c = hi + 1 + lane; c <= A.g.ncell; c += 64) A.start[c] = tot;
            }
        }
        base += s_total;
        __syncthreads();      // (s_wtot / s_woff / s_total are written again by the next tile)
    }
    const int K = base;
    if (tid == 0 && A.total_out) *A.total_out = (unsigned long long)K;
    if (K > TINY_KMAX) {      // (uniform) too many for the all-pairs search: the host repeats the pass with the three launches
        if (tid == 0) atomicExch(A.overflow, (unsigned long long)K);
        return;
    }

    // ---- 3: ring / amide sets from the residue tags this block has just written ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    group_masks(A.gm, tid, TINY_THREADS);

    // ---- 4: search_all over the kept atoms: thread (b, half) tests atom b against the atoms before it ----
    unsigned long long n_cand = 0, n_acc = 0;
    {
        const int b = tid >> 1, half = tid & 1;
        if (b < K && b > 0) {
            const int a0 = half ? (b >> 1) : 0, a1 = half ? b : (b >> 1);
            const float4 xb = lx[b];
            const int4 ab = la[b], cb = lcc[b];
            const uint32_t mb = __float_as_uint(xb.w);
            const num::d3 pb3 = {(double)xb.x, (double)xb.y, (double)xb.z};
            for (int a = a0; a < a1; ++a) {
                const int4 ca = lcc[a];
                // candidates of the grid search: atoms of the same or of neighbouring cells (each unordered pair once)
                if (abs(ca.x - cb.x) > 1 || abs(ca.y - cb.y) > 1 || abs(ca.z - cb.z) > 1) continue;
                ++n_cand;
                const float4 xa = lx[a];
                // Bio.PDB.kdtrees: float64 d2 <= r2 (k_search's float32 pre-filter decides the same way outside its band)
                if (!(num::dist2_kd(num::d3{(double)xa.x, (double)xa.y, (double)xa.z}, pb3) <= A.r2)) continue;
                ++n_acc;
                const int4 aa = la[a];
                const uint32_t ma = __float_as_uint(xa.w);
                // canonical orientation and the straight-line filters of k_search (stage 2): bgn = lower packed index; interactions.py:729
                // same residue; 733-741 sequence-adjacent residues; ownership
                const bool a_first = aa.x < ab.x;
                const uint32_t m_bgn = a_first ? ma : mb, m_end = a_first ? mb : ma;
                const unsigned adj = min(min((unsigned)(aa.w ^ ab.y), (unsigned)(aa.z ^ ab.y)), min((unsigned)(ab.w ^ aa.y), (unsigned)(ab.z ^ aa.y)));
                const unsigned gate = (A.include_seq_adj ? 0u : 1u) & ((m_end & M_RES_POLY) ? 1u : 0u) & ((ma & mb & M_RES_HASSEQ) ? 1u : 0u);
                const unsigned drop = (aa.y == ab.y ? 1u : 0u) | (gate & (adj == 0u ? 1u : 0u)) | ((m_bgn & M_HOME) ? 0u : 1u);
                if (drop) continue;
                const int slot = atomicAdd(&s_qn, 1);      // (LDS: one block owns the whole list)
                if ((u64)slot < A.cap) A.pairs[slot] = a_first ? make_int2(a, b) : make_int2(b, a);
            }
        }
    }
    {
        unsigned long long c_all, a_all;
        block_sum2(n_cand, n_acc, c_all, a_all);
        if (tid == 0) {
            atomicAdd(A.ctr_cand, c_all);
            atomicAdd(A.ctr_acc, a_all);
            if (s_qn > 0) atomicAdd(A.ctr_pairs, (unsigned long long)s_qn);
        }
    }
}
