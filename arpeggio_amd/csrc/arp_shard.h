// arp_shard.h — halo records cut out and merged on the device (the reference is single-process).
//
// Run once per structure and rank, not per pass: clarity over tuning.  Three steps, each a handful of launches:
//   face    home records with x in [x_lo, x_hi]  ->  a record buffer for the neighbour (flags, exclusive scans, copy)
//   merge   home + left + right record buffers   ->  positions in ascending global id (three sorted lists: the place of
//                                                    an element is its own index plus its lower bounds in the other two)
//   fill    the arrays of a blob (arp_blob_header layout) + ownership, origin, single-bond-neighbour coordinates
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/arpeggio_hip.h"

struct RecList {          // one record buffer as the kernels see it
    const arp_rec_atom* a; const double* h; const int* b; const arp_rec_ring* r; const arp_rec_amide* m;
    int na, nh, nb, nr, nm;
};
struct RecLists { RecList l[3]; };   // 0 home, 1 from the left neighbour, 2 from the right neighbour

// ---- exclusive scans: segment s of `seg` independent int arrays (each n[s] + 1 long: out[n] = total), one block each.
// A block walks its array in chunks of 4096 with a running carry — tens of microseconds for a slab, once per structure.
struct ScanSegs { int* p[5]; int n[5]; };
__global__ __launch_bounds__(1024) void k_scan_segments(ScanSegs S) {
    __shared__ int sh[17];
    int* const a = S.p[blockIdx.x];
    const int n = S.n[blockIdx.x];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int carry = 0;
    for (int base = 0; base < n; base += 4096) {
        const int i = base + 4 * threadIdx.x;
        int v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (i + k < n) ? a[i + k] : 0;
        const int sum = v[0] + v[1] + v[2] + v[3];
        int incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) sh[wv] = incl;
        __syncthreads();
        if (threadIdx.x == 0) {
            int run = 0;
            for (int k = 0; k < 16; ++k) { const int t = sh[k]; sh[k] = run; run += t; }
            sh[16] = run;
        }
        __syncthreads();
        int run = carry + sh[wv] + incl - sum;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i + k < n) a[i + k] = run;
            run += v[k];
        }
        carry += sh[16];
        __syncthreads();
    }
    if (threadIdx.x == 0) a[n] = carry;
}

// ---- face ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_face_flags(RecList H, double x_lo, double x_hi, int* __restrict__ fa, int* __restrict__ fh,
                                                    int* __restrict__ fb, int* __restrict__ fr, int* __restrict__ fm) {
    const int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H.na; i += stride) {
        const arp_rec_atom& r = H.a[i];
        const double x = (double)r.x;
        const bool in = x >= x_lo && x <= x_hi;
        fa[i] = in ? 1 : 0;
        fh[i] = in ? r.h_cnt : 0;
        fb[i] = in ? r.bond_cnt : 0;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H.nr; i += stride) {
        const double x = H.r[i].c[0];
        fr[i] = (x >= x_lo && x <= x_hi) ? 1 : 0;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H.nm; i += stride) {
        const double x = (double)H.m[i].c[0];
        fm[i] = (x >= x_lo && x <= x_hi) ? 1 : 0;
    }
}

// after the scans: fa[i + 1] > fa[i] <=> record i is in the face, and fa[i] is its slot
__global__ __launch_bounds__(256) void k_face_write(RecList H, const int* __restrict__ fa, const int* __restrict__ fh,
                                                    const int* __restrict__ fb, const int* __restrict__ fr, const int* __restrict__ fm,
                                                    arp_rec_atom* __restrict__ oa, double* __restrict__ oh, int* __restrict__ ob,
                                                    arp_rec_ring* __restrict__ orr, arp_rec_amide* __restrict__ om) {
    const int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H.na; i += stride) {
        if (fa[i + 1] == fa[i]) continue;
        arp_rec_atom r = H.a[i];
        const double* hs = H.h + 3 * (size_t)r.h_start;
        const int* bs = H.b + r.bond_start;
        r.h_start = fh[i];
        r.bond_start = fb[i];
        oa[fa[i]] = r;
        for (int k = 0; k < 3 * r.h_cnt; ++k) oh[3 * (size_t)r.h_start + k] = hs[k];
        for (int k = 0; k < r.bond_cnt; ++k) ob[r.bond_start + k] = bs[k];
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H.nr; i += stride)
        if (fr[i + 1] != fr[i]) orr[fr[i]] = H.r[i];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H.nm; i += stride)
        if (fm[i + 1] != fm[i]) om[fm[i]] = H.m[i];
}

// ---- merge ---------------------------------------------------------------------------------------------------
template <class R>
__device__ inline int lower_bound_gid(const R* __restrict__ a, int n, int gid, bool* found) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid].gid < gid) lo = mid + 1; else hi = mid;
    }
    if (lo < n && a[lo].gid == gid) *found = true;
    return lo;
}
// merged position of the atom with this global id, -1 when it is in none of the lists
__device__ inline int atom_position(const RecLists& L, int gid) {
    bool f = false;
    int pos = 0;
    for (int s = 0; s < 3; ++s) pos += lower_bound_gid(L.l[s].a, L.l[s].na, gid, &f);
    return f ? pos : -1;
}

// one thread per atom of any list: where it goes (src[pos] = {list, index}), how many hydrogens and LOCAL bonds it brings
__global__ __launch_bounds__(256) void k_merge_positions(RecLists L, int2* __restrict__ src, int* __restrict__ hcnt, int* __restrict__ bcnt,
                                                         int* __restrict__ err) {
    const int total = L.l[0].na + L.l[1].na + L.l[2].na;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        int s = 0, k = t;
        if (k >= L.l[0].na) { k -= L.l[0].na; s = 1; if (k >= L.l[1].na) { k -= L.l[1].na; s = 2; } }
        const arp_rec_atom& r = L.l[s].a[k];
        int pos = k;
        bool dup = (k > 0 && L.l[s].a[k - 1].gid >= r.gid);           // the lists must be strictly ascending
        for (int o = 0; o < 3; ++o)
            if (o != s) pos += lower_bound_gid(L.l[o].a, L.l[o].na, r.gid, &dup);
        if (dup || r.h_cnt < 0 || r.bond_cnt < 0 || r.h_start < 0 || r.bond_start < 0 || r.h_start > L.l[s].nh - r.h_cnt ||
            r.bond_start > L.l[s].nb - r.bond_cnt) {
            atomicOr(err, 1);
            continue;
        }
        int nb = 0;
        const int* bs = L.l[s].b + r.bond_start;
        for (int q = 0; q < r.bond_cnt; ++q) nb += atom_position(L, bs[q]) >= 0 ? 1 : 0;
        src[pos] = make_int2(s, k);
        hcnt[pos] = r.h_cnt;
        bcnt[pos] = nb;
    }
}

struct MergeOut {
    int n, nres, n_rad;
    float4* xyz; double2* rad; uint16_t* tmask; uint16_t* flags; int* res_id;
    uint8_t* res_flags; int* res_prev; int* res_next;
    int* bond_off; int* bond_idx; int* h_off; double* h_xyz; int* sb_nbr; uint16_t* rad_idx; const double2* rad_tab;
    float4* sb; int* gid; uint8_t* home; int8_t* origin; uint8_t* sel;
};
__global__ __launch_bounds__(256) void k_init_residues(int nres, uint8_t* __restrict__ fl, int* __restrict__ prev, int* __restrict__ next) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nres; i += gridDim.x * blockDim.x) { fl[i] = 0; prev[i] = -1; next[i] = -1; }
}
__global__ __launch_bounds__(256) void k_merge_fill(RecLists L, MergeOut O, const int2* __restrict__ src, const int* __restrict__ hoff,
                                                    const int* __restrict__ boff, int* __restrict__ err) {
    __shared__ unsigned long long s_tab[512];
    for (int k = threadIdx.x; k < 2 * O.n_rad; k += blockDim.x) s_tab[k] = reinterpret_cast<const unsigned long long*>(O.rad_tab)[k];
    __syncthreads();
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < O.n; p += gridDim.x * blockDim.x) {
        const int2 sk = src[p];
        const arp_rec_atom r = L.l[sk.x].a[sk.y];
        O.xyz[p] = make_float4(r.x, r.y, r.z, 0.0f);
        O.rad[p] = make_double2(r.vdw, r.cov);
        O.tmask[p] = r.tmask; O.flags[p] = r.flags;
        O.res_id[p] = r.res_gid;
        if (r.res_gid < 0 || r.res_gid >= O.nres) atomicOr(err, 2);
        else { O.res_flags[r.res_gid] = r.res_flags; O.res_prev[r.res_gid] = r.res_prev; O.res_next[r.res_gid] = r.res_next; }
        O.h_off[p] = hoff[p]; O.bond_off[p] = boff[p];
        const double* hs = L.l[sk.x].h + 3 * (size_t)r.h_start;
        for (int k = 0; k < 3 * r.h_cnt; ++k) O.h_xyz[3 * (size_t)hoff[p] + k] = hs[k];
        const int* bs = L.l[sk.x].b + r.bond_start;
        int w = boff[p];
        for (int q = 0; q < r.bond_cnt; ++q) {
            const int partner = atom_position(L, bs[q]);
            if (partner >= 0) O.bond_idx[w++] = partner;
        }
        O.sb_nbr[p] = -1;
        O.sb[p] = r.sb_has ? make_float4(r.sb_x, r.sb_y, r.sb_z, 1.0f) : make_float4(0, 0, 0, 0);
        const unsigned long long kv = (unsigned long long)__double_as_longlong(r.vdw), kc = (unsigned long long)__double_as_longlong(r.cov);
        int idx = 0xFFFF;
        for (int k = 0; k < O.n_rad; ++k)
            if (s_tab[2 * k] == kv && s_tab[2 * k + 1] == kc) { idx = k; break; }
        O.rad_idx[p] = (uint16_t)idx;
        O.gid[p] = r.gid;
        O.home[p] = sk.x == 0 ? 1 : 0;
        O.origin[p] = sk.x == 0 ? 0 : (sk.x == 1 ? -1 : 1);
        O.sel[p] = r.sel;
        if (p == O.n - 1) { O.h_off[O.n] = hoff[O.n]; O.bond_off[O.n] = boff[O.n]; }
    }
    if (O.n == 0 && blockIdx.x == 0 && threadIdx.x == 0) { O.h_off[0] = 0; O.bond_off[0] = 0; }
}

struct GroupOut {
    double* ring_c; double* ring_n; int* ring_res; int* ring_gid; uint8_t* ring_home; int8_t* ring_origin;
    float* am_c; float* am_n; int* am_res; int* am_gid; uint8_t* am_home; int8_t* am_origin;
};
__global__ __launch_bounds__(256) void k_merge_groups(RecLists L, GroupOut G, int* __restrict__ err) {
    const int stride = gridDim.x * blockDim.x;
    const int tr = L.l[0].nr + L.l[1].nr + L.l[2].nr;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < tr; t += stride) {
        int s = 0, k = t;
        if (k >= L.l[0].nr) { k -= L.l[0].nr; s = 1; if (k >= L.l[1].nr) { k -= L.l[1].nr; s = 2; } }
        const arp_rec_ring r = L.l[s].r[k];
        int pos = k;
        bool dup = (k > 0 && L.l[s].r[k - 1].gid >= r.gid);
        for (int o = 0; o < 3; ++o)
            if (o != s) pos += lower_bound_gid(L.l[o].r, L.l[o].nr, r.gid, &dup);
        if (dup) { atomicOr(err, 4); continue; }
        for (int q = 0; q < 3; ++q) { G.ring_c[3 * pos + q] = r.c[q]; G.ring_n[3 * pos + q] = r.n[q]; }
        G.ring_res[pos] = r.res; G.ring_gid[pos] = r.gid; G.ring_home[pos] = s == 0; G.ring_origin[pos] = s == 0 ? 0 : (s == 1 ? -1 : 1);
    }
    const int tm = L.l[0].nm + L.l[1].nm + L.l[2].nm;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < tm; t += stride) {
        int s = 0, k = t;
        if (k >= L.l[0].nm) { k -= L.l[0].nm; s = 1; if (k >= L.l[1].nm) { k -= L.l[1].nm; s = 2; } }
        const arp_rec_amide r = L.l[s].m[k];
        int pos = k;
        bool dup = (k > 0 && L.l[s].m[k - 1].gid >= r.gid);
        for (int o = 0; o < 3; ++o)
            if (o != s) pos += lower_bound_gid(L.l[o].m, L.l[o].nm, r.gid, &dup);
        if (dup) { atomicOr(err, 8); continue; }
        for (int q = 0; q < 3; ++q) { G.am_c[3 * pos + q] = r.c[q]; G.am_n[3 * pos + q] = r.n[q]; }
        G.am_res[pos] = r.res; G.am_gid[pos] = r.gid; G.am_home[pos] = s == 0; G.am_origin[pos] = s == 0 ? 0 : (s == 1 ? -1 : 1);
    }
}
