// arp_prepare.h — the geometric part of InteractionComplex.initialize(): what turns perceived
// rings / amide groups (atom index lists) into the centres, normals and ring residues the contact kernels read.
//   k_ring_geometry   _perceive_rings                       interactions.py:1697-1733
//   k_amide_geometry  _perceive_amide_groups                interactions.py:1531-1589
//   k_ring_residue    _assign_aromatic_rings_to_residues    interactions.py:1453-1492
// Ring / amide PERCEPTION (SSSR, aromaticity, the amide SMARTS) is OpenBabel's and stays out of scope; these kernels
// start from the atom lists it yields.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "arp_planes.h"

// OBRing::findCenterAndNormal (OpenBabel ring.cpp — third party, not in /root/reference; restated from its published
// source): centre = mean of the ring atoms' vectors; normal = sum over consecutive atoms of (v_j - centre) x
// (v_j+1 - centre), divided by the ring size, then normalised (left alone when its length is ~0).  All float64 on the
// float32 atom coordinates (OpenBabel holds its own float64 copy parsed from the same file text).
__global__ __launch_bounds__(256) void k_ring_geometry(int nring, const int* __restrict__ off, const int* __restrict__ idx,
                                                       const float4* __restrict__ xyz, double* __restrict__ center,
                                                       double* __restrict__ normal) {
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < nring; r += gridDim.x * blockDim.x) {
        const int a0 = off[r], na = off[r + 1] - a0;
        double cx = 0, cy = 0, cz = 0;
        for (int j = 0; j < na; ++j) {
            const float4 v = xyz[idx[a0 + j]];
            cx += (double)v.x; cy += (double)v.y; cz += (double)v.z;
        }
        const double inv_n = 1.0 / (double)na;   // vector3::operator/= multiplies by the reciprocal
        cx *= inv_n; cy *= inv_n; cz *= inv_n;
        double nx = 0, ny = 0, nz = 0;
        for (int j = 0; j < na; ++j) {
            const float4 p = xyz[idx[a0 + j]], q = xyz[idx[a0 + ((j + 1 == na) ? 0 : j + 1)]];
            const double ax = (double)p.x - cx, ay = (double)p.y - cy, az = (double)p.z - cz;
            const double bx = (double)q.x - cx, by = (double)q.y - cy, bz = (double)q.z - cz;
            nx += ay * bz - az * by;   // cross(v1, v2), vector3.cpp
            ny += az * bx - ax * bz;
            nz += ax * by - ay * bx;
        }
        nx *= inv_n; ny *= inv_n; nz *= inv_n;
        const double l = sqrt(nx * nx + ny * ny + nz * nz);
        if (!(fabs(l) < 2e-6)) {   // vector3::normalize: IsNearZero(length) -> unchanged
            const double inv_l = 1.0 / l;
            nx *= inv_l; ny *= inv_l; nz *= inv_l;
        }
        center[3 * (size_t)r] = cx; center[3 * (size_t)r + 1] = cy; center[3 * (size_t)r + 2] = cz;
        normal[3 * (size_t)r] = nx; normal[3 * (size_t)r + 1] = ny; normal[3 * (size_t)r + 2] = nz;
    }
}

// I:1566-1580: atoms = [N, C, O, C-alpha]; bond centroid = (C + N) / 2.0 in float32 (cn.sum(0) / float(len(cn)), I:1569);
// normal = last right-singular vector of the centred C, O, N coordinates (np.linalg.svd, float32).  Three points
// always lie in a plane: that vector is the unit normal of the plane, i.e. the normalised cross product of two
// centred rows; the sign LAPACK happens to return is not reproduced (every consumer folds the angle, U:656-660), and
// the last bits differ from an SVD's (agreement ~1e-6 in the components, tested against numpy).
__global__ __launch_bounds__(256) void k_amide_geometry(int namide, const int* __restrict__ atoms, const float4* __restrict__ xyz,
                                                        float* __restrict__ center, float* __restrict__ normal) {
    for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < namide; a += gridDim.x * blockDim.x) {
        const float4 N = xyz[atoms[4 * a]], Cc = xyz[atoms[4 * a + 1]], O = xyz[atoms[4 * a + 2]];
        center[3 * (size_t)a] = (Cc.x + N.x) / 2.0f;
        center[3 * (size_t)a + 1] = (Cc.y + N.y) / 2.0f;
        center[3 * (size_t)a + 2] = (Cc.z + N.z) / 2.0f;
        // centred rows (amide centroid = mean of C, O, N, I:1568) in float64 for a well-conditioned cross product
        const double mx = ((double)Cc.x + O.x + N.x) / 3.0, my = ((double)Cc.y + O.y + N.y) / 3.0, mz = ((double)Cc.z + O.z + N.z) / 3.0;
        const double ux = Cc.x - mx, uy = Cc.y - my, uz = Cc.z - mz;
        const double vx = O.x - mx, vy = O.y - my, vz = O.z - mz;
        double nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
        const double l = sqrt(nx * nx + ny * ny + nz * nz);
        if (l > 0) { nx /= l; ny /= l; nz /= l; }
        normal[3 * (size_t)a] = (float)nx; normal[3 * (size_t)a + 1] = (float)ny; normal[3 * (size_t)a + 2] = (float)nz;
    }
}

// I:1460-1492: one wavefront per ring; the atoms of the 27 cells around the centre (all atoms, hydrogens included:
// the tree is built on s_atoms, I:1455); membership = the KD-tree's inclusive float64 test at 3.0 A; distance =
// np.linalg.norm(atom.coord - centre) (float64, I:1471); strict '<' keeps the first of equal distances, which in the
// reference is the KD-tree's delivery order — here the lowest packed atom index.  ring_res = residue of that atom,
// -1 when no atom is that close (I:1476-1479).
__global__ __launch_bounds__(256) void k_ring_residue(GridDesc g, const int* __restrict__ start, const float4* __restrict__ s_xyzm,
                                                      const int4* __restrict__ s_aux, int nring, const double* __restrict__ ring_c,
                                                      int* __restrict__ ring_res, double* __restrict__ ring_dist) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < nring; r += nwave) {
        const num::d3 ctr_ = ld3(ring_c, r);
        const Stencil st = stencil_load(g, start, cell_box(g, ctr_), lane);
        double best = 1e300;
        int best_lid = 0x7FFFFFFF, best_res = -1;
        for (int kb = 0; kb < st.pre[9]; kb += 64) {
            const int k = kb + lane;
            if (k < st.pre[9]) {
                const int j = stencil_pos(st, k);
                const float4 v = s_xyzm[j];
                const num::d3 x = {(double)v.x, (double)v.y, (double)v.z};
                if (num::dist2_kd(ctr_, x) <= 9.0) {
                    const double d = num::norm(num::sub(x, ctr_));
                    const int4 a = s_aux[j];
                    if (d < best || (d == best && a.x < best_lid)) { best = d; best_lid = a.x; best_res = a.y; }
                }
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const double od = __shfl_xor(best, o);
            const int ol = __shfl_xor(best_lid, o), orr = __shfl_xor(best_res, o);
            if (od < best || (od == best && ol < best_lid)) { best = od; best_lid = ol; best_res = orr; }
        }
        if (lane == 0) {
            ring_res[r] = best_res;
            if (ring_dist) ring_dist[r] = (best_res >= 0) ? best : -1.0;
        }
    }
}

// Bio.PDB.NeighborSearch.search(center, radius) for many centres at once (the reference asks for one at a time, I:960, 1463):
// one wavefront per centre, the 27 cells around it (cell edge >= radius), membership = the KD-tree's inclusive float64 test;
// hits {centre, atom local id} are compacted per wave and appended with one atomicAdd.
__global__ __launch_bounds__(256) void k_center_search(GridDesc g, const int* __restrict__ start, const float4* __restrict__ s_xyzm,
                                                       const int4* __restrict__ s_aux, int ncenter, const double* __restrict__ centers,
                                                       double r2, int2* __restrict__ out, unsigned long long cap,
                                                       unsigned long long* __restrict__ count) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (gridDim.x * blockDim.x) >> 6;
    for (int c = wave; c < ncenter; c += nwave) {
        const num::d3 ctr_ = ld3(centers, c);
        const Stencil st = stencil_load(g, start, cell_box(g, ctr_), lane);
        for (int kb = 0; kb < st.pre[9]; kb += 64) {
            const int k = kb + lane;
            bool hit = false;
            int lid = 0;
            if (k < st.pre[9]) {
                const int j = stencil_pos(st, k);
                const float4 v = s_xyzm[j];
                hit = num::dist2_kd(ctr_, num::d3{(double)v.x, (double)v.y, (double)v.z}) <= r2;
                if (hit) lid = s_aux[j].x;
            }
            const unsigned long long m = __ballot(hit);
            if (!m) continue;
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(count, (unsigned long long)__popcll(m));
            base = __shfl(base, 0);
            const unsigned long long slot = base + __popcll(m & ((1ull << lane) - 1ull));
            if (hit && slot < cap) out[slot] = make_int2(c, lid);
        }
    }
}
