/*
 * arpeggio_hip.h — C ABI of the MI355X-native contact-detection hot path.
 *
 * This is the drop-in boundary for arpeggio's `InteractionComplex.run_arpeggio`
 * (reference: arpeggio/core/interactions.py:329-347).  The reference is pure
 * Python with no FFI of its own, so every entry point below names the reference
 * function it replaces (file:line, relative to the reference tree; `I:` =
 * arpeggio/core/interactions.py, `U:` = arpeggio/core/utils.py, `C:` =
 * arpeggio/core/config.py).  INTEGRATION.md shows the ctypes binding a
 * maintainer would add on the reference side.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - All host buffers are caller-allocated; the library never keeps a host
 *     pointer after a call returns.  Device memory is owned by the context.
 *   - One HIP stream per context; calls are blocking unless named *_launch.
 *   - Return value: ARP_OK or a negative ARP_E_* code; arp_last_error() gives
 *     the message.  ARP_E_CAPACITY stores the required element count in *count
 *     so the caller can re-call with a larger buffer.
 *   - Atom indices in every output are indices into the arrays given to
 *     arp_set_atoms (the "packed atom order").  bgn = lower packed index
 *     (canonical orientation; the reference's own orientation is a hash/KD-tree
 *     artefact).
 */
#ifndef ARPEGGIO_HIP_H
#define ARPEGGIO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes -------------------------------------------------------- */
#define ARP_OK            0
#define ARP_E_ARG        -1   /* bad argument / call order */
#define ARP_E_HIP        -2   /* HIP runtime error */
#define ARP_E_CAPACITY   -3   /* output buffer too small; *count = required */
#define ARP_E_XBOND_NBR  -4   /* U:173 dereferenced None: xbond donor without a
                                 single-bond heavy neighbour (reference raises
                                 AttributeError) */
#define ARP_E_NOMEM      -5

/* ---- atom type mask: bit = position in this list (keys of C:53-145) ------ */
#define ARP_T_HBOND_ACCEPTOR       (1u << 0)
#define ARP_T_HBOND_DONOR          (1u << 1)
#define ARP_T_XBOND_ACCEPTOR       (1u << 2)
#define ARP_T_XBOND_DONOR          (1u << 3)
#define ARP_T_WEAK_HBOND_ACCEPTOR  (1u << 4)
#define ARP_T_WEAK_HBOND_DONOR     (1u << 5)
#define ARP_T_POS_IONISABLE        (1u << 6)
#define ARP_T_NEG_IONISABLE        (1u << 7)
#define ARP_T_HYDROPHOBE           (1u << 8)
#define ARP_T_CARBONYL_OXYGEN      (1u << 9)
#define ARP_T_CARBONYL_CARBON      (1u << 10)
#define ARP_T_AROMATIC             (1u << 11)

/* ---- per-atom flags (u16) ------------------------------------------------ */
#define ARP_F_METAL     (1u << 0)   /* I:1990  element in C:27-31            */
#define ARP_F_HALOGEN   (1u << 1)   /* I:1991  element in C:33               */
#define ARP_F_WATER     (1u << 2)   /* get_full_id()[3][0] == 'W' (I:678)    */
#define ARP_F_HYDROGEN  (1u << 3)   /* element.strip() == 'H' (I:712, I:964) */
#define ARP_F_ELEM_C    (1u << 4)   /* element == 'C' (I:1009)               */
#define ARP_F_ELEM_S    (1u << 5)   /* element == 'S' (I:1023)               */
#define ARP_F_RES_MET   (1u << 6)   /* parent resname == 'MET' (I:1023)      */

/* ---- per-residue flags (u8) ---------------------------------------------- */
#define ARP_R_POLYPEPTIDE (1u << 0) /* residue.is_polypeptide (I:1671,1860)  */
#define ARP_R_HAS_SEQ     (1u << 1) /* hasattr(prev_residue/next_residue) (I:736) */

/* ---- atom-atom SIFt bits (order of the name list at I:178-180) ----------- */
#define ARP_S_CLASH         (1u << 0)
#define ARP_S_COVALENT      (1u << 1)
#define ARP_S_VDW_CLASH     (1u << 2)
#define ARP_S_VDW           (1u << 3)
#define ARP_S_PROXIMAL      (1u << 4)
#define ARP_S_HBOND         (1u << 5)
#define ARP_S_WEAK_HBOND    (1u << 6)
#define ARP_S_XBOND         (1u << 7)
#define ARP_S_IONIC         (1u << 8)
#define ARP_S_METAL_COMPLEX (1u << 9)
#define ARP_S_AROMATIC      (1u << 10)
#define ARP_S_HYDROPHOBIC   (1u << 11)
#define ARP_S_CARBONYL      (1u << 12)
#define ARP_S_POLAR         (1u << 13)
#define ARP_S_WEAK_POLAR    (1u << 14)

/* ---- interacting-entities codes (I:643-691, I:985-997, I:1095-1108) ------ */
#define ARP_CT_INTRA_NON_SELECTION  0
#define ARP_CT_INTRA_SELECTION      1
#define ARP_CT_INTER                2
#define ARP_CT_SELECTION_WATER      3
#define ARP_CT_NON_SELECTION_WATER  4
#define ARP_CT_WATER_WATER          5
#define ARP_CT_INTRA_BINDING_SITE   6

/* ---- atom-plane interaction bits (I:1007-1024; alphabetical = bit order) -- */
#define ARP_AP_CARBONPI      (1u << 0)
#define ARP_AP_CATIONPI      (1u << 1)
#define ARP_AP_DONORPI       (1u << 2)
#define ARP_AP_HALOGENPI     (1u << 3)
#define ARP_AP_METSULPHURPI  (1u << 4)

/* ---- plane-plane classes (I:1127-1148); 9 = '' (NaN angles) */
#define ARP_PP_FF 0
#define ARP_PP_OF 1
#define ARP_PP_EE 2
#define ARP_PP_FT 3
#define ARP_PP_OT 4
#define ARP_PP_ET 5
#define ARP_PP_FE 6
#define ARP_PP_OE 7
#define ARP_PP_EF 8
#define ARP_PP_NONE 9
#define ARP_PP_SAME 254      /* reverse visit happened and produced the same class (nothing appended) */
#define ARP_PP_SKIPPED 255   /* reverse visit dropped by the intra-residue EE rule (I:1154) or never made */

typedef struct arp_ctx arp_ctx;

/* ---- lifetime ------------------------------------------------------------ */
const char* arp_version(void);
/* device = HIP device ordinal.  Fails with ARP_E_HIP when no gfx950 device is
 * usable: there is no CPU fallback in this library. */
/* GPUs this process can see (0: none, or no HIP runtime) */
int  arp_device_count(void);
/* Waits until everything enqueued on the context's device has completed (every stream of every context on it): the
 * bracket of a timed region in a host program that has no HIP runtime of its own to ask (bench.py). */
int  arp_device_synchronize(arp_ctx* ctx);
int  arp_create(int device, arp_ctx** out);
void arp_destroy(arp_ctx* ctx);
const char* arp_last_error(arp_ctx* ctx);   /* ctx may be NULL: last create error */

/* ---- inputs: what InteractionComplex.initialize() (I:288-327) leaves behind */
/* Atoms = self.s_atoms (I:54-60).  xyz: atom.coord float32 (P:327).
 * vdw/cov: atom.vdw_radius / atom.cov_radius, Python floats (I:1501,1509).
 * type_mask: atom.atom_types (I:1959-1983).  flags: ARP_F_*.
 * res_id: index of atom.get_parent() in the residue table. */
int arp_set_atoms(arp_ctx* ctx, int64_t n, const float* xyz, const double* vdw,
                  const double* cov, const uint16_t* type_mask,
                  const uint16_t* flags, const int32_t* res_id);
/* Residue table: is_polypeptide / prev_residue / next_residue (I:1671-1693);
 * prev/next are residue indices, -1 = None. */
int arp_set_residues(arp_ctx* ctx, int64_t nres, const uint8_t* res_flags,
                     const int32_t* prev, const int32_t* next);
/* OpenBabel bond graph as CSR over packed atom indices: the set iterated by
 * ob.OBAtomAtomIter(ob_atom_bgn) (I:750). */
int arp_set_bonds(arp_ctx* ctx, const int32_t* bond_off, const int32_t* bond_idx);
/* atom.h_coords (I:1513-1529): CSR, float64 xyz per hydrogen. */
int arp_set_hydrogens(arp_ctx* ctx, const int32_t* h_off, const double* h_xyz);
/* utils.get_single_bond_neighbour (U:612-635) resolved to a packed atom index,
 * -1 = None. */
int arp_set_single_bond_neighbours(arp_ctx* ctx, const int32_t* sb_nbr);
/* biopython_str.rings (I:1697-1733, residue from I:1453-1492; -1 = None). */
int arp_set_rings(arp_ctx* ctx, int64_t nring, const double* center,
                  const double* normal, const int32_t* ring_res);
/* biopython_str.amides (I:1531-1589): float32 centre and normal. */
int arp_set_amides(arp_ctx* ctx, int64_t namide, const float* center,
                   const float* normal, const int32_t* amide_res);

/* ---- the same inputs as ONE blob --------------------------------------------------------
 * Everything arp_set_atoms ... arp_set_amides take, laid out device-ready in one contiguous host
 * buffer (ideally page-locked: arp_host_alloc) and uploaded with a single asynchronous copy.
 * The packer of a structure (what the reference's initialize(), I:288-327, leaves behind) writes
 * straight into the views arp_blob_layout describes; nothing is converted or copied again on the
 * way to the GPU.  Arrays, in the order of off[]:
 *   0 xyz4      float[4n]   x, y, z, 0 per atom (P:327)            11 h_xyz     double[3 nh]   (I:1529)
 *   1 rad       double[2n]  vdw, cov per atom (I:1501,1509)        12 sb_nbr    int32[n]       (U:612-635, -1 = None)
 *   2 type_mask uint16[n]   ARP_T_*                                13 ring_c    double[3 nring]
 *   3 flags     uint16[n]   ARP_F_*                                14 ring_n    double[3 nring]
 *   4 res_id    int32[n]                                           15 ring_res  int32[nring]   (-1 = None)
 *   5 res_flags uint8[nres] ARP_R_*                                16 amide_c   float[3 namide]
 *   6 res_prev  int32[nres]                                        17 amide_n   float[3 namide]
 *   7 res_next  int32[nres]                                        18 amide_res int32[namide]
 *   8 bond_off  int32[n+1]  CSR (I:750)                            19 rad_idx   uint16[n]  index into rad_tab, 0xFFFF = not in it
 *   9 bond_idx  int32[nbond]                                       20 rad_tab   double[2 * 256]  distinct {vdw, cov} pairs
 *  10 h_off     int32[n+1]  CSR (I:1513-1529)
 * lo/hi, ring_lo/hi, amide_lo/hi = bounding boxes of the atom coordinates / ring centres / amide centres (the grids are
 * sized from them; arp_set_blob verifies that every point lies inside).  n_rad = entries of rad_tab in use.
 * rad[i] MUST equal rad_tab[rad_idx[i]] bit for bit wherever rad_idx[i] != 0xFFFF (arp_blob_fill writes them so): from
 * 32 768 atoms on, when every atom is in the table, arp_set_blob leaves array 1 on the host and the device writes the
 * per-atom radii from the table — a blob that disagrees with itself would be evaluated with the table's values. */
#define ARP_BLOB_MAGIC 0x31424F4C42505241ull   /* "ARPBLOB1" */
#define ARP_BLOB_ARRAYS 21
typedef struct arp_blob_header {
    uint64_t magic, bytes;
    int64_t n, nres, nbond, nh, nring, namide, n_rad;
    uint64_t off[ARP_BLOB_ARRAYS];      /* byte offsets from the start of the blob, 16-byte aligned */
    double lo[3], hi[3], ring_lo[3], ring_hi[3], amide_lo[3], amide_hi[3];
} arp_blob_header;
/* bytes needed for a blob with these counts (header included); 0 on bad counts */
uint64_t arp_blob_size(int64_t n, int64_t nres, int64_t nbond, int64_t nh, int64_t nring, int64_t namide);
/* writes the header (magic, counts, offsets) at the start of `blob`; the caller then fills the arrays, the boxes and n_rad */
int arp_blob_layout(void* blob, uint64_t bytes, int64_t n, int64_t nres, int64_t nbond, int64_t nh, int64_t nring, int64_t namide);
/* Replaces every input of the context with the blob's.  One asynchronous host-to-device copy on the context's stream,
 * then a device-side check of what the classic setters check on the host (finite coordinates inside the box, index
 * ranges, CSR offsets) that the call waits for: ARP_E_ARG if the blob fails it.  The blob may be reused or freed when
 * the call returns.  Selection and ownership are reset (whole structure, I:1395).
 * With its own streams (no caller-owned one) the context also builds what depends on the uploaded arrays only — the 6 A grids of
 * the ring / amide centres and the candidate lists of the ring / amide loops (I:938-1382) — on its second stream beside the
 * check; they may still be running when the call returns, and every later call that reads or replaces what they read or
 * write waits for them on the device (nothing for the caller to do). */
int arp_set_blob(arp_ctx* ctx, const void* blob, uint64_t bytes);
/* Host-only packer: fills every array of a blob whose header arp_blob_layout has written from the arrays the classic
 * setters take (same types and meanings; xyz is float[3 * n], hydrogen coordinates double[3 * nh], ...), builds the
 * dictionary of distinct {vdw, cov} pairs (compared bit for bit; the 256 most frequent if there are more) and the
 * three bounding boxes.  Any pointer whose count is zero may be NULL. */
int arp_blob_fill(void* blob, uint64_t bytes, const float* xyz, const double* vdw, const double* cov, const uint16_t* type_mask,
                  const uint16_t* flags, const int32_t* res_id, const uint8_t* res_flags, const int32_t* res_prev,
                  const int32_t* res_next, const int32_t* bond_off, const int32_t* bond_idx, const int32_t* h_off,
                  const double* h_xyz, const int32_t* sb_nbr, const double* ring_center, const double* ring_normal,
                  const int32_t* ring_res, const float* amide_center, const float* amide_normal, const int32_t* amide_res);

/* ---- Bio.PDB.NeighborSearch equivalents (I:707,960,1394,1420,1442) -------- */
/* NeighborSearch(atoms).search_all(radius): every unordered pair with
 * float64 d^2 <= radius^2, once, as (i<j).  active = optional u8[n] mask of the
 * atoms the tree is built on (NULL = all atoms). */
int arp_search_all(arp_ctx* ctx, double radius, const uint8_t* active,
                   int64_t cap, int32_t* out_i, int32_t* out_j, int64_t* count);

/* NeighborSearch(atoms).search(center, radius) (I:960, 1463) for ncenters centres at once: every (centre, atom) with
 * float64 d^2 <= radius^2 over ALL atoms of the structure (hydrogens included, as the tree of I:1394 / 1455 holds them).
 * centers = double[3 * ncenters].  Output sorted by (centre, atom index).  ARP_E_CAPACITY with the required count in
 * *count when cap is too small. */
int arp_search(arp_ctx* ctx, double radius, int64_t ncenters, const double* centers, int64_t cap, int32_t* out_center,
               int32_t* out_atom, int64_t* count);

/* ---- _make_selection (I:1384-1451) --------------------------------------- */
/* in_selection: u8[n] = utils.selection_parser result (NULL = keep the mask already uploaded,
 * or the whole structure if none was).
 * Computes selection_plus (6.0 A expansion over ALL atoms incl. hydrogens,
 * I:1420-1424), residue sets and ring/amide id sets (I:1416-1417,1434-1437)
 * and keeps them in the context.  Outputs may be NULL. */
int arp_make_selection(arp_ctx* ctx, const uint8_t* in_selection, double expand_radius,
                       uint8_t* out_plus /*n*/, uint8_t* out_ring_sel /*nring*/,
                       uint8_t* out_ring_plus /*nring*/, uint8_t* out_amide_sel /*namide*/,
                       uint8_t* out_amide_plus /*namide*/);

/* Download the sets computed by the last expansion (selection_plus, I:1444-1451). */
int arp_get_selection(arp_ctx* ctx, uint8_t* out_plus, uint8_t* out_ring_sel, uint8_t* out_ring_plus,
                      uint8_t* out_amide_sel, uint8_t* out_amide_plus);

/* Upload a selection mask without expanding it yet (arp_run_launch expands it). */
int arp_set_selection(arp_ctx* ctx, const uint8_t* in_selection);

/* ---- run_arpeggio (I:329-347) --------------------------------------------- */
/* The whole hot path on the resident structure: _make_selection (6.0 A expansion of the
 * selection uploaded by arp_set_selection / arp_make_selection; whole structure if none),
 * _calculate_atom_contacts, _calculate_ring_contacts, _calculate_group_contacts.  Every
 * kernel is enqueued back to back on the context stream with ONE host synchronisation at
 * the end; results stay in HBM until fetched.  counts = records per bag in the order of
 * get_contacts (I:183-210): atom-atom, plane-plane, atom-plane, group-group, group-plane. */
int arp_run_launch(arp_ctx* ctx, double cutoff, double vdw_comp, int include_sequence_adjacent,
                   double expand_radius, int64_t counts[5]);
/* The same pass in two calls, so that ONE host thread can keep several contexts (structures) busy: arp_run_enqueue returns
 * as soon as the launches are enqueued (~15 us of host time), arp_run_wait blocks for the pass, checks the capacities
 * (re-running the pass itself if a buffer was too small) and reports the counts.  Between the two calls only other contexts
 * may be used; every enqueue needs its wait before the next one on the same context. */
int arp_run_enqueue(arp_ctx* ctx, double cutoff, double vdw_comp, int include_sequence_adjacent, double expand_radius);
int arp_run_wait(arp_ctx* ctx, int64_t counts[5]);

/* ---- _calculate_atom_contacts (I:693-936) -------------------------------- */
/* Enqueue bin + sort + neighbour search + fused per-pair SIFt kernels on the
 * context stream; results stay in HBM.  Synchronises the stream before
 * returning and reports the number of emitted contacts.  If the device output
 * buffer was too small it is regrown and the pass re-run internally. */
int arp_atom_contacts_launch(arp_ctx* ctx, double cutoff, double vdw_comp,
                             int include_sequence_adjacent, int64_t* count);
/* Copy the results of the last launch to the host: in the canonical order — ascending
 * (i, j), the order the boundary defines for the reference's KD-tree delivery order
 * (I:707, exported in that order at I:183-190) — once arp_atom_contacts_sort has run on
 * them, otherwise in the order of the device's pair list (any order). */
int arp_atom_contacts_fetch(arp_ctx* ctx, int64_t cap, int32_t* out_i, int32_t* out_j,
                            float* out_dist, uint16_t* out_sift, uint8_t* out_ctype,
                            int64_t* count);
/* Put the atom-atom records of the last launch into the canonical (i, j) order in HBM
 * (csrc/arp_sort.h: least-significant-digit radix passes over the bits of i, three launches per
 * 9-bit digit; then one launch ranks every record by j inside the run of its i and writes the
 * five columns).  Enqueued on the context stream, no host
 * synchronisation; a second call on the same results is a no-op.  Per-atom accumulators
 * and integer sifts do not depend on it. */
int arp_atom_contacts_sort(arp_ctx* ctx);
/* Every result bag of the last pass with ONE device-to-host copy: the atom-atom bag in
 * canonical order (sorted on the device first if it is not yet) and the used prefixes of the
 * four ring / amide bags behind it.  host = caller's buffer (arp_host_alloc for PCIe speed)
 * of host_bytes; counts = records per bag in get_contacts order (I:183-210: atom-atom,
 * plane-plane, atom-plane, group-group, group-plane); offsets[0..4] = byte offsets of i
 * (int32), j (int32), distance (float32), SIFt (uint16), contact type (uint8) of the
 * atom-atom bag; offsets[5 + 12 * b + q] = array q of ring / amide bag b (b = 0..3 in the
 * order of counts[1..4]; q = 0, 1: the two int32 id columns, 2..5: float64 columns, 6..8:
 * float32 columns, 9..11: uint8 columns — the columns of the bag's *_fetch call in its
 * argument order within each type; 0 = the bag has no such array).  Every ring / amide bag
 * arrives in ITS canonical order as well — plane-plane, group-group and group-plane by (first
 * id, second id), atom-plane by (ring, atom): the order the reference's loops create the
 * records in (I:947-1382) — made on the device (up to ARP_BAG_SORT_MAX records by one block,
 * beyond that by the radix passes of the atom-atom bag).
 * bytes_used = bytes written; ARP_E_CAPACITY with bytes_used set when host_bytes is too small. */
#define ARP_BAG_SORT_MAX 8192
#define ARP_PACKED_OFFSETS 53
int arp_fetch_packed(arp_ctx* ctx, void* host, uint64_t host_bytes, int64_t counts[5],
                     uint64_t offsets[ARP_PACKED_OFFSETS], uint64_t* bytes_used);
/* launch + fetch. */
int arp_atom_contacts(arp_ctx* ctx, double cutoff, double vdw_comp,
                      int include_sequence_adjacent, int64_t cap, int32_t* out_i,
                      int32_t* out_j, float* out_dist, uint16_t* out_sift,
                      uint8_t* out_ctype, int64_t* count);

/* Per-atom side effects of the contact loop, computed from the contact list of the last
 * launch (I:821-852, 923-934; utils.update_atom_sift / update_atom_fsift U:182-221):
 *   out_sift4[4*a + {0,1,2,3}] = atom.sift / sift_inter_only / sift_intra_only / sift_water_only as
 *       15-bit masks (actual_fsift* = the same masks >> 5);
 *   out_counts8[8*a + k] = actual_hbonds, _intra_only, _inter_only, _water_only, actual_polars, ... (same order).
 * utils.update_atom_integer_sift (U:224-242): see arp_atom_integer_sifts. */
int arp_atom_accumulators(arp_ctx* ctx, uint16_t* out_sift4, int32_t* out_counts8);

/* utils.update_atom_integer_sift (U:224-242, called first at I:924-925) from the contact list of the last launch.
 * The reference overwrites atom.integer_sift* at every pair with "binary sift before this pair + this pair", so what
 * is left is decided by the LAST pair of each class that touched the atom, i.e. by the order in which its KD-tree
 * delivered the pairs.  Here the order is the canonical one of this library (contacts sorted by (bgn, end), bgn = lower
 * packed index — the same convention that fixes the bgn/end orientation):
 *   out_isift[60*a + 15*slot + k], slot = {integer_sift, _inter_only, _intra_only, _water_only}, k = SIFt bit,
 *   value = (bit set by an earlier pair of the class) + (bit set by the last pair of the class) in {0, 1, 2}. */
int arp_atom_integer_sifts(arp_ctx* ctx, uint8_t* out_isift);

/* ---- _calculate_ring_contacts (I:938-1206) ------------------------------- */
/* __calculate_atom_plane_contacts (I:947-1062). mask = ARP_AP_* bits. */
int arp_atom_plane(arp_ctx* ctx, int64_t cap, int32_t* out_atom, int32_t* out_ring,
                   double* out_dist, double* out_theta, uint8_t* out_mask,
                   uint8_t* out_ctype, int64_t* count);
/* __calculate_plane_plane_contacts (I:1064-1194), one record per unordered ring
 * pair after the reference's dedupe: type1 = class from the creating visit,
 * type2 = class appended by the reverse visit, ARP_PP_SAME or ARP_PP_SKIPPED. */
int arp_plane_plane(arp_ctx* ctx, int64_t cap, int32_t* out_bgn, int32_t* out_end,
                    double* out_dist, double* out_dihedral, double* out_theta_bgn,
                    double* out_theta_end, uint8_t* out_type1, uint8_t* out_type2,
                    uint8_t* out_ctype, int64_t* count);

/* launch-only / fetch-only halves of the four calls in this section and the next */
int arp_atom_plane_launch(arp_ctx* ctx, int64_t* count);
int arp_plane_plane_launch(arp_ctx* ctx, int64_t* count);
int arp_group_group_launch(arp_ctx* ctx, int64_t* count);
int arp_group_plane_launch(arp_ctx* ctx, int64_t* count);
int arp_atom_plane_fetch(arp_ctx* ctx, int64_t cap, int32_t* out_atom, int32_t* out_ring,
                         double* out_dist, double* out_theta, uint8_t* out_mask,
                         uint8_t* out_ctype, int64_t* count);
int arp_plane_plane_fetch(arp_ctx* ctx, int64_t cap, int32_t* out_bgn, int32_t* out_end,
                          double* out_dist, double* out_dihedral, double* out_theta_bgn,
                          double* out_theta_end, uint8_t* out_type1, uint8_t* out_type2,
                          uint8_t* out_ctype, int64_t* count);
int arp_group_group_fetch(arp_ctx* ctx, int64_t cap, int32_t* out_bgn, int32_t* out_end,
                          float* out_dist, float* out_dihedral, float* out_theta,
                          uint8_t* out_ctype, int64_t* count);
int arp_group_plane_fetch(arp_ctx* ctx, int64_t cap, int32_t* out_amide, int32_t* out_ring,
                          double* out_dist, double* out_dihedral, double* out_theta,
                          uint8_t* out_ctype, int64_t* count);

/* ---- _calculate_group_contacts (I:1208-1382) ----------------------------- */
/* __calculate_group_group_contacts (I:1217-1300): ordered amide pairs, float32. */
int arp_group_group(arp_ctx* ctx, int64_t cap, int32_t* out_bgn, int32_t* out_end,
                    float* out_dist, float* out_dihedral, float* out_theta,
                    uint8_t* out_ctype, int64_t* count);
/* __calculate_group_plane_contacts (I:1302-1382): amide x ring, float64. */
int arp_group_plane(arp_ctx* ctx, int64_t cap, int32_t* out_amide, int32_t* out_ring,
                    double* out_dist, double* out_dihedral, double* out_theta,
                    uint8_t* out_ctype, int64_t* count);

/* ---- multi-GPU slab halo -------------------------------------- */
/* Marks which atoms this rank owns (u8[n]; NULL = all).  A pair is emitted by
 * the rank that owns the atom with the lower global id. global_id: i32[n] id
 * used for the ownership/orientation rule and reported in outputs instead of
 * the local index (NULL = local index). */
int arp_set_ownership(arp_ctx* ctx, const uint8_t* is_home, const int32_t* global_id);

/* Same for rings and amides: which ones this rank owns and their global ids (strictly
 * increasing).  plane-plane pairs are emitted by the owner of the lower ring id, atom-plane
 * records by the owner of the ring, group-group / group-plane by the owner of the (bgn) amide. */
int arp_set_group_ownership(arp_ctx* ctx, const uint8_t* ring_home, const int32_t* ring_gid,
                            const uint8_t* amide_home, const int32_t* amide_gid);
/* Coordinates of utils.get_single_bond_neighbour (U:612-635) given inline (float32[n,3] +
 * presence flag), for shards whose neighbour atom lives on another rank. */
int arp_set_single_bond_neighbour_coords(arp_ctx* ctx, const float* sb_xyz, const uint8_t* sb_present);
/* Install an externally combined _make_selection result (I:1444-1451): used by the slab
 * sharding after the halo exchange of selection_plus bits and the all-reduce of the
 * residue sets. */
int arp_set_selection_state(arp_ctx* ctx, const uint8_t* in_selection, const uint8_t* in_plus,
                            const uint8_t* ring_sel, const uint8_t* ring_plus,
                            const uint8_t* amide_sel, const uint8_t* amide_plus);

/* Staged form of arp_run_launch for slab-sharded runs: between the stages the caller combines the
 * selection state of the ranks ON THE DEVICE (RCCL on tensors that alias the context's buffers).
 *   stage 0  selection_plus of the local atoms (I:1384-1424); exact for the atoms this rank owns.
 *            -> caller overwrites the bits of its halo atoms with their owners' (ARP_BUF_PLUS).
 *   stage 1  residue sets of the selection and of selection_plus (I:1413, 1431).
 *            -> caller all-reduces (MAX) the set over the ranks (ARP_BUF_RES_SETS; the shard uses
 *               global residue ids, so the buffer has the same layout on every rank).
 *   stage 2  ring / amide id sets (I:1416-1437), atom contacts, ring and group contacts; fills counts
 *            like arp_run_launch.
 * Each stage returns after the context's streams have drained. */
#define ARP_BUF_PLUS      0   /* u8[n]        selection_plus                                            */
#define ARP_BUF_RES_SETS  1   /* u8[2 * nres] [0,nres) selection residues, [nres,2nres) selection_plus  */
int arp_device_buffer(arp_ctx* ctx, int which, uint64_t* device_ptr, int64_t* bytes);
int arp_run_stage(arp_ctx* ctx, int stage, double cutoff, double vdw_comp, int include_sequence_adjacent,
                  double expand_radius, int64_t counts[5]);

/* ---- measurement ---------------------------------------------------------- */
/* stats[0]=candidate pairs tested by the last atom-contact search,
 * stats[1]=pairs with d<=cutoff, stats[2]=pairs passing the residue filters
 * (= emitted contacts), stats[3]=atoms binned, stats[4]=grid cells,
 * stats[5]/[6]=candidates / hits of the last selection-expansion search. */
int arp_get_stats(arp_ctx* ctx, int64_t stats[8]);
/* When enabled, every kernel of arp_atom_contacts_launch is bracketed by
 * hipEvents on the context stream.  ms[k], launches[k] accumulate per kernel
 * slot: 0 bin, 1 scan, 2 scatter (+ record build), 3 unused, 4 contact search, 5 sift,
 * 6 selection-expansion search, 7 ring/amide kernels.
 * reset != 0 clears the accumulators after reading. */
int arp_set_profiling(arp_ctx* ctx, int enabled);
int arp_get_kernel_times(arp_ctx* ctx, double ms[8], int64_t launches[8], int reset);
/* ---- geometric part of initialize() (I:288-327): from perceived rings / amide groups to what arp_set_rings /
 * arp_set_amides take.  Perception itself (SSSR, aromaticity, the amide SMARTS) is OpenBabel's and not provided. ----
 *
 * _perceive_rings (I:1697-1733) -> OBRing::findCenterAndNormal: centre = mean of the ring atoms, normal = normalised
 * mean of the cross products of consecutive centre->atom vectors, float64.  ring_off[nring+1] / ring_idx: CSR of the
 * ring atoms in ring order (indices into the atoms of arp_set_atoms).  out_*: double[3 * nring]. */
int arp_ring_geometry(arp_ctx* ctx, int64_t nring, const int32_t* ring_off, const int32_t* ring_idx, double* out_center,
                      double* out_normal);
/* _perceive_amide_groups (I:1531-1589): amide_atoms[4 * namide] = N, C, O, C-alpha of every match; centre = (C + N) / 2
 * (float32, I:1569), normal = unit normal of the C, O, N plane (the reference takes the last right-singular vector
 * of the centred coordinates: same direction, sign not reproduced).  out_*: float[3 * namide]. */
int arp_amide_geometry(arp_ctx* ctx, int64_t namide, const int32_t* amide_atoms, float* out_center, float* out_normal);
/* _assign_aromatic_rings_to_residues (I:1453-1492): residue of the atom nearest to each ring centre among ALL atoms
 * within 3.0 A (float64, inclusive), -1 when there is none; out_shortest (may be NULL) = that distance
 * ('residue_shortest_distance', I:1485), -1 when there is none.  Needs arp_set_atoms (and the residue ids given there). */
int arp_ring_residues(arp_ctx* ctx, int64_t nring, const double* center, int32_t* out_ring_res, double* out_shortest);
/* Page-locked (pinned) host memory.  Result buffers allocated here make the *_fetch calls DMA transfers at PCIe speed
 * (a 1.25 M-contact list: 0.5 ms instead of 2.5 ms into pageable memory); any host pointer is accepted by every call. */
int arp_host_alloc(uint64_t bytes, void** out);
int arp_host_free(void* p);
/* The contact grid of a whole-structure pass (no selection, or every atom selected: interactions.py:1395, 1407) is the
 * structure's own neighbour grid — the counterpart of the KD-tree over `entity` (interactions.py:1394) — and is kept: the
 * next pass with the same structure, cutoff and whole selection builds none.  Passes with any other selection compact
 * selection_plus into a new grid every time, as the reference rebuilds NeighborSearch(selection_plus) (interactions.py:1442).
 * enabled = 0 makes every pass build its grid (measurements); default 1. */
int  arp_set_grid_reuse(arp_ctx* ctx, int enabled);
/* enabled = 1: arp_run_launch / arp_run_wait enqueue the canonical sort of the atom-atom bag (arp_atom_contacts_sort) as soon as
 * the pass has reported its record count, before they return — for callers that fetch the sorted bag next (arp_fetch_packed,
 * arp_atom_contacts_fetch after a sort; I:183-190 exports every record): the sort then starts a host round trip earlier and runs
 * while the caller gets back to the library.  Default 0 (a caller that only wants counts pays for no sort). */
int  arp_set_sort_after_pass(arp_ctx* ctx, int enabled);
/* Layout of the atom-atom bag inside arp_fetch_packed's one piece.  The canonical order is by (bgn, end), so the bgn column of k
 * records is N + 1 row offsets of information — and the reference's consumers walk the bag atom by atom (get_contacts,
 * interactions.py:183-190; the per-atom accumulators, utils.py:182-242).
 *   ARP_LAYOUT_RECORDS (default)  offsets[0] -> int32 bgn[k]
 *   ARP_LAYOUT_ROWS               offsets[0] -> int32 row[N + 1], N = atoms of the resident structure: the records of atom a are
 *                                 [row[a], row[a + 1]) of the other four columns (end, distance, SIFt, type), row[N] = k.
 *                                 4 k bytes less to copy (100 k atoms, 1.25 M records: 18.8 -> 14.2 MB over PCIe).
 * Not available on a context that holds a shard with global ids (arp_fetch_packed then returns ARP_E_ARG).
 * arp_atom_contacts_fetch always hands out records. */
#define ARP_LAYOUT_RECORDS 0
#define ARP_LAYOUT_ROWS    1
int  arp_set_packed_layout(arp_ctx* ctx, int layout);

/* Sharded runs with NO selection (the reference's default, I:1395: every atom of the structure): the caller
 * asserts that the selection is the whole global structure.  Then selection_plus = selection on every rank and every
 * residue of the table is in both residue sets (I:1413-1437) — including residues whose atoms live on another rank,
 * which the local atoms cannot tell — so arp_run_launch on each shard is exact without any exchange between the
 * ranks.  arp_run_launch refuses (ARP_E_ARG) when the flag is on and the uploaded selection is partial.
 * Precondition: every residue of the table has at least one atom somewhere. */
int arp_set_whole_structure(arp_ctx* ctx, int enabled);
/* ---- exchange between the shards of a distributed structure: RCCL behind the C ABI (the reference has no
 * multi-process mode, so nothing is replaced — these carry the one-cell halo the grid sharding needs).  One process per
 * GPU, one arp_ctx each.  Rank 0 asks for a 128-byte id (arp_comm_unique_id) and hands it to the other processes by
 * whatever means the application has (a file, MPI, a TCP store); every rank then calls arp_comm_init with it.  All
 * transfers run on the context's stream (ncclSend / ncclRecv grouped per neighbour, one ncclAllReduce): no host
 * synchronisation between the kernels of a stage and the exchange that follows.  `librccl.so` is loaded on first use. */
int arp_comm_unique_id(uint8_t* out_id, uint64_t capacity /* >= 128 */);
int arp_comm_init(arp_ctx* ctx, int rank, int world, const uint8_t* unique_id /* 128 bytes */);
int arp_comm_destroy(arp_ctx* ctx);
int arp_comm_info(arp_ctx* ctx, int* rank, int* world);          /* rank / size as RCCL reports them (ncclCommUserRank / ncclCommCount);
                                                                   * ARP_E_ARG while there is no communicator */
/* Halo records of a structure: the two face buffers cut out by arp_shard_pack_face (device pointers; bytes = 0 or a missing
 * neighbour: nothing sent) go to ranks rank - 1 / rank + 1, theirs arrive in buffers the context owns:
 * received = {left pointer, left bytes, right pointer, right bytes}, valid until the next call — the arguments of
 * arp_shard_assemble.  Sizes travel first (one 64-bit word each way), then the records. */
int arp_shard_exchange_faces(arp_ctx* ctx, uint64_t left_ptr, uint64_t left_bytes, uint64_t right_ptr, uint64_t right_bytes,
                             uint64_t received[4]);
/* Per-pass selection exchange of a sharded run WITH a selection (arp_run_stage): the local indices of the atoms whose
 * selection_plus bit each neighbour needs (my home atoms inside its halo, ascending global id) and of the halo atoms it
 * owns (same order on its side).  Uploaded once per structure. */
int arp_shard_set_exchange_lists(arp_ctx* ctx, const int32_t* send_left, int64_t n_send_left, const int32_t* send_right, int64_t n_send_right,
                                 const int32_t* recv_left, int64_t n_recv_left, const int32_t* recv_right, int64_t n_recv_right);
/* between stage 0 and stage 1: halo atoms take their selection_plus bit from their owner (gather, grouped send / recv, scatter) */
int arp_shard_exchange_plus(arp_ctx* ctx);
/* between stage 1 and stage 2: the residue sets (I:1413, 1431) OR-ed over all ranks, in place (ncclAllReduce, MAX over uint8) */
int arp_shard_reduce_residue_sets(arp_ctx* ctx);
/* Several structures in ONE pass — the reference's production use is the weekly PDBe release, ~10^5 entries of a few
 * thousand atoms each (README.md:4), and one such structure is four launches of fixed cost on this chip.  The caller
 * uploads the CONCATENATION of nstruct structures through the usual setters (atom / residue / ring / amide indices
 * shifted by each structure's offsets, coordinates unchanged) and then declares the partition: structure s = atoms
 * [atom_off[s], atom_off[s + 1]), rings [ring_off[s], ..), amides [amide_off[s], ..) (nstruct + 1 entries each, first 0,
 * last = the resident counts); boxes[6 * s ..] = lo x, y, z, hi x, y, z of everything structure s holds (atoms, ring and
 * amide centres).  Every grid of a pass then gives each structure its own cells — its box at an integer cell offset, an
 * empty cell between any two structures — so no neighbour search, selection expansion (I:1420) or ring / amide loop
 * ever pairs items of different structures, while every distance is computed from the same coordinates as in a
 * single-structure run: each structure's five bags are bit-identical to its own run.  Selections: one mask over the
 * concatenated atoms (arp_set_selection).  Results carry the concatenated ids; a record belongs to the structure of its
 * first id.  nstruct = 0 returns to one structure.  Not available on a shard (arp_set_ownership). */
int arp_set_batch(arp_ctx* ctx, int64_t nstruct, const int64_t* atom_off, const int64_t* ring_off, const int64_t* amide_off,
                  const double* boxes);
/* Host side of arp_run_launch, accumulated over *passes calls: us[0] = time spent enqueueing the pass
 * (kernel launches, memsets, events), us[1] = time spent blocked in the one synchronisation. */
int arp_get_host_times(arp_ctx* ctx, double us[2], int64_t* passes, int reset);
/* The context's hipStream_t (as an integer), for callers that want to order their own work
 * against it. */
uint64_t arp_stream_handle(arp_ctx* ctx);
/* Enqueue every pass on the caller's stream instead (0 = back to the context's own stream).
 * With a caller stream, stages 0 and 1 of arp_run_stage return WITHOUT synchronising: the caller
 * orders its exchange (e.g. RCCL through torch.distributed) on the same stream, and only stage 2
 * blocks.  The stream must outlive the context or be reset with 0 first. */
int arp_use_stream(arp_ctx* ctx, uint64_t stream);

/* ---- sharded structures assembled on the device (no counterpart in the reference) ---------------
 * One rank = one context.  The rank uploads the records of its HOME atoms / rings / amides once, the device cuts out
 * the records its two neighbours need (the atoms within one halo width of the slab faces), the caller moves those
 * buffers between the GPUs (RCCL isend / irecv on the device pointers), and the device merges home + received halos
 * into the resident structure: sorted by global id, CSR sections rebuilt, bonded global ids turned into local indices
 * (partners that are not local are dropped), residue table filled, ownership (arp_set_ownership /
 * arp_set_group_ownership) and single-bond-neighbour coordinates set.  Nothing of this goes through host memory; the
 * host sees five counters per buffer.
 *
 * A record buffer = arp_rec_header, then the sections at header.off[] (16-byte aligned): atoms (arp_rec_atom, ascending
 * gid), hydrogen coordinates (double[3] each; an atom's run starts at h_start), bonded GLOBAL ids (int32; bond_start),
 * rings (arp_rec_ring, ascending gid), amides (arp_rec_amide, ascending gid).  Residue ids are global. */
#define ARP_REC_MAGIC 0x3143455250524141ull   /* "AARPREC1" */
typedef struct arp_rec_header {
    uint64_t magic, bytes;
    int64_t na, nh, nb, nring, namide, n_rad;
    uint64_t off[5];
    double lo[3], hi[3], ring_lo[3], ring_hi[3], amide_lo[3], amide_hi[3];   /* boxes that contain the points (any such box) */
    double rad_tab[512];            /* n_rad distinct {vdw, cov} pairs of the sender's home atoms */
} arp_rec_header;
typedef struct arp_rec_atom {       /* 96 bytes */
    float x, y, z; int32_t gid;
    double vdw, cov;
    float sb_x, sb_y, sb_z; int32_t sb_has;      /* coordinates of the single-bond heavy neighbour (U:340-359) */
    int32_t res_gid, res_prev, res_next; uint16_t tmask, flags;
    int32_t h_start, h_cnt, bond_start, bond_cnt;
    uint8_t res_flags, sel, pad[14];
} arp_rec_atom;
typedef struct arp_rec_ring { double c[3], n[3]; int32_t gid, res, pad[2]; } arp_rec_ring;      /* 64 bytes */
typedef struct arp_rec_amide { float c[3], n[3]; int32_t gid, res; } arp_rec_amide;            /* 32 bytes */
/* host-only helpers: size of a record buffer, and header (magic, counts, offsets) written at its start */
uint64_t arp_records_size(int64_t na, int64_t nh, int64_t nb, int64_t nring, int64_t namide);
int arp_records_layout(void* buf, uint64_t bytes, int64_t na, int64_t nh, int64_t nb, int64_t nring, int64_t namide);
/* Host-only packer: the records of the given atoms / rings / amides (ascending ids = global ids) of a structure held as
 * the arrays of the classic setters, written into a buffer whose header arp_records_layout has prepared (na, nring,
 * namide = the list lengths; nh, nb = the hydrogens / bonds of the listed atoms).  sel = selection mask over ALL atoms
 * (NULL: everything selected).  Fills the sections, the radius dictionary of the listed atoms and the boxes. */
int arp_records_fill(void* buf, uint64_t bytes, int64_t n_atoms_total, int64_t n_res_total, int64_t n_rings_total, int64_t n_amides_total,
                     const float* xyz, const double* vdw, const double* cov,
                     const uint16_t* type_mask, const uint16_t* flags, const int32_t* res_id, const uint8_t* res_flags,
                     const int32_t* res_prev, const int32_t* res_next, const int32_t* bond_off, const int32_t* bond_idx,
                     const int32_t* h_off, const double* h_xyz, const int32_t* sb_nbr, const double* ring_center,
                     const double* ring_normal, const int32_t* ring_res, const float* amide_center, const float* amide_normal,
                     const int32_t* amide_res, const uint8_t* sel, const int64_t* atom_ids, const int64_t* ring_ids,
                     const int64_t* amide_ids);
/* Upload the home records (one asynchronous copy; the buffer may be reused when the call returns). */
int arp_shard_set_home(arp_ctx* ctx, const void* records, uint64_t bytes);
/* Records of the home atoms with x_lo <= x <= x_hi (float64 comparison; rings and amides by their centre), packed on
 * the device into a buffer the context owns until the next call with the same `slot` (0 or 1).  The order of the home
 * records is kept.  *device_ptr / *out_bytes: what to send. */
int arp_shard_pack_face(arp_ctx* ctx, int slot, double x_lo, double x_hi, uint64_t* device_ptr, uint64_t* out_bytes);
/* Merge the home records with the record buffers received from the left / right neighbour (DEVICE pointers, 0 = no
 * neighbour on that side) and make the result the resident structure of the context (as arp_set_blob would: selection
 * reset to the whole structure).  nres_global = rows of the global residue table.  counts: atoms, rings, amides.
 * ARP_E_ARG when an atom, ring or amide occurs twice. */
int arp_shard_assemble(arp_ctx* ctx, uint64_t dev_left, uint64_t bytes_left, uint64_t dev_right, uint64_t bytes_right,
                       int64_t nres_global, int64_t counts[3]);
/* What the caller needs to map results back and to exchange selection bits: global id, origin (0 home, -1 / +1 received
 * from the left / right) and selection bit of every local atom; global id and origin of every ring and amide.  Any
 * pointer may be NULL. */
int arp_shard_layout(arp_ctx* ctx, int32_t* global_id, int8_t* origin, uint8_t* sel, int32_t* ring_gid, int8_t* ring_origin,
                     int32_t* amide_gid, int8_t* amide_origin);
/* Read back the resident structure in blob form (arp_blob_header + arrays) when it came from arp_set_blob or
 * arp_shard_assemble: *bytes = size; copies when cap suffices (host may be NULL to ask for the size). */
int arp_get_blob(arp_ctx* ctx, void* host, uint64_t cap, uint64_t* bytes);

/* ---- mmCIF category reader (host only; no context, no GPU) — SURVEY.md 8 row f3 ---------------------------------
 * What the reference asks gemmi for (protein_reader.py:258-289, 415-441): one category of the file as columns
 * (cif_block.get_mmcif_category('_atom_site.') / ('_chem_comp.')).  arp_cif_open parses `text` (CIF 1.1 syntax: loops,
 * pair items, quoted strings, text fields, comments) and keeps the items of the FIRST data block whose tags start with
 * `category` (e.g. "_atom_site.", case-insensitive).  A cell is a slice of the text plus a kind: 0 value, 1 bare '?'
 * (gemmi: None), 2 bare '.' (gemmi: False).  Errors: negative return, message in err (if given). */
typedef struct arp_cif arp_cif;
int arp_cif_open(const char* text, uint64_t len, const char* category, arp_cif** out, char* err, uint64_t err_cap);
void arp_cif_close(arp_cif* t);
int64_t arp_cif_rows(const arp_cif* t);
int arp_cif_cols(const arp_cif* t);
int arp_cif_blocks(const arp_cif* t);                 /* data_ blocks in the file (gemmi's sole_block() wants exactly one) */
const char* arp_cif_tag(const arp_cif* t, int col);   /* item name after the category prefix */
const char* arp_cif_text(const arp_cif* t);           /* the text the cells point into */
/* cells of one column: begin / len / kind arrays of arp_cif_rows() entries, filled by the call */
int arp_cif_column(const arp_cif* t, int col, uint64_t* begin, uint32_t* len, uint8_t* kind);
/* float(value) / int(value) of every cell (C strtod / strtoll over the whole cell); '?' and '.' give `missing`;
 * a cell that is not a number: returns -1 and its row in *bad_row */
int arp_cif_column_f64(const arp_cif* t, int col, double missing, double* out, int64_t* bad_row);
int arp_cif_column_i64(const arp_cif* t, int col, int64_t missing, int64_t* out, int64_t* bad_row);

/* ---- JSON output (host only; no context, no GPU) ------------------------------------------
 * The reference's CLI ends with json.dump(get_contacts(), fh, indent=4, sort_keys=True) (scripts/process_protein_cli.py:
 * 184-188; records of I:172-212).  This writes the same bytes for the atom-atom records straight from the result arrays
 * (sorted keys, indentation, float repr and string escaping of Python's json module) — a Python dict per record costs
 * seconds on a whole-structure run — followed by `tail_records`: the n_tail records of the four ring / amide bags, already
 * rendered at the same indentation and joined with ",\n" (they are few; the Python layer renders them).
 * dist = float64(distance) of every contact; flags bit 0: round it here as I:190 does (round(x, 2) = rint(x * 100) / 100),
 * flags = 0: the caller has rounded already.  sift_names: 15 strings, ctype_names: 7 strings.
 * The records are rendered and written by several host threads (ARP_EXPORT_THREADS; default: the CPUs / CPU quota of the
 * process, at most 32): 814 MB for a 100 k-atom structure in 0.12 s of library time on a 16-CPU quota, 0.41 s with one.
 * Returns 0, or a negative number (-1 bad argument, -2 cannot open, -3 index out of range, -4 write error). */
int arp_write_contacts_json(const char* path, int indent, int flags, int64_t n, const int32_t* ci, const int32_t* cj,
                            const double* dist, const uint16_t* sift, const uint8_t* ctype, int64_t n_atoms,
                            const int32_t* atom_res, const char* const* atom_name, int64_t n_res,
                            const char* const* res_name, const int32_t* res_seq, const char* const* res_chain,
                            const char* const* res_icode, const char* const* res_comp_type,
                            const char* const* sift_names, const char* const* ctype_names, const char* tail_records,
                            int64_t n_tail);

#ifdef __cplusplus
}
#endif
#endif /* ARPEGGIO_HIP_H */
