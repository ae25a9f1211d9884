"""ctypes binding of include/arpeggio_hip.h (libarpeggio_hip.so).

There is no CPU fallback: if the library is missing or no GPU is usable the calls
raise NativeLibraryError.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .core.exceptions import NativeLibraryError

_HERE = os.path.dirname(os.path.abspath(__file__))
# ARP_LIB_PATH: another build of the same library (A/B comparisons of kernel variants on one GPU box)
LIB_PATH = os.environ.get('ARP_LIB_PATH') or os.path.join(_HERE, 'csrc', 'libarpeggio_hip.so')

ARP_OK, ARP_E_ARG, ARP_E_HIP, ARP_E_CAPACITY, ARP_E_XBOND_NBR, ARP_E_NOMEM = 0, -1, -2, -3, -4, -5

# every symbol declared in include/arpeggio_hip.h
SYMBOLS = (
    'arp_version', 'arp_create', 'arp_destroy', 'arp_last_error', 'arp_set_atoms', 'arp_set_residues',
    'arp_set_bonds', 'arp_set_hydrogens', 'arp_set_single_bond_neighbours', 'arp_set_rings', 'arp_set_amides',
    'arp_search_all', 'arp_search', 'arp_make_selection', 'arp_atom_contacts_launch', 'arp_atom_contacts_fetch',
    'arp_atom_contacts', 'arp_atom_plane', 'arp_plane_plane', 'arp_group_group', 'arp_group_plane',
    'arp_set_ownership', 'arp_get_stats', 'arp_set_profiling', 'arp_get_kernel_times', 'arp_stream_handle',
    'arp_set_selection', 'arp_run_launch', 'arp_run_enqueue', 'arp_run_wait', 'arp_atom_plane_launch', 'arp_plane_plane_launch', 'arp_group_group_launch',
    'arp_group_plane_launch', 'arp_atom_plane_fetch', 'arp_plane_plane_fetch', 'arp_group_group_fetch',
    'arp_group_plane_fetch', 'arp_get_selection', 'arp_set_group_ownership', 'arp_set_single_bond_neighbour_coords',
    'arp_set_selection_state', 'arp_atom_accumulators', 'arp_device_buffer', 'arp_run_stage', 'arp_use_stream',
    'arp_get_host_times', 'arp_set_whole_structure', 'arp_set_grid_reuse', 'arp_set_sort_after_pass', 'arp_set_packed_layout', 'arp_set_batch', 'arp_device_count', 'arp_device_synchronize', 'arp_comm_unique_id', 'arp_comm_init', 'arp_comm_destroy', 'arp_comm_info',
    'arp_shard_exchange_faces', 'arp_shard_set_exchange_lists', 'arp_shard_exchange_plus', 'arp_shard_reduce_residue_sets', 'arp_ring_geometry', 'arp_amide_geometry', 'arp_ring_residues',
    'arp_host_alloc', 'arp_host_free', 'arp_atom_integer_sifts', 'arp_blob_size', 'arp_blob_layout', 'arp_set_blob', 'arp_blob_fill',
    'arp_write_contacts_json', 'arp_records_size', 'arp_records_layout', 'arp_records_fill', 'arp_shard_set_home', 'arp_shard_pack_face',
    'arp_shard_assemble', 'arp_shard_layout', 'arp_get_blob', 'arp_cif_open', 'arp_cif_close', 'arp_cif_rows', 'arp_cif_cols',
    'arp_cif_blocks', 'arp_cif_tag', 'arp_cif_text', 'arp_cif_column', 'arp_cif_column_f64', 'arp_cif_column_i64',
    'arp_atom_contacts_sort', 'arp_fetch_packed',
)

_lib = None


def device_count():
    """GPUs this process can see (the library's own answer: no torch involved)."""
    return int(load().arp_device_count())


def _declare_host_entry_points(L):
    """argtypes of the host-only entry points (mmCIF reader, JSON writer): the same in both libraries."""
    vp, i64, dbl, i32 = C.c_void_p, C.c_int64, C.c_double, C.c_int
    L.arp_cif_open.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.POINTER(vp), C.c_char_p, C.c_uint64]
    L.arp_cif_close.argtypes = [vp]
    L.arp_cif_close.restype = None
    L.arp_cif_rows.argtypes = [vp]
    L.arp_cif_rows.restype = C.c_int64
    L.arp_cif_cols.argtypes = [vp]
    L.arp_cif_blocks.argtypes = [vp]
    L.arp_cif_tag.argtypes = [vp, i32]
    L.arp_cif_tag.restype = C.c_char_p
    L.arp_cif_text.argtypes = [vp]
    L.arp_cif_text.restype = vp
    L.arp_cif_column.argtypes = [vp, i32, vp, vp, vp]
    L.arp_cif_column_f64.argtypes = [vp, i32, dbl, vp, C.POINTER(i64)]
    L.arp_cif_column_i64.argtypes = [vp, i32, i64, vp, C.POINTER(i64)]
    L.arp_write_contacts_json.argtypes = [C.c_char_p, i32, i32, i64, vp, vp, vp, vp, vp, i64, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, C.c_char_p, i64]


HOST_LIB_PATH = os.path.join(_HERE, 'csrc', 'libarpeggio_host.so')
HOST_SYMBOLS = ('arp_cif_open', 'arp_cif_close', 'arp_cif_rows', 'arp_cif_cols', 'arp_cif_blocks', 'arp_cif_tag', 'arp_cif_text', 'arp_cif_column',
                'arp_cif_column_f64', 'arp_cif_column_i64', 'arp_write_contacts_json')
_host_lib = None


def load_host():
    """The library for the HOST-ONLY entry points (the mmCIF category reader, the JSON writer of the records): the HIP library
    when it has been built — it holds them too —, otherwise ``libarpeggio_host.so``, the same sources (csrc/arp_cif.h,
    arp_cif_api.h, arp_json.h) compiled with g++ (``arpeggio_amd.build.build_host``; built on first use).  No compute entry
    point lives there: a machine without hipcc can read files and regenerate the golden fixtures, nothing more."""
    global _host_lib
    if _lib is not None:
        return _lib
    if _host_lib is not None:
        return _host_lib
    if os.path.exists(LIB_PATH):
        return load()
    from . import build as _build
    path = _build.build_host()
    try:
        L = C.CDLL(path)
    except OSError as e:
        raise NativeLibraryError(f'cannot load {path}: {e}') from e
    _declare_host_entry_points(L)
    _host_lib = L
    return L


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(f'{LIB_PATH} not found: build it with `python -m arpeggio_amd.build` '
                                 '(hipcc --offload-arch=gfx950); there is no CPU fallback')
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:
        raise NativeLibraryError(f'cannot load {LIB_PATH}: {e}') from e
    vp, i64, dbl, i32 = C.c_void_p, C.c_int64, C.c_double, C.c_int
    L.arp_version.restype = C.c_char_p
    L.arp_last_error.restype = C.c_char_p
    L.arp_last_error.argtypes = [vp]
    L.arp_create.argtypes = [i32, C.POINTER(vp)]
    L.arp_destroy.argtypes = [vp]
    L.arp_destroy.restype = None
    L.arp_set_atoms.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp]
    L.arp_set_residues.argtypes = [vp, i64, vp, vp, vp]
    L.arp_set_bonds.argtypes = [vp, vp, vp]
    L.arp_set_hydrogens.argtypes = [vp, vp, vp]
    L.arp_set_single_bond_neighbours.argtypes = [vp, vp]
    L.arp_set_rings.argtypes = [vp, i64, vp, vp, vp]
    L.arp_set_amides.argtypes = [vp, i64, vp, vp, vp]
    L.arp_search_all.argtypes = [vp, dbl, vp, i64, vp, vp, C.POINTER(i64)]
    L.arp_search.argtypes = [vp, dbl, i64, vp, i64, vp, vp, C.POINTER(i64)]
    L.arp_make_selection.argtypes = [vp, vp, dbl, vp, vp, vp, vp, vp]
    L.arp_atom_contacts_launch.argtypes = [vp, dbl, dbl, i32, C.POINTER(i64)]
    L.arp_atom_contacts_fetch.argtypes = [vp, i64, vp, vp, vp, vp, vp, C.POINTER(i64)]
    L.arp_atom_contacts.argtypes = [vp, dbl, dbl, i32, i64, vp, vp, vp, vp, vp, C.POINTER(i64)]
    L.arp_atom_contacts_sort.argtypes = [vp]
    L.arp_fetch_packed.argtypes = [vp, vp, C.c_uint64, vp, vp, C.POINTER(C.c_uint64)]
    L.arp_atom_plane.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, C.POINTER(i64)]
    L.arp_plane_plane.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(i64)]
    L.arp_group_group.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, C.POINTER(i64)]
    L.arp_group_plane.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, C.POINTER(i64)]
    L.arp_set_ownership.argtypes = [vp, vp, vp]
    L.arp_set_selection.argtypes = [vp, vp]
    L.arp_atom_accumulators.argtypes = [vp, vp, vp]
    L.arp_atom_integer_sifts.argtypes = [vp, vp]
    L.arp_blob_size.argtypes = [i64] * 6
    L.arp_blob_size.restype = C.c_uint64
    L.arp_blob_layout.argtypes = [vp, C.c_uint64] + [i64] * 6
    L.arp_set_blob.argtypes = [vp, vp, C.c_uint64]
    L.arp_blob_fill.argtypes = [vp, C.c_uint64] + [vp] * 20
    L.arp_records_size.argtypes = [i64] * 5
    L.arp_records_size.restype = C.c_uint64
    L.arp_records_layout.argtypes = [vp, C.c_uint64] + [i64] * 5
    L.arp_records_fill.argtypes = [vp, C.c_uint64, i64, i64, i64, i64] + [vp] * 24
    L.arp_shard_set_home.argtypes = [vp, vp, C.c_uint64]
    L.arp_shard_pack_face.argtypes = [vp, i32, dbl, dbl, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.arp_shard_assemble.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, i64, vp]
    L.arp_shard_layout.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.arp_get_blob.argtypes = [vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    _declare_host_entry_points(L)
    L.arp_device_buffer.argtypes = [vp, i32, C.POINTER(C.c_uint64), C.POINTER(i64)]
    L.arp_run_stage.argtypes = [vp, i32, dbl, dbl, i32, dbl, vp]
    L.arp_set_group_ownership.argtypes = [vp, vp, vp, vp, vp]
    L.arp_set_single_bond_neighbour_coords.argtypes = [vp, vp, vp]
    L.arp_set_selection_state.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.arp_get_selection.argtypes = [vp, vp, vp, vp, vp, vp]
    L.arp_run_launch.argtypes = [vp, dbl, dbl, i32, dbl, vp]
    L.arp_run_enqueue.argtypes = [vp, dbl, dbl, i32, dbl]
    L.arp_run_wait.argtypes = [vp, vp]
    for nm in ('atom_plane', 'plane_plane', 'group_group', 'group_plane'):
        getattr(L, f'arp_{nm}_launch').argtypes = [vp, C.POINTER(i64)]
        getattr(L, f'arp_{nm}_fetch').argtypes = getattr(L, f'arp_{nm}').argtypes
    L.arp_get_stats.argtypes = [vp, vp]
    L.arp_set_profiling.argtypes = [vp, i32]
    L.arp_get_kernel_times.argtypes = [vp, vp, vp, i32]
    L.arp_get_host_times.argtypes = [vp, vp, vp, i32]
    L.arp_set_whole_structure.argtypes = [vp, i32]
    L.arp_set_grid_reuse.argtypes = [vp, i32]
    L.arp_set_sort_after_pass.argtypes = [vp, i32]
    L.arp_set_packed_layout.argtypes = [vp, i32]
    L.arp_device_synchronize.argtypes = [vp]
    L.arp_set_batch.argtypes = [vp, i64, vp, vp, vp, vp]
    L.arp_comm_unique_id.argtypes = [vp, C.c_uint64]
    L.arp_comm_init.argtypes = [vp, i32, i32, vp]
    L.arp_comm_destroy.argtypes = [vp]
    L.arp_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.arp_shard_exchange_faces.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, vp]
    L.arp_shard_set_exchange_lists.argtypes = [vp, vp, i64, vp, i64, vp, i64, vp, i64]
    L.arp_shard_exchange_plus.argtypes = [vp]
    L.arp_shard_reduce_residue_sets.argtypes = [vp]
    L.arp_ring_geometry.argtypes = [vp, i64, vp, vp, vp, vp]
    L.arp_amide_geometry.argtypes = [vp, i64, vp, vp, vp]
    L.arp_ring_residues.argtypes = [vp, i64, vp, vp, vp]
    L.arp_host_alloc.argtypes = [C.c_uint64, C.POINTER(C.c_void_p)]
    L.arp_host_free.argtypes = [vp]
    L.arp_stream_handle.argtypes = [vp]
    L.arp_use_stream.argtypes = [vp, C.c_uint64]
    L.arp_stream_handle.restype = C.c_uint64
    for s in SYMBOLS:
        f = getattr(L, s)
        if s not in ('arp_version', 'arp_last_error', 'arp_destroy', 'arp_stream_handle', 'arp_blob_size', 'arp_records_size',
                     'arp_cif_close', 'arp_cif_rows', 'arp_cif_tag', 'arp_cif_text'):
            f.restype = C.c_int
    _lib = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def pinned_empty(n, dtype):
    """NumPy array over page-locked memory from arp_host_alloc (freed when the array and its views are gone): the
    fetch calls fill such arrays at PCIe speed.  Falls back to ordinary memory when the allocation fails."""
    import weakref
    dt = np.dtype(dtype)
    nbytes = int(n) * dt.itemsize
    if nbytes == 0:
        return np.empty(int(n), dt)
    L = load()
    ptr = C.c_void_p()
    if L.arp_host_alloc(nbytes, C.byref(ptr)) != 0 or not ptr.value:
        return np.empty(int(n), dt)
    buf = (C.c_char * nbytes).from_address(ptr.value)
    weakref.finalize(buf, L.arp_host_free, C.c_void_p(ptr.value))
    return np.frombuffer(buf, dtype=dt, count=int(n))


class BlobHeader(C.Structure):
    """arp_blob_header of include/arpeggio_hip.h."""
    _fields_ = [('magic', C.c_uint64), ('bytes', C.c_uint64), ('n', C.c_int64), ('nres', C.c_int64), ('nbond', C.c_int64),
                ('nh', C.c_int64), ('nring', C.c_int64), ('namide', C.c_int64), ('n_rad', C.c_int64), ('off', C.c_uint64 * 21),
                ('lo', C.c_double * 3), ('hi', C.c_double * 3), ('ring_lo', C.c_double * 3), ('ring_hi', C.c_double * 3),
                ('amide_lo', C.c_double * 3), ('amide_hi', C.c_double * 3)]


_BLOB_DTYPES = (np.float32, np.float64, np.uint16, np.uint16, np.int32, np.uint8, np.int32, np.int32, np.int32, np.int32, np.int32,
                np.float64, np.int32, np.float64, np.float64, np.int32, np.float32, np.float32, np.int32, np.uint16, np.float64)


def pack_blob(pc, pinned=True):
    """The PackedComplex as ONE device-ready buffer (arp_blob_header + arrays, include/arpeggio_hip.h): a uint8 NumPy
    array over page-locked memory (``pinned``), ready for ``Context.set_blob``.  This is the packing step a producer of
    structures does once per structure; nothing is converted again on the way to the GPU."""
    L = load()
    n, nres, nring, namide = pc.n_atoms, pc.n_residues, pc.n_rings, pc.n_amides
    nbond, nh = int(pc.bond_idx.shape[0]), int(pc.h_xyz.shape[0])
    size = int(L.arp_blob_size(n, nres, nbond, nh, nring, namide))
    if size == 0:
        raise ValueError('pack_blob: counts out of range')
    buf = pinned_empty(size, np.uint8) if pinned else np.empty(size, np.uint8)
    if L.arp_blob_layout(_p(buf), size, n, nres, nbond, nh, nring, namide) != ARP_OK:
        raise ValueError('arp_blob_layout failed')
    c = lambda a, dt: np.ascontiguousarray(a, dt)
    arrays = [c(pc.xyz, np.float32), c(pc.vdw, np.float64), c(pc.cov, np.float64), c(pc.type_mask, np.uint16), c(pc.flags, np.uint16),
              c(pc.res_id, np.int32), c(pc.res_flags, np.uint8), c(pc.res_prev, np.int32), c(pc.res_next, np.int32),
              c(pc.bond_off, np.int32), c(pc.bond_idx, np.int32), c(pc.h_off, np.int32), c(pc.h_xyz, np.float64), c(pc.sb_nbr, np.int32),
              c(pc.ring_center, np.float64), c(pc.ring_normal, np.float64), c(pc.ring_res, np.int32), c(pc.amide_center, np.float32),
              c(pc.amide_normal, np.float32), c(pc.amide_res, np.int32)]
    if L.arp_blob_fill(_p(buf), size, *[_p(a) for a in arrays]) != ARP_OK:
        raise ValueError('arp_blob_fill failed')
    return buf


def unpack_blob(buf):
    """The arrays of a blob (as written by ``pack_blob`` or read back with ``Context.get_blob``) as a dict of NumPy views,
    radii decoded through the dictionary; for tests and tools."""
    hdr = BlobHeader.from_buffer_copy(buf[:C.sizeof(BlobHeader)].tobytes())
    n, nres, nring, namide, nbond, nh = int(hdr.n), int(hdr.nres), int(hdr.nring), int(hdr.namide), int(hdr.nbond), int(hdr.nh)
    counts = (4 * n, 2 * n, n, n, n, nres, nres, nres, n + 1, nbond, n + 1, 3 * nh, n, 3 * nring, 3 * nring, nring, 3 * namide,
              3 * namide, namide, n, 512)
    v = [np.frombuffer(buf, dtype=dt, count=cnt, offset=int(hdr.off[k])) for k, (dt, cnt) in enumerate(zip(_BLOB_DTYPES, counts))]
    names = ('xyz4', 'rad', 'type_mask', 'flags', 'res_id', 'res_flags', 'res_prev', 'res_next', 'bond_off', 'bond_idx', 'h_off',
             'h_xyz', 'sb_nbr', 'ring_center', 'ring_normal', 'ring_res', 'amide_center', 'amide_normal', 'amide_res', 'rad_idx',
             'rad_tab')
    d = dict(zip(names, v))
    d['xyz'] = d['xyz4'].reshape(-1, 4)[:, :3]
    d['rad'] = d['rad'].reshape(-1, 2)
    d['rad_tab'] = d['rad_tab'].reshape(256, 2)
    for k in ('h_xyz', 'ring_center', 'ring_normal', 'amide_center', 'amide_normal'):
        d[k] = d[k].reshape(-1, 3)
    d['header'] = hdr
    return d


# ---- record buffers of the device-side shard assembly (arp_rec_header / arp_rec_atom / ... of the header file) ----------
class RecHeader(C.Structure):
    _fields_ = [('magic', C.c_uint64), ('bytes', C.c_uint64), ('na', C.c_int64), ('nh', C.c_int64), ('nb', C.c_int64),
                ('nring', C.c_int64), ('namide', C.c_int64), ('n_rad', C.c_int64), ('off', C.c_uint64 * 5),
                ('lo', C.c_double * 3), ('hi', C.c_double * 3), ('ring_lo', C.c_double * 3), ('ring_hi', C.c_double * 3),
                ('amide_lo', C.c_double * 3), ('amide_hi', C.c_double * 3), ('rad_tab', C.c_double * 512)]


REC_ATOM = np.dtype([('xyz', '<f4', 3), ('gid', '<i4'), ('vdw', '<f8'), ('cov', '<f8'), ('sb_xyz', '<f4', 3), ('sb_has', '<i4'),
                     ('res_gid', '<i4'), ('res_prev', '<i4'), ('res_next', '<i4'), ('tmask', '<u2'), ('flags', '<u2'),
                     ('h_start', '<i4'), ('h_cnt', '<i4'), ('bond_start', '<i4'), ('bond_cnt', '<i4'),
                     ('res_flags', 'u1'), ('sel', 'u1'), ('pad', 'u1', 14)])
REC_RING = np.dtype([('c', '<f8', 3), ('n', '<f8', 3), ('gid', '<i4'), ('res', '<i4'), ('pad', '<i4', 2)])
REC_AMIDE = np.dtype([('c', '<f4', 3), ('n', '<f4', 3), ('gid', '<i4'), ('res', '<i4')])
assert REC_ATOM.itemsize == 96 and REC_RING.itemsize == 64 and REC_AMIDE.itemsize == 32


def _rec_views(buf, hdr):
    return (np.frombuffer(buf, REC_ATOM, int(hdr.na), int(hdr.off[0])), np.frombuffer(buf, np.float64, 3 * int(hdr.nh), int(hdr.off[1])).reshape(-1, 3),
            np.frombuffer(buf, np.int32, int(hdr.nb), int(hdr.off[2])), np.frombuffer(buf, REC_RING, int(hdr.nring), int(hdr.off[3])),
            np.frombuffer(buf, REC_AMIDE, int(hdr.namide), int(hdr.off[4])))


def pack_records_buffer(rec, pinned=True):
    """``sharding.pack_records`` output (dict of arrays, ascending ids) as ONE record buffer for ``Context.shard_set_home``."""
    L = load()
    na, nh, nb = int(rec['gid'].size), int(rec['h_xyz'].shape[0]), int(rec['bond_gid'].size)
    nr, nm = int(rec['ring_gid'].size), int(rec['amide_gid'].size)
    size = int(L.arp_records_size(na, nh, nb, nr, nm))
    if size == 0:
        raise ValueError('pack_records_buffer: counts out of range')
    buf = pinned_empty(size, np.uint8) if pinned else np.empty(size, np.uint8)
    buf[:] = 0
    if L.arp_records_layout(_p(buf), size, na, nh, nb, nr, nm) != ARP_OK:
        raise ValueError('arp_records_layout failed')
    hdr = RecHeader.from_buffer(buf)
    a, h, b, r, m = _rec_views(buf, hdr)
    a['xyz'], a['gid'], a['vdw'], a['cov'] = rec['xyz'], rec['gid'], rec['vdw'], rec['cov']
    a['sb_xyz'], a['sb_has'] = rec['sb_xyz'], rec['sb_has']
    a['res_gid'], a['res_prev'], a['res_next'], a['res_flags'] = rec['res_gid'], rec['res_prev'], rec['res_next'], rec['res_flags']
    a['tmask'], a['flags'], a['sel'] = rec['tmask'], rec['flags'], rec['sel']
    a['h_cnt'], a['bond_cnt'] = rec['h_cnt'], rec['bond_cnt']
    a['h_start'] = np.concatenate([[0], np.cumsum(rec['h_cnt'])])[:-1] if na else 0
    a['bond_start'] = np.concatenate([[0], np.cumsum(rec['bond_cnt'])])[:-1] if na else 0
    h[:], b[:] = rec['h_xyz'], rec['bond_gid']
    r['c'], r['n'], r['gid'], r['res'] = rec['ring_center'], rec['ring_normal'], rec['ring_gid'], rec['ring_res']
    m['c'], m['n'], m['gid'], m['res'] = rec['amide_center'], rec['amide_normal'], rec['amide_gid'], rec['amide_res']
    tab = np.zeros((256, 2))
    if na:
        keys = np.stack([rec['vdw'], rec['cov']], axis=1).astype(np.float64).view(np.uint64)
        uniq = np.unique(keys, axis=0)[:256]
        tab[:len(uniq)] = uniq.view(np.float64)
        hdr.n_rad = len(uniq)
    for k, val in enumerate(tab.reshape(-1)):
        hdr.rad_tab[k] = val
    for name, arr in (('', rec['xyz']), ('ring_', rec['ring_center']), ('amide_', rec['amide_center'])):
        lo = arr.min(axis=0).astype(np.float64) if len(arr) else np.zeros(3)
        hi = arr.max(axis=0).astype(np.float64) if len(arr) else np.zeros(3)
        for k in range(3):
            getattr(hdr, name + 'lo')[k] = lo[k]
            getattr(hdr, name + 'hi')[k] = hi[k]
    del hdr
    return buf


def pack_records_native(pc, atom_ids, ring_ids, amide_ids, sel=None, pinned=True):
    """The record buffer of the given atoms / rings / amides of ``pc`` (ascending packed indices = global ids), packed by the
    library (``arp_records_fill``): the same bytes as ``pack_records_buffer(sharding.pack_records(...))``, ~50x sooner."""
    L = load()
    a = np.ascontiguousarray(atom_ids, np.int64)
    r = np.ascontiguousarray(ring_ids, np.int64)
    m = np.ascontiguousarray(amide_ids, np.int64)
    h_off, b_off = np.ascontiguousarray(pc.h_off, np.int32), np.ascontiguousarray(pc.bond_off, np.int32)
    nh = int((h_off[a + 1] - h_off[a]).sum()) if a.size else 0
    nb = int((b_off[a + 1] - b_off[a]).sum()) if a.size else 0
    size = int(L.arp_records_size(a.size, nh, nb, r.size, m.size))
    if size == 0:
        raise ValueError('pack_records_native: counts out of range')
    buf = pinned_empty(size, np.uint8) if pinned else np.empty(size, np.uint8)
    if L.arp_records_layout(_p(buf), size, a.size, nh, nb, r.size, m.size) != ARP_OK:
        raise ValueError('arp_records_layout failed')
    c = lambda x, dt: np.ascontiguousarray(x, dt)
    arrays = [c(pc.xyz, np.float32), c(pc.vdw, np.float64), c(pc.cov, np.float64), c(pc.type_mask, np.uint16), c(pc.flags, np.uint16),
              c(pc.res_id, np.int32), c(pc.res_flags, np.uint8), c(pc.res_prev, np.int32), c(pc.res_next, np.int32), b_off,
              c(pc.bond_idx, np.int32), h_off, c(pc.h_xyz, np.float64), c(pc.sb_nbr, np.int32), c(pc.ring_center, np.float64),
              c(pc.ring_normal, np.float64), c(pc.ring_res, np.int32), c(pc.amide_center, np.float32), c(pc.amide_normal, np.float32),
              c(pc.amide_res, np.int32), (None if sel is None else c(sel, np.uint8)), a, r, m]
    if L.arp_records_fill(_p(buf), size, pc.n_atoms, pc.n_residues, pc.n_rings, pc.n_amides, *[_p(x) for x in arrays]) != ARP_OK:
        raise ValueError('arp_records_fill failed (ids must ascend and lie inside the structure)')
    return buf


def unpack_records_buffer(buf):
    """Inverse of ``pack_records_buffer`` (the dict ``sharding.pack_records`` makes); for tests."""
    hdr = RecHeader.from_buffer_copy(buf[:C.sizeof(RecHeader)].tobytes())
    a, h, b, r, m = _rec_views(buf, hdr)
    return {'gid': a['gid'].copy(), 'xyz': a['xyz'].copy(), 'vdw': a['vdw'].copy(), 'cov': a['cov'].copy(), 'tmask': a['tmask'].copy(),
            'flags': a['flags'].copy(), 'res_gid': a['res_gid'].copy(), 'res_flags': a['res_flags'].copy(), 'res_prev': a['res_prev'].copy(),
            'res_next': a['res_next'].copy(), 'sel': a['sel'].copy(), 'sb_xyz': a['sb_xyz'].copy(), 'sb_has': a['sb_has'].astype(np.uint8),
            'h_cnt': a['h_cnt'].copy(), 'h_xyz': h.copy(), 'bond_cnt': a['bond_cnt'].copy(), 'bond_gid': b.copy(),
            'h_start': a['h_start'].copy(), 'bond_start': a['bond_start'].copy(),
            'ring_gid': r['gid'].copy(), 'ring_center': r['c'].copy(), 'ring_normal': r['n'].copy(), 'ring_res': r['res'].copy(),
            'amide_gid': m['gid'].copy(), 'amide_center': m['c'].copy(), 'amide_normal': m['n'].copy(), 'amide_res': m['res'].copy(),
            'header': hdr}


class RowsBag(dict):
    """The atom-atom bag in ROWS layout (``Context.set_packed_layout(rows=True)``): ``row`` (N + 1 offsets) instead of ``i``;
    ``bag['i']`` is made from it on first use."""

    def __missing__(self, key):
        if key != 'i':
            raise KeyError(key)
        row = self['row']
        v = np.repeat(np.arange(len(row) - 1, dtype=np.int32), np.diff(row))
        self['i'] = v
        return v


class CifCategory:
    """One category of an mmCIF text (``arp_cif_open``): what gemmi's ``cif_block.get_mmcif_category(name)`` gives the
    reference — ``columns()[item]`` is a list with ``None`` for '?', ``False`` for '.', the unquoted string otherwise —
    plus bulk numeric access for the big ``_atom_site`` table."""

    def __init__(self, text, category):
        self._L = load_host()
        raw = text.encode('utf-8') if isinstance(text, str) else bytes(text)
        h, err = C.c_void_p(), C.create_string_buffer(256)
        rc = self._L.arp_cif_open(raw, len(raw), category.encode(), C.byref(h), err, 256)
        if rc != ARP_OK:
            raise ValueError(err.value.decode() or f'arp_cif_open failed ({rc})')
        self._h = h
        self.category = category
        self.rows, self.n_blocks = int(self._L.arp_cif_rows(h)), int(self._L.arp_cif_blocks(h))
        self.tags = [self._L.arp_cif_tag(h, k).decode() for k in range(int(self._L.arp_cif_cols(h)))]
        self._raw = raw

    def close(self):
        if getattr(self, '_h', None):
            self._L.arp_cif_close(self._h)
            self._h = None

    __del__ = close

    def __contains__(self, item):
        return item in self.tags

    def _cells(self, item):
        col = self.tags.index(item)
        b, ln, kd = np.empty(self.rows, np.uint64), np.empty(self.rows, np.uint32), np.empty(self.rows, np.uint8)
        if self._L.arp_cif_column(self._h, col, _p(b), _p(ln), _p(kd)) != ARP_OK:
            raise ValueError(f'arp_cif_column({item})')
        return b, ln, kd

    def column(self, item):
        """The column as gemmi delivers it: str / None ('?') / False ('.')."""
        b, ln, kd = self._cells(item)
        raw = self._raw
        return [raw[s:s + n].decode('utf-8') if k == 0 else (None if k == 1 else False)
                for s, n, k in zip(b.tolist(), ln.tolist(), kd.tolist())]

    def columns(self):
        return {t: self.column(t) for t in self.tags}

    def floats(self, item, missing=np.nan):
        out, bad = np.empty(self.rows, np.float64), C.c_int64(-1)
        if self._L.arp_cif_column_f64(self._h, self.tags.index(item), float(missing), _p(out), C.byref(bad)) != ARP_OK:
            raise ValueError(f'{self.category}{item}: row {bad.value} is not a number')
        return out

    def ints(self, item, missing=0):
        out, bad = np.empty(self.rows, np.int64), C.c_int64(-1)
        if self._L.arp_cif_column_i64(self._h, self.tags.index(item), int(missing), _p(out), C.byref(bad)) != ARP_OK:
            raise ValueError(f'{self.category}{item}: row {bad.value} is not an integer')
        return out


BAG_SORT_MAX = 8192       # ARP_BAG_SORT_MAX of include/arpeggio_hip.h
KERNEL_SLOTS = ('bin', 'scan', 'scatter', 'unused', 'search', 'sift', 'mark_search', 'planes')


class Context:
    """One arp_ctx: one GPU, one HIP stream, device-resident structure."""

    def __init__(self, device: int = 0):
        self._L = load()
        h = C.c_void_p()
        rc = self._L.arp_create(int(device), C.byref(h))
        if rc != ARP_OK:
            msg = self._L.arp_last_error(None).decode()
            raise NativeLibraryError(f'arp_create(device={device}) failed ({rc}): {msg}')
        self._h = h
        self.device = device
        self.n = self.n_rings = self.n_amides = 0
        # the five bag sizes of a pass land here (one context per host thread: no sharing); a ctypes array made once costs the
        # call nothing, a NumPy array + pointer per call cost ~2 us of a ~90 us pass
        self._counts = (C.c_int64 * 5)()
        self._counts_p = C.cast(self._counts, C.c_void_p)

    def close(self):
        if getattr(self, '_h', None):
            self._L.arp_destroy(self._h)
            self._h = None

    def set_grid_reuse(self, on=True):
        """Whole-structure passes keep their contact grid (default); ``False``: every pass builds it (measurements)."""
        self._check(self._L.arp_set_grid_reuse(self._h, int(bool(on))), 'arp_set_grid_reuse')

    def set_sort_after_pass(self, on=True):
        """``run_launch`` / ``run_wait`` enqueue the canonical sort of the atom-atom bag before they return (for callers that
        fetch the sorted bag next: ``fetch_packed``)."""
        self._check(self._L.arp_set_sort_after_pass(self._h, int(bool(on))), 'arp_set_sort_after_pass')

    def set_packed_layout(self, rows=False):
        """Layout of the atom-atom bag in ``fetch_packed`` (arp_set_packed_layout): records (a bgn id per record) or ROWS — ``row``:
        N + 1 offsets, the records of atom a are [row[a], row[a + 1]) — 4 bytes per record less over PCIe.  With rows the returned
        bag is a ``RowsBag``: ``bag['i']`` expands the offsets on first use (np.repeat), everything else is as before."""
        self._check(self._L.arp_set_packed_layout(self._h, 1 if rows else 0), 'arp_set_packed_layout')
        self._packed_rows = bool(rows)

    def device_synchronize(self):
        """Everything enqueued on this context's GPU has completed (the bracket of a timed region)."""
        self._check(self._L.arp_device_synchronize(self._h), 'arp_device_synchronize')

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc == ARP_OK:
            return
        msg = self._L.arp_last_error(self._h).decode()
        if rc == ARP_E_XBOND_NBR:
            # the reference dereferences None here (utils.py:173)
            raise AttributeError("'NoneType' object has no attribute 'GetId'")
        if rc == ARP_E_ARG:
            raise ValueError(f'{what}: {msg}')
        err = NativeLibraryError(f'{what} failed ({rc}): {msg}')
        err.code = rc
        raise err

    # ---- inputs ----
    def set_complex(self, pc):
        """Upload a PackedComplex (everything initialize() leaves behind)."""
        L, h = self._L, self._h
        self._keep = pc
        self._check(L.arp_set_atoms(h, pc.n_atoms, _p(pc.xyz), _p(pc.vdw), _p(pc.cov), _p(pc.type_mask), _p(pc.flags),
                                    _p(pc.res_id)), 'arp_set_atoms')
        self._check(L.arp_set_residues(h, pc.n_residues, _p(pc.res_flags), _p(pc.res_prev), _p(pc.res_next)), 'arp_set_residues')
        self._check(L.arp_set_bonds(h, _p(pc.bond_off), _p(pc.bond_idx)), 'arp_set_bonds')
        self._check(L.arp_set_hydrogens(h, _p(pc.h_off), _p(pc.h_xyz)), 'arp_set_hydrogens')
        self._check(L.arp_set_single_bond_neighbours(h, _p(pc.sb_nbr)), 'arp_set_single_bond_neighbours')
        self._check(L.arp_set_rings(h, pc.n_rings, _p(pc.ring_center), _p(pc.ring_normal), _p(pc.ring_res)), 'arp_set_rings')
        self._check(L.arp_set_amides(h, pc.n_amides, _p(pc.amide_center), _p(pc.amide_normal), _p(pc.amide_res)), 'arp_set_amides')
        self.n, self.n_rings, self.n_amides = pc.n_atoms, pc.n_rings, pc.n_amides

    def set_batch(self, pcs, selections=None):
        """Several structures in one pass (arpeggio_amd.batch): uploads the concatenation of ``pcs`` and tells the library
        where each structure begins.  ``selections``: per-structure uint8 masks (None = whole structure).  Returns the
        offsets that ``run_batch`` / ``batch.split_*`` use."""
        from . import batch as _batch
        big, off = _batch.concat_complexes(pcs)
        self.set_complex(big)
        self.declare_batch(off)
        if selections is not None:
            sel = np.concatenate([np.ones(pc.n_atoms, np.uint8) if s_ is None else np.asarray(s_, np.uint8) for pc, s_ in zip(pcs, selections)])
            self.set_selection(sel)
        return off

    def declare_batch(self, off):
        """Tell the library how the resident (concatenated) structure splits into structures: ``off`` as returned by
        ``batch.concat_complexes`` (after ``set_complex`` / ``set_blob`` of the concatenation)."""
        a, r, m, bx = (np.ascontiguousarray(off[k]) for k in ('atom', 'ring', 'amide', 'boxes'))
        self._check(self._L.arp_set_batch(self._h, len(a) - 1, _p(a), _p(r), _p(m), _p(bx)), 'arp_set_batch')
        self._batch = off

    def run_batch(self, cutoff=5.0, vdw_comp=0.1, include_sequence_adjacent=False, expand_radius=6.0, fetch=True):
        """run_arpeggio on every structure of the resident batch in ONE pass.  Returns a list of per-structure dicts
        {atom_atom, atom_plane, plane_plane, group_group, group_plane} with structure-local ids, canonically sorted."""
        from . import batch as _batch
        counts = self.run_launch(cutoff, vdw_comp, include_sequence_adjacent, expand_radius)
        if not fetch:
            return counts
        off = self._batch
        per = [dict() for _ in range(len(off['atom']) - 1)]
        for s_, d in enumerate(_batch.split_atom_contacts(self.atom_contacts_fetch(counts['atom_atom'], sort=True), off)):
            per[s_]['atom_atom'] = d
        for name in ('atom_plane', 'plane_plane', 'group_group', 'group_plane'):
            for s_, d in enumerate(_batch.split_bag(name, self.fetch_bag(name, sort=True), off)):
                per[s_][name] = d
        return per

    def set_blob(self, blob, counts=None):
        """Upload a structure packed by ``pack_blob`` (one host-to-device copy); ``blob`` must stay alive during the call."""
        self._keep = blob
        self._check(self._L.arp_set_blob(self._h, _p(blob), int(blob.nbytes)), 'arp_set_blob')
        hdr = BlobHeader.from_buffer_copy(blob[:C.sizeof(BlobHeader)].tobytes())
        self.n, self.n_rings, self.n_amides = int(hdr.n), int(hdr.nring), int(hdr.namide)

    def get_blob(self):
        """The resident structure read back in blob form (uint8 array; ``unpack_blob`` gives the arrays)."""
        nb = C.c_uint64()
        self._check(self._L.arp_get_blob(self._h, None, 0, C.byref(nb)), 'arp_get_blob')
        buf = np.empty(int(nb.value), np.uint8)
        self._check(self._L.arp_get_blob(self._h, _p(buf), int(buf.nbytes), C.byref(nb)), 'arp_get_blob')
        return buf

    # ---- exchange between shards: RCCL behind the C ABI (include/arpeggio_hip.h, arp_comm_*)
    @staticmethod
    def comm_unique_id():
        """128-byte id of a new communicator (rank 0 calls this and hands the bytes to the other ranks)."""
        buf = np.zeros(128, np.uint8)
        rc = load().arp_comm_unique_id(_p(buf), 128)
        if rc != ARP_OK:
            raise NativeLibraryError('arp_comm_unique_id failed: ' + load().arp_last_error(None).decode())
        return buf

    def comm_init(self, rank, world, unique_id):
        uid = np.ascontiguousarray(unique_id, np.uint8)
        if uid.size != 128:
            raise ValueError('the unique id has 128 bytes')
        self._check(self._L.arp_comm_init(self._h, int(rank), int(world), _p(uid)), 'arp_comm_init')
        self.comm_rank, self.comm_world = int(rank), int(world)

    def comm_info(self):
        """(rank, world) of the context's RCCL communicator as the library sees it (arp_comm_info)."""
        r, w = C.c_int(-1), C.c_int(0)
        self._check(self._L.arp_comm_info(self._h, C.byref(r), C.byref(w)), 'arp_comm_info')
        return int(r.value), int(w.value)

    def comm_destroy(self):
        self._check(self._L.arp_comm_destroy(self._h), 'arp_comm_destroy')

    def shard_exchange_faces(self, left, right):
        """``left`` / ``right``: (device pointer, bytes) from ``shard_pack_face`` or None.  Returns {-1: (ptr, bytes), +1: ...}
        of what the neighbours sent (buffers owned by the context)."""
        lp, lb = left if left else (0, 0)
        rp, rb = right if right else (0, 0)
        out = np.zeros(4, np.uint64)
        self._check(self._L.arp_shard_exchange_faces(self._h, lp, lb, rp, rb, _p(out)), 'arp_shard_exchange_faces')
        got = {}
        if out[1]:
            got[-1] = (int(out[0]), int(out[1]))
        if out[3]:
            got[+1] = (int(out[2]), int(out[3]))
        return got

    def shard_set_exchange_lists(self, send_left, send_right, recv_left, recv_right):
        a = [np.ascontiguousarray(x if x is not None else [], np.int32) for x in (send_left, send_right, recv_left, recv_right)]
        self._keep_lists = a
        self._check(self._L.arp_shard_set_exchange_lists(self._h, _p(a[0]), len(a[0]), _p(a[1]), len(a[1]), _p(a[2]), len(a[2]), _p(a[3]), len(a[3])),
                    'arp_shard_set_exchange_lists')

    def shard_exchange_plus(self):
        self._check(self._L.arp_shard_exchange_plus(self._h), 'arp_shard_exchange_plus')

    def shard_reduce_residue_sets(self):
        self._check(self._L.arp_shard_reduce_residue_sets(self._h), 'arp_shard_reduce_residue_sets')

    # ---- shard assembled on the device (sharding.make_shard_device)
    def shard_set_home(self, records):
        self._check(self._L.arp_shard_set_home(self._h, _p(records), int(records.nbytes)), 'arp_shard_set_home')

    def shard_pack_face(self, slot, x_lo, x_hi):
        """(device pointer, bytes) of the record buffer with the home items whose x lies in [x_lo, x_hi]."""
        ptr, nb = C.c_uint64(), C.c_uint64()
        self._check(self._L.arp_shard_pack_face(self._h, int(slot), float(x_lo), float(x_hi), C.byref(ptr), C.byref(nb)), 'arp_shard_pack_face')
        return int(ptr.value), int(nb.value)

    def shard_assemble(self, left, right, n_res_global):
        """``left`` / ``right``: (device pointer, bytes) of the halo buffers received from the neighbours, or None."""
        lp, lb = left[:2] if left else (0, 0)       # (a third element — the owner of the memory — is the caller's business)
        rp, rb = right[:2] if right else (0, 0)
        counts = np.zeros(3, np.int64)
        self._check(self._L.arp_shard_assemble(self._h, int(lp), int(lb), int(rp), int(rb), int(n_res_global), _p(counts)), 'arp_shard_assemble')
        self.n, self.n_rings, self.n_amides = int(counts[0]), int(counts[1]), int(counts[2])
        return counts

    def shard_layout(self):
        out = dict(global_id=np.empty(self.n, np.int32), origin=np.empty(self.n, np.int8), sel=np.empty(self.n, np.uint8),
                   ring_gid=np.empty(self.n_rings, np.int32), ring_origin=np.empty(self.n_rings, np.int8),
                   amide_gid=np.empty(self.n_amides, np.int32), amide_origin=np.empty(self.n_amides, np.int8))
        self._check(self._L.arp_shard_layout(self._h, *[_p(out[k]) for k in ('global_id', 'origin', 'sel', 'ring_gid', 'ring_origin',
                                                                              'amide_gid', 'amide_origin')]), 'arp_shard_layout')
        return out

    def set_ownership(self, is_home=None, global_id=None):
        home = None if is_home is None else np.ascontiguousarray(is_home, np.uint8)
        gid = None if global_id is None else np.ascontiguousarray(global_id, np.int32)
        self._check(self._L.arp_set_ownership(self._h, _p(home), _p(gid)), 'arp_set_ownership')

    def set_group_ownership(self, ring_home, ring_gid, amide_home, amide_gid):
        a = [np.ascontiguousarray(ring_home, np.uint8), np.ascontiguousarray(ring_gid, np.int32),
             np.ascontiguousarray(amide_home, np.uint8), np.ascontiguousarray(amide_gid, np.int32)]
        self._check(self._L.arp_set_group_ownership(self._h, *[_p(x) for x in a]), 'arp_set_group_ownership')

    def set_single_bond_neighbour_coords(self, sb_xyz, sb_present):
        x = np.ascontiguousarray(sb_xyz, np.float32).reshape(-1, 3)
        f = np.ascontiguousarray(sb_present, np.uint8)
        self._check(self._L.arp_set_single_bond_neighbour_coords(self._h, _p(x), _p(f)), 'arp_set_single_bond_neighbour_coords')

    def set_selection_state(self, sel, plus, ring_sel, ring_plus, amide_sel, amide_plus):
        a = [np.ascontiguousarray(x, np.uint8) for x in (sel, plus, ring_sel, ring_plus, amide_sel, amide_plus)]
        self._check(self._L.arp_set_selection_state(self._h, *[_p(x) for x in a]), 'arp_set_selection_state')

    BUF_PLUS, BUF_RES_SETS = 0, 1

    def device_buffer(self, which):
        """(device pointer, bytes) of a context buffer, for torch tensors that alias it (sharded runs)."""
        ptr, nb = C.c_uint64(0), C.c_int64(0)
        self._check(self._L.arp_device_buffer(self._h, int(which), C.byref(ptr), C.byref(nb)), 'arp_device_buffer')
        return int(ptr.value), int(nb.value)

    def run_stage(self, stage, cutoff=5.0, vdw_comp=0.1, include_sequence_adjacent=False, expand_radius=6.0):
        counts = np.zeros(5, np.int64)
        self._check(self._L.arp_run_stage(self._h, int(stage), float(cutoff), float(vdw_comp), int(bool(include_sequence_adjacent)),
                                          float(expand_radius), _p(counts)), 'arp_run_stage')
        return dict(atom_atom=int(counts[0]), plane_plane=int(counts[1]), atom_plane=int(counts[2]),
                    group_group=int(counts[3]), group_plane=int(counts[4]))

    def launch_bag(self, name):
        cnt = C.c_int64(0)
        self._check(getattr(self._L, f'arp_{name}_launch')(self._h, C.byref(cnt)), f'arp_{name}_launch')
        return int(cnt.value)

    # ---- searches ----
    def search_all(self, radius, active=None):
        act = None if active is None else np.ascontiguousarray(active, np.uint8)
        cap = max(16 * self.n, 1024)
        while True:
            oi, oj = np.empty(cap, np.int32), np.empty(cap, np.int32)
            cnt = C.c_int64(0)
            rc = self._L.arp_search_all(self._h, float(radius), _p(act), cap, _p(oi), _p(oj), C.byref(cnt))
            if rc == ARP_E_CAPACITY:
                cap = int(cnt.value)
                continue
            self._check(rc, 'arp_search_all')
            k = int(cnt.value)
            o = np.lexsort((oj[:k], oi[:k]))
            return oi[:k][o], oj[:k][o]

    def search(self, centers, radius):
        """``NeighborSearch.search(center, radius)`` for many centres: (centre index, atom index) of every atom within
        ``radius`` of a centre (all atoms, hydrogens included), sorted by centre then atom."""
        ctr = np.ascontiguousarray(centers, np.float64).reshape(-1, 3)
        cap = max(64 * len(ctr), 1024)
        while True:
            oc, oa = np.empty(cap, np.int32), np.empty(cap, np.int32)
            cnt = C.c_int64(0)
            rc = self._L.arp_search(self._h, float(radius), len(ctr), _p(ctr), cap, _p(oc), _p(oa), C.byref(cnt))
            if rc == ARP_E_CAPACITY:
                cap = int(cnt.value)
                continue
            self._check(rc, 'arp_search')
            k = int(cnt.value)
            return oc[:k].copy(), oa[:k].copy()

    def set_selection(self, in_selection):
        sel = np.ascontiguousarray(in_selection, np.uint8)
        if sel.shape != (self.n,):
            raise ValueError('in_selection must have one entry per atom')
        self._check(self._L.arp_set_selection(self._h, _p(sel)), 'arp_set_selection')

    def run_enqueue(self, cutoff=5.0, vdw_comp=0.1, include_sequence_adjacent=False, expand_radius=6.0):
        """First half of ``run_launch``: the pass is enqueued, the call returns.  Pair with ``run_wait``."""
        self._check(self._L.arp_run_enqueue(self._h, float(cutoff), float(vdw_comp), int(bool(include_sequence_adjacent)),
                                            float(expand_radius)), 'arp_run_enqueue')

    def run_wait(self):
        """Second half of ``run_launch``: waits for the enqueued pass, returns the five bag sizes."""
        rc = self._L.arp_run_wait(self._h, self._counts_p)
        if rc != ARP_OK:
            self._check(rc, 'arp_run_wait')
        c = self._counts
        return {'atom_atom': c[0], 'plane_plane': c[1], 'atom_plane': c[2], 'group_group': c[3], 'group_plane': c[4]}

    def run_launch(self, cutoff=5.0, vdw_comp=0.1, include_sequence_adjacent=False, expand_radius=6.0):
        """run_arpeggio on the resident structure; results stay in HBM.  Returns the five bag sizes."""
        rc = self._L.arp_run_launch(self._h, cutoff, vdw_comp, 1 if include_sequence_adjacent else 0, expand_radius, self._counts_p)
        if rc != ARP_OK:
            self._check(rc, 'arp_run_launch')
        c = self._counts
        return {'atom_atom': c[0], 'plane_plane': c[1], 'atom_plane': c[2], 'group_group': c[3], 'group_plane': c[4]}

    def make_selection_masks(self):
        """Masks computed by the last selection expansion (downloads only; no recomputation)."""
        L = self._L
        plus = np.empty(self.n, np.uint8)
        rs, rp = np.empty(self.n_rings, np.uint8), np.empty(self.n_rings, np.uint8)
        as_, ap = np.empty(self.n_amides, np.uint8), np.empty(self.n_amides, np.uint8)
        self._check(L.arp_get_selection(self._h, _p(plus), _p(rs), _p(rp), _p(as_), _p(ap)), 'arp_get_selection')
        return dict(plus=plus, ring_sel=rs, ring_plus=rp, amide_sel=as_, amide_plus=ap)

    def make_selection(self, in_selection=None, radius=6.0):
        sel = np.ones(self.n, np.uint8) if in_selection is None else np.ascontiguousarray(in_selection, np.uint8)
        if sel.shape != (self.n,):
            raise ValueError('in_selection must have one entry per atom')
        plus = np.empty(self.n, np.uint8)
        rs, rp = np.empty(self.n_rings, np.uint8), np.empty(self.n_rings, np.uint8)
        as_, ap = np.empty(self.n_amides, np.uint8), np.empty(self.n_amides, np.uint8)
        self._check(self._L.arp_make_selection(self._h, _p(sel), float(radius), _p(plus), _p(rs), _p(rp), _p(as_), _p(ap)),
                    'arp_make_selection')
        return dict(plus=plus, ring_sel=rs, ring_plus=rp, amide_sel=as_, amide_plus=ap)

    # ---- atom-atom contacts ----
    def atom_contacts_launch(self, cutoff=5.0, vdw_comp=0.1, include_sequence_adjacent=False):
        cnt = C.c_int64(0)
        self._check(self._L.arp_atom_contacts_launch(self._h, float(cutoff), float(vdw_comp), int(bool(include_sequence_adjacent)),
                                                     C.byref(cnt)), 'arp_atom_contacts_launch')
        return int(cnt.value)

    @staticmethod
    def pinned_contact_buffers(capacity):
        """Page-locked result buffers for ``atom_contacts_fetch(..., out=...)``: allocate once, reuse for every structure
        (the fetch then runs at PCIe speed; pinning memory costs more than one copy, so it is not done per call)."""
        cap = max(int(capacity), 1)
        return dict(i=pinned_empty(cap, np.int32), j=pinned_empty(cap, np.int32), dist=pinned_empty(cap, np.float32),
                    sift=pinned_empty(cap, np.uint16), ctype=pinned_empty(cap, np.uint8))

    def atom_contacts_fetch(self, count, sort=True, out=None):
        """Download the resident contact list.  ``out``: buffers from ``pinned_contact_buffers`` (the returned arrays
        are then views into them, valid until the next fetch into the same buffers)."""
        cap = max(int(count), 1)
        if out is not None and min(len(out[k]) for k in ('i', 'j', 'dist', 'sift', 'ctype')) >= cap:
            oi, oj, od, osf, oc = out['i'], out['j'], out['dist'], out['sift'], out['ctype']
            cap = min(len(a) for a in (oi, oj, od, osf, oc))
        else:
            oi, oj = np.empty(cap, np.int32), np.empty(cap, np.int32)
            od, osf, oc = np.empty(cap, np.float32), np.empty(cap, np.uint16), np.empty(cap, np.uint8)
        cnt = C.c_int64(0)
        if sort:      # canonical (i, j) order, made on the device (csrc/arp_sort.h)
            self._check(self._L.arp_atom_contacts_sort(self._h), 'arp_atom_contacts_sort')
        self._check(self._L.arp_atom_contacts_fetch(self._h, cap, _p(oi), _p(oj), _p(od), _p(osf), _p(oc), C.byref(cnt)),
                    'arp_atom_contacts_fetch')
        k = int(cnt.value)
        return dict(i=oi[:k], j=oj[:k], dist=od[:k], sift=osf[:k], ctype=oc[:k])

    def sort_contacts(self):
        """Put the resident atom-atom bag into the canonical (i, j) order (device radix sort; fetches after it deliver that order)."""
        self._check(self._L.arp_atom_contacts_sort(self._h), 'arp_atom_contacts_sort')

    # columns of the four ring / amide bags inside a packed fetch: (key, slot q of arp_fetch_packed's offsets, dtype)
    _PACKED_BAGS = (
        ('plane_plane', (('bgn', 0, np.int32), ('end', 1, np.int32), ('dist', 2, np.float64), ('dihedral', 3, np.float64),
                         ('theta_bgn', 4, np.float64), ('theta_end', 5, np.float64), ('type1', 9, np.uint8), ('type2', 10, np.uint8),
                         ('ctype', 11, np.uint8))),
        ('atom_plane', (('atom', 0, np.int32), ('ring', 1, np.int32), ('dist', 2, np.float64), ('theta', 3, np.float64),
                        ('mask', 9, np.uint8), ('ctype', 10, np.uint8))),
        ('group_group', (('bgn', 0, np.int32), ('end', 1, np.int32), ('dist', 6, np.float32), ('dihedral', 7, np.float32),
                         ('theta', 8, np.float32), ('ctype', 9, np.uint8))),
        ('group_plane', (('amide', 0, np.int32), ('ring', 1, np.int32), ('dist', 2, np.float64), ('dihedral', 3, np.float64),
                         ('theta', 4, np.float64), ('ctype', 9, np.uint8))),
    )

    def fetch_packed(self, buf=None, sort_bags=True):
        """All five result bags of the last pass with ONE device-to-host copy (arp_fetch_packed): the atom-atom bag in
        canonical order (sorted on the device) and the ring / amide bags behind it.  ``buf``: a page-locked uint8 buffer
        from ``pinned_empty`` (reused from structure to structure; the returned arrays are views into it, valid until the
        next fetch into it); grown when too small.  Returns ``(bags, buf)``."""
        counts = (C.c_int64 * 5)()
        offs = (C.c_uint64 * 53)()
        used = C.c_uint64(0)
        if buf is None:
            buf = pinned_empty(1 << 20, np.uint8)
        rc = self._L.arp_fetch_packed(self._h, _p(buf), buf.nbytes, counts, offs, C.byref(used))
        if rc == ARP_E_CAPACITY and int(used.value) > buf.nbytes:
            buf = pinned_empty(int(used.value) + int(used.value) // 4 + 4096, np.uint8)
            rc = self._L.arp_fetch_packed(self._h, _p(buf), buf.nbytes, counts, offs, C.byref(used))
        if rc == ARP_E_CAPACITY and int(used.value) == 0:
            # a result too large for one piece (an array or the whole beyond 4 GiB): bag by bag, as before arp_fetch_packed
            # (counts[0]: what arp_fetch_packed filled before it failed — self._counts is only fresh after run_launch / run_wait)
            try:
                aa = self.atom_contacts_fetch(int(counts[0]), sort=True)
            except NativeLibraryError as e:
                if getattr(e, 'code', None) != ARP_E_CAPACITY:
                    raise
                # 2^31 records or more: beyond the device sort's tables — unsorted fetch, canonical order on the host
                aa = self.atom_contacts_fetch(int(counts[0]), sort=False)
                order = np.lexsort((aa['j'], aa['i']))
                aa = {k: v[order] for k, v in aa.items()}
            bags = {'atom_atom': aa}
            for name, _ in self._PACKED_BAGS:
                bags[name] = self.fetch_bag(name, sort=sort_bags)
            return bags, buf
        self._check(rc, 'arp_fetch_packed')

        def view(off, dtype, n):
            return np.frombuffer(buf, dtype, n, int(off)) if n else np.empty(0, dtype)
        k = int(counts[0])
        if getattr(self, '_packed_rows', False):
            bags = {'atom_atom': RowsBag(row=np.frombuffer(buf, np.int32, self.n + 1, int(offs[0])), j=view(offs[1], np.int32, k),
                                         dist=view(offs[2], np.float32, k), sift=view(offs[3], np.uint16, k), ctype=view(offs[4], np.uint8, k))}
        else:
            bags = {'atom_atom': dict(i=view(offs[0], np.int32, k), j=view(offs[1], np.int32, k), dist=view(offs[2], np.float32, k),
                                      sift=view(offs[3], np.uint16, k), ctype=view(offs[4], np.uint8, k))}
        for b, (name, cols) in enumerate(self._PACKED_BAGS):
            m = int(counts[1 + b])
            res = {key: view(offs[5 + 12 * b + q], dt, m) for key, q, dt in cols}
            # (every bag arrives in its canonical order, made on the device: one block's bitonic network for a bag of up to
            # BAG_SORT_MAX records, the radix passes of the atom-atom bag beyond that)
            bags[name] = res
        return bags, buf

    def atom_contacts(self, cutoff=5.0, vdw_comp=0.1, include_sequence_adjacent=False, sort=True):
        n = self.atom_contacts_launch(cutoff, vdw_comp, include_sequence_adjacent)
        out = self.atom_contacts_fetch(n, sort=sort)
        out['stats'] = self.stats()
        return out

    def atom_accumulators(self):
        """Per-atom sift masks (n,4) and hbond/polar counters (n,8) of the last contact launch."""
        sift = np.zeros((max(self.n, 1), 4), np.uint16)
        cnt = np.zeros((max(self.n, 1), 8), np.int32)
        self._check(self._L.arp_atom_accumulators(self._h, _p(sift), _p(cnt)), 'arp_atom_accumulators')
        return dict(sift=sift[:self.n], counts=cnt[:self.n])

    def atom_integer_sifts(self):
        """update_atom_integer_sift (U:224-242) under the canonical pair order: uint8 [n, 4, 15] =
        integer_sift / _inter_only / _intra_only / _water_only of the last contact launch."""
        out = np.zeros((max(self.n, 1), 4, 15), np.uint8)
        self._check(self._L.arp_atom_integer_sifts(self._h, _p(out)), 'arp_atom_integer_sifts')
        return out[:self.n]

    # ---- ring / amide contacts ----
    def pinned_bag_buffers(self, name, capacity):
        """Page-locked result buffers for ``fetch_bag(name, out=...)`` (allocate once, reuse for every structure)."""
        mk, keys, order = self._BAGS[name]
        cap = max(int(capacity), 64)
        return [pinned_empty(cap, a.dtype) for a in mk(1)]

    def fetch_bag(self, name, sort=True, out=None):
        """Fetch one of the ring/amide bags left in HBM by run_launch (no re-computation).  ``out``: buffers from
        ``pinned_bag_buffers`` (the returned arrays are then views into them)."""
        fn = getattr(self._L, f'arp_{name}_fetch')
        mk, keys, order = self._BAGS[name]
        arrs = None
        if out is not None:
            cap, cnt = min(len(a) for a in out), C.c_int64(0)
            rc = fn(self._h, cap, *[_p(x) for x in out], C.byref(cnt))
            if rc == ARP_OK:
                arrs = [a[:int(cnt.value)] for a in out]
            elif rc != ARP_E_CAPACITY:
                self._check(rc, f'arp_{name}_fetch')
        if arrs is None:
            arrs = self._grow(lambda cap, r, cnt: fn(self._h, cap, *[_p(x) for x in r], C.byref(cnt)), mk, f'arp_{name}_fetch', 64)
        res = dict(zip(keys, arrs))
        if sort:
            o = np.lexsort((res[order[1]], res[order[0]]))
            res = {k: v[o] for k, v in res.items()}
        return res

    def _grow(self, call, arrays_factory, what, guess):
        cap = max(int(guess), 64)
        while True:
            arrs = arrays_factory(cap)
            cnt = C.c_int64(0)
            rc = call(cap, arrs, cnt)
            if rc == ARP_E_CAPACITY:
                cap = int(cnt.value)
                continue
            self._check(rc, what)
            k = int(cnt.value)
            return [a[:k] for a in arrs]

    def atom_plane(self):
        def mk(cap):
            return [np.empty(cap, np.int32), np.empty(cap, np.int32), np.empty(cap, np.float64), np.empty(cap, np.float64),
                    np.empty(cap, np.uint8), np.empty(cap, np.uint8)]
        a = self._grow(lambda cap, r, cnt: self._L.arp_atom_plane(self._h, cap, *[_p(x) for x in r], C.byref(cnt)), mk,
                       'arp_atom_plane', 8 * self.n_rings)
        out = dict(atom=a[0], ring=a[1], dist=a[2], theta=a[3], mask=a[4], ctype=a[5])
        o = np.lexsort((out['atom'], out['ring']))
        return {k: v[o] for k, v in out.items()}

    def plane_plane(self):
        def mk(cap):
            return [np.empty(cap, np.int32), np.empty(cap, np.int32)] + [np.empty(cap, np.float64) for _ in range(4)] + \
                   [np.empty(cap, np.uint8) for _ in range(3)]
        a = self._grow(lambda cap, r, cnt: self._L.arp_plane_plane(self._h, cap, *[_p(x) for x in r], C.byref(cnt)), mk,
                       'arp_plane_plane', 16 * self.n_rings)
        out = dict(bgn=a[0], end=a[1], dist=a[2], dihedral=a[3], theta_bgn=a[4], theta_end=a[5], type1=a[6], type2=a[7], ctype=a[8])
        o = np.lexsort((out['end'], out['bgn']))   # = the reference's creation order (ring-id order)
        return {k: v[o] for k, v in out.items()}

    def group_group(self):
        def mk(cap):
            return [np.empty(cap, np.int32), np.empty(cap, np.int32)] + [np.empty(cap, np.float32) for _ in range(3)] + \
                   [np.empty(cap, np.uint8)]
        a = self._grow(lambda cap, r, cnt: self._L.arp_group_group(self._h, cap, *[_p(x) for x in r], C.byref(cnt)), mk,
                       'arp_group_group', 8 * self.n_amides)
        out = dict(bgn=a[0], end=a[1], dist=a[2], dihedral=a[3], theta=a[4], ctype=a[5])
        o = np.lexsort((out['end'], out['bgn']))
        return {k: v[o] for k, v in out.items()}

    def group_plane(self):
        def mk(cap):
            return [np.empty(cap, np.int32), np.empty(cap, np.int32)] + [np.empty(cap, np.float64) for _ in range(3)] + \
                   [np.empty(cap, np.uint8)]
        a = self._grow(lambda cap, r, cnt: self._L.arp_group_plane(self._h, cap, *[_p(x) for x in r], C.byref(cnt)), mk,
                       'arp_group_plane', 8 * self.n_amides)
        out = dict(amide=a[0], ring=a[1], dist=a[2], dihedral=a[3], theta=a[4], ctype=a[5])
        o = np.lexsort((out['ring'], out['amide']))
        return {k: v[o] for k, v in out.items()}

    _BAGS = {
        'atom_plane': (lambda cap: [np.empty(cap, np.int32), np.empty(cap, np.int32), np.empty(cap, np.float64),
                                    np.empty(cap, np.float64), np.empty(cap, np.uint8), np.empty(cap, np.uint8)],
                       ('atom', 'ring', 'dist', 'theta', 'mask', 'ctype'), ('ring', 'atom')),
        'plane_plane': (lambda cap: [np.empty(cap, np.int32), np.empty(cap, np.int32)] + [np.empty(cap, np.float64) for _ in range(4)]
                        + [np.empty(cap, np.uint8) for _ in range(3)],
                        ('bgn', 'end', 'dist', 'dihedral', 'theta_bgn', 'theta_end', 'type1', 'type2', 'ctype'), ('bgn', 'end')),
        'group_group': (lambda cap: [np.empty(cap, np.int32), np.empty(cap, np.int32)] + [np.empty(cap, np.float32) for _ in range(3)]
                        + [np.empty(cap, np.uint8)],
                        ('bgn', 'end', 'dist', 'dihedral', 'theta', 'ctype'), ('bgn', 'end')),
        'group_plane': (lambda cap: [np.empty(cap, np.int32), np.empty(cap, np.int32)] + [np.empty(cap, np.float64) for _ in range(3)]
                        + [np.empty(cap, np.uint8)],
                        ('amide', 'ring', 'dist', 'dihedral', 'theta', 'ctype'), ('amide', 'ring')),
    }

    # ---- measurement ----
    def stats(self):
        s = np.zeros(8, np.int64)
        self._check(self._L.arp_get_stats(self._h, _p(s)), 'arp_get_stats')
        return dict(candidates=int(s[0]), accepted=int(s[1]), emitted=int(s[2]), binned=int(s[3]), cells=int(s[4]),
                    expand_candidates=int(s[5]), expand_hits=int(s[6]))

    def set_profiling(self, on=True):
        self._check(self._L.arp_set_profiling(self._h, int(on)), 'arp_set_profiling')

    def kernel_times(self, reset=False):
        ms, ln = np.zeros(8, np.float64), np.zeros(8, np.int64)
        self._check(self._L.arp_get_kernel_times(self._h, _p(ms), _p(ln), int(reset)), 'arp_get_kernel_times')
        return {name: dict(ms=float(ms[k]), launches=int(ln[k])) for k, name in enumerate(KERNEL_SLOTS)}

    # ---- geometric part of initialize() (needs the atoms: set_complex / arp_set_atoms first) ----
    def ring_geometry(self, ring_atoms):
        """I:1697-1733: (center f64 [R,3], normal f64 [R,3]) of rings given as lists of atom indices in ring order."""
        nr = len(ring_atoms)
        off = np.zeros(nr + 1, np.int32)
        off[1:] = np.cumsum([len(a) for a in ring_atoms])
        idx = np.ascontiguousarray(np.concatenate([np.asarray(a, np.int32) for a in ring_atoms]) if nr else np.zeros(0, np.int32))
        ctr, nrm = np.zeros((max(nr, 1), 3), np.float64), np.zeros((max(nr, 1), 3), np.float64)
        self._check(self._L.arp_ring_geometry(self._h, nr, _p(off), _p(idx), _p(ctr), _p(nrm)), 'arp_ring_geometry')
        return ctr[:nr], nrm[:nr]

    def amide_geometry(self, amide_atoms):
        """I:1531-1589: (center f32 [A,3], normal f32 [A,3]) of amide groups given as [N, C, O, CA] atom indices."""
        at = np.ascontiguousarray(np.asarray(amide_atoms, np.int32).reshape(-1, 4))
        na = at.shape[0]
        ctr, nrm = np.zeros((max(na, 1), 3), np.float32), np.zeros((max(na, 1), 3), np.float32)
        self._check(self._L.arp_amide_geometry(self._h, na, _p(at), _p(ctr), _p(nrm)), 'arp_amide_geometry')
        return ctr[:na], nrm[:na]

    def ring_residues(self, ring_center):
        """I:1453-1492: (ring_res i32 [R], shortest distance f64 [R]) — residue of the nearest atom within 3.0 A, -1: none."""
        ctr = np.ascontiguousarray(np.asarray(ring_center, np.float64).reshape(-1, 3))
        nr = ctr.shape[0]
        res, dist = np.full(max(nr, 1), -1, np.int32), np.zeros(max(nr, 1), np.float64)
        self._check(self._L.arp_ring_residues(self._h, nr, _p(ctr), _p(res), _p(dist)), 'arp_ring_residues')
        return res[:nr], dist[:nr]

    def set_whole_structure(self, on=True):
        """Sharded runs without a selection: assert that the selection is the whole global structure (no exchange needed)."""
        self._check(self._L.arp_set_whole_structure(self._h, int(bool(on))), 'arp_set_whole_structure')

    def host_times(self, reset=False):
        """Per-pass host cost of run_launch: (enqueue_us, wait_us) averaged over the passes since the last reset."""
        us, n = np.zeros(2, np.float64), np.zeros(1, np.int64)
        self._check(self._L.arp_get_host_times(self._h, _p(us), _p(n), int(reset)), 'arp_get_host_times')
        k = max(int(n[0]), 1)
        return dict(enqueue_us=float(us[0]) / k, wait_us=float(us[1]) / k, passes=int(n[0]))

    def use_stream(self, stream_handle):
        self._check(self._L.arp_use_stream(self._h, int(stream_handle)), 'arp_use_stream')

    def stream_handle(self):
        return int(self._L.arp_stream_handle(self._h))
