"""Host-side mirror of the reference interface (no GPU): selection parser, labels, packing, C ABI symbols."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

from arpeggio_amd import synth
from arpeggio_amd.core import (AtomSerialError, InteractionComplex, NativeLibraryError, PackedComplex,
                               SelectionError, config, utils)
from helpers import tiny_complex

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pack_from_rows(rows):
    res_keys, res_id = {}, []
    for r in rows:
        key = (r['chain'], r['resname'], r['het'], r['seq'], r['icode'])
        res_keys.setdefault(key, len(res_keys))
        res_id.append(res_keys[key])
    keys = list(res_keys)
    pc = tiny_complex(np.zeros((len(rows), 3), np.float32), res_id=res_id,
                      res_flags=[config.R_POLYPEPTIDE if rows[res_id.index(k)]['poly'] else 0 for k in range(len(keys))])
    pc.atom_name = [r['name'] for r in rows]
    pc.element = [r['element'] for r in rows]
    pc.res_name = [k[1] for k in keys]
    pc.res_chain = [k[0] for k in keys]
    pc.res_seq = np.array([k[3] for k in keys], np.int32)
    pc.res_icode = [k[4] for k in keys]
    return pc


def test_selection_parser_matches_executed_reference(golden_dir):
    g = json.load(open(os.path.join(golden_dir, 'selection_parser.json')))
    pc = _pack_from_rows(g['atoms'])
    assert len(g['cases']) >= 20
    for case in g['cases']:
        if 'error' in case:
            with pytest.raises(SelectionError) as ei:
                utils.selection_parser(case['selectors'], pc)
            assert [str(a) for a in ei.value.args] == case['args'], case
        else:
            got = utils.selection_parser(case['selectors'], pc)
            assert got.tolist() == case['atoms'], case


def test_labels_and_json_schema_keys():
    pc = synth.config3(2000, seed=1).ensure_labels()
    a = utils.make_pymol_json(pc, atom=5)
    assert set(a) == {'label_comp_id', 'auth_seq_id', 'auth_asym_id', 'auth_atom_id', 'pdbx_PDB_ins_code'}
    assert isinstance(a['auth_seq_id'], int)
    r = utils.make_pymol_json(pc, residue=3)
    assert set(r) == {'label_comp_id', 'auth_seq_id', 'auth_asym_id', 'pdbx_PDB_ins_code'}
    assert utils.make_pymol_string(pc, atom=5).count('/') == 2
    with pytest.raises(TypeError):
        utils.make_pymol_json(pc)


def test_packed_complex_roundtrip_and_validation(tmp_path):
    pc = synth.config3(3000, seed=2)
    p = tmp_path / 'x.npz'
    pc.save(p)
    q = PackedComplex.load(p)
    for k in PackedComplex._ARRAYS:
        assert np.array_equal(getattr(pc, k), getattr(q, k)), k
    assert q.atom_name == pc.atom_name and q.component_types == pc.component_types
    assert all(np.array_equal(a, b) for a, b in zip(pc.ring_atoms, q.ring_atoms))
    with pytest.raises(ValueError):
        tiny_complex(np.zeros((3, 3)), res_id=[0, 1, 5], res_flags=[0, 0])
    bad = synth.config3(500, seed=1)
    bad.bond_off = bad.bond_off[:-1]
    with pytest.raises(ValueError):
        bad.validate()


def test_interaction_complex_constructor_and_checks(tmp_path):
    pc = synth.config3(1000, seed=4)
    ic = InteractionComplex(pc, 0.1, 5.0, 7.4)
    ic.structure_checks()
    assert ic.params.vdw_comp_factor == 0.1 and ic.params.interacting_threshold == 5.0
    assert ic.atom_contacts == [] and ic.get_contacts() == []
    pc.serial[10] = pc.serial[11]
    with pytest.raises(AtomSerialError):
        InteractionComplex(pc).structure_checks()
    with pytest.raises(IOError, match='File 1tqn_h.cif not found'):      # (P:58-59: the reader's own message for a missing file)
        InteractionComplex('1tqn_h.cif')
    with pytest.raises(NotImplementedError):                              # other formats: gemmi's conversion is out of scope
        InteractionComplex('1tqn_h.pdb')
    p = tmp_path / '1abc_h.npz'
    synth.config3(600, seed=1).save(p)
    assert InteractionComplex(str(p)).id == '1abc_h'


def test_synthetic_generator_is_deterministic():
    a, b = synth.config3(5000, seed=3), synth.config3(5000, seed=3)
    for k in PackedComplex._ARRAYS:
        assert np.array_equal(getattr(a, k), getattr(b, k)), k
    c = synth.config3(5000, seed=4)
    assert not np.array_equal(a.xyz, c.xyz)
    # splitmix64 known answers (seed 0: first outputs of the reference splitmix64 sequence)
    assert int(synth.splitmix64(0, [0])[0]) == 0xE220A8397B1DCDAF
    assert int(synth.splitmix64(0, [1])[0]) == 0x6E789E6AA1B965F4


def test_c_abi_library_exports_every_declared_symbol():
    """The shared library loads without a GPU and exports exactly what include/arpeggio_hip.h declares."""
    from arpeggio_amd import _capi
    hdr = open(os.path.join(ROOT, 'include', 'arpeggio_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(arp_[a-z_0-9]+)\s*\(', hdr))
    assert declared == set(_capi.SYMBOLS), declared ^ set(_capi.SYMBOLS)
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for s in declared:
        assert hasattr(lib, s), s
    L = _capi.load()
    assert b'gfx950' in L.arp_version()


def test_product_fails_loudly_without_gpu():
    from arpeggio_amd import _capi
    if _capi.device_count() > 0:
        pytest.skip('GPU present')
    with pytest.raises(NativeLibraryError):
        _capi.Context(0)
    ic = InteractionComplex(synth.config3(500, seed=1))
    with pytest.raises(NativeLibraryError):
        ic.run_arpeggio([], 5.0, 0.1, False)


def test_host_only_library_has_the_reader_and_the_writer_and_nothing_that_computes(tmp_path):
    """libarpeggio_host.so (g++, arpeggio_amd.build.build_host): the mmCIF reader and the JSON writer of include/arpeggio_hip.h —
    what the fixture generators need on a machine without hipcc — and no compute entry point."""
    import ctypes as C
    import subprocess
    from arpeggio_amd import _capi, build
    path = build.build_host()
    L = C.CDLL(path)
    for sym in _capi.HOST_SYMBOLS:
        assert hasattr(L, sym), sym
    exported = {ln.split()[-1] for ln in subprocess.run(['nm', '-D', '--defined-only', path], capture_output=True, text=True).stdout.splitlines() if ' T arp_' in ln}
    assert exported == set(_capi.HOST_SYMBOLS) | {'arp_host_version'}, exported
    _capi._declare_host_entry_points(L)
    text = b"data_t\nloop_\n_atom_site.id\n_atom_site.Cartn_x\n1 1.5\n2 'a b'\n"
    h, err = C.c_void_p(), C.create_string_buffer(64)
    assert L.arp_cif_open(text, len(text), b'_atom_site.', C.byref(h), err, 64) == 0 and L.arp_cif_rows(h) == 2 and L.arp_cif_cols(h) == 2
    L.arp_cif_close(h)


def test_product_never_imports_the_oracle():
    """A product path routed through oracle/ would void every parity claim."""
    for base, _, files in os.walk(os.path.join(ROOT, 'arpeggio_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(base, f)).read()
                assert not re.search(r'^\s*(import|from)\s+oracle\b', src, flags=re.M), f
                assert 'liborc' not in src and 'ref_c' not in src, f


def test_proteinlike_generator():
    a, b = synth.proteinlike(n_res=60, seed=2, n_waters=20), synth.proteinlike(n_res=60, seed=2, n_waters=20)
    assert np.array_equal(a.xyz, b.xyz) and np.array_equal(a.bond_idx, b.bond_idx)
    a.validate()
    h = (a.flags & config.F_HYDROGEN) != 0
    assert h.sum() > 100 and a.h_xyz.shape[0] == h.sum()                 # every hydrogen atom is also its parent's h_coord
    assert set(a.res_name) >= {'ALA', 'HEM', 'HOH'} and a.n_rings >= 4 and a.n_amides == 59
    sel = utils.selection_parser(['/A/508/'], a)
    assert len(sel) > 20 and {a.res_name[r] for r in a.res_id[sel]} == {'HEM'}
    # peptide-bonded neighbours are sequence neighbours
    assert a.res_next[0] == 1 and a.res_prev[1] == 0 and a.res_prev[0] == -1


def test_blob_layout_without_a_gpu():
    """arp_blob_size / arp_blob_layout are host-only: header, 16-byte aligned offsets, arrays where the header says."""
    from arpeggio_amd import _capi
    pc = synth.proteinlike(n_res=40, n_waters=12)
    blob = _capi.pack_blob(pc, pinned=False)
    h = _capi.BlobHeader.from_buffer_copy(blob[:ctypes.sizeof(_capi.BlobHeader)].tobytes())
    assert h.magic == 0x31424F4C42505241 and h.bytes == blob.nbytes
    assert (h.n, h.nres, h.nring, h.namide) == (pc.n_atoms, pc.n_residues, pc.n_rings, pc.n_amides)
    offs = list(h.off)
    assert all(o % 16 == 0 for o in offs) and offs == sorted(offs) and offs[0] >= ctypes.sizeof(_capi.BlobHeader)
    x4 = np.frombuffer(blob, np.float32, 4 * pc.n_atoms, offs[0]).reshape(-1, 4)
    assert np.array_equal(x4[:, :3], pc.xyz) and not x4[:, 3].any()
    idx = np.frombuffer(blob, np.uint16, pc.n_atoms, offs[19])
    tab = np.frombuffer(blob, np.float64, 512, offs[20]).reshape(256, 2)
    assert np.array_equal(tab[idx, 0], pc.vdw) and np.array_equal(tab[idx, 1], pc.cov) and h.n_rad == len(np.unique(tab[:h.n_rad], axis=0))
    assert np.array_equal(np.frombuffer(blob, np.float64, 3 * pc.h_xyz.shape[0], offs[11]).reshape(-1, 3), pc.h_xyz)
    assert list(h.lo) == pc.xyz.min(axis=0).astype(np.float64).tolist()
    L = _capi.load()
    assert L.arp_blob_size(-1, 0, 0, 0, 0, 0) == 0
    assert L.arp_blob_layout(None, 0, 1, 1, 0, 0, 0, 0) != 0


def test_native_json_writer_equals_json_dump(tmp_path):
    """arp_write_contacts_json (host-only) == json.dump(records, indent=4, sort_keys=True) of the Python exporter, on
    made-up result bags with awkward strings and distances."""
    from arpeggio_amd.core import export
    pc = synth.proteinlike(n_res=30, n_waters=10).ensure_labels()
    pc.atom_name[3] = 'O2\'"\\'
    pc.atom_name[4] = 'Cα'            # non-ASCII -> \\uXXXX
    pc.res_icode[1] = 'B'
    rng = np.random.default_rng(5)
    n = 400
    i = rng.integers(0, pc.n_atoms - 1, n).astype(np.int32)
    j = (i + 1 + rng.integers(0, 5, n)).clip(max=pc.n_atoms - 1).astype(np.int32)
    dist = (rng.random(n) * 6).astype(np.float32)
    dist[:6] = [0.0, 3.0, 4.999999, 2.675, 1e-5, 0.004999]
    bags = {'atom_atom': dict(i=i, j=j, dist=dist, sift=rng.integers(0, 1 << 15, n).astype(np.uint16),
                              ctype=rng.integers(0, 6, n).astype(np.uint8)),
            'plane_plane': dict(bgn=np.array([0, 1], np.int32), end=np.array([1, 2], np.int32), dist=np.array([4.7234, 5.0]),
                                type1=np.array([3, 9], np.uint8), type2=np.array([5, 255], np.uint8), ctype=np.array([2, 6], np.uint8)),
            'group_group': dict(bgn=np.array([0], np.int32), end=np.array([2], np.int32), dist=np.array([4.29], np.float32),
                                ctype=np.array([6], np.uint8))}
    bags['atom_atom']['sift'][7] = 0
    for case, b in (('full', bags), ('only_tail', {k: v for k, v in bags.items() if k != 'atom_atom'}), ('empty', {}),
                    ('only_atoms', {'atom_atom': bags['atom_atom']})):
        path = tmp_path / f'{case}.json'
        export.write_contacts_json(str(path), pc, b, pc.component_types)
        want = json.dumps(export.contacts_json(pc, b, pc.component_types), indent=4, sort_keys=True)
        assert open(path, encoding='utf-8').read() == want, case


def test_record_buffer_layout_without_a_gpu():
    """arp_records_size / arp_records_layout are host-only; the packer's structured dtypes are the C structs."""
    from arpeggio_amd import _capi, sharding
    pc = synth.proteinlike(n_res=40, n_waters=12)
    ids = np.arange(5, pc.n_atoms, 3)
    rec = sharding.pack_records(pc, ids, np.arange(pc.n_rings), np.arange(0, pc.n_amides, 2))
    buf = _capi.pack_records_buffer(rec, pinned=False)
    h = _capi.RecHeader.from_buffer_copy(buf[:ctypes.sizeof(_capi.RecHeader)].tobytes())
    assert h.magic == 0x3143455250524141 and h.bytes == buf.nbytes and ctypes.sizeof(_capi.RecHeader) == 4344
    assert (h.na, h.nh, h.nb, h.nring, h.namide) == (ids.size, rec['h_xyz'].shape[0], rec['bond_gid'].size, pc.n_rings, (pc.n_amides + 1) // 2)
    offs = list(h.off)
    assert all(o % 16 == 0 for o in offs) and offs == sorted(offs) and offs[0] >= ctypes.sizeof(_capi.RecHeader)
    back = _capi.unpack_records_buffer(buf)
    for k, v in rec.items():
        assert np.array_equal(back[k], v), k
    assert np.array_equal(back['h_start'][1:], np.cumsum(rec['h_cnt'])[:-1]) and back['h_start'][0] == 0
    tab = np.array(h.rad_tab[:2 * h.n_rad]).reshape(-1, 2)
    assert {(a, b) for a, b in zip(rec['vdw'], rec['cov'])} == {tuple(t) for t in tab}
    L = _capi.load()
    assert L.arp_records_size(-1, 0, 0, 0, 0) == 0 and L.arp_records_layout(None, 0, 1, 0, 0, 0, 0) != 0


def test_rings_listed_off_their_path_are_reported():
    """compute_plane_geometry sums cross products of consecutive ring atoms: a ring listed in another order must not be
    accepted silently (PackedComplex.rings_not_in_path_order)."""
    import copy
    pc = synth.proteinlike(n_res=60, n_waters=10)
    assert pc.n_rings > 0 and pc.rings_not_in_path_order() == []
    q = copy.deepcopy(pc)
    r = [int(x) for x in q.ring_atoms[0]]
    q.ring_atoms[0] = np.array([r[0], r[2], r[1]] + r[3:], np.int32)
    assert q.rings_not_in_path_order() == [0]
    assert synth.config3(3000).rings_not_in_path_order() == []      # ring atoms without bonds among them: not judged


def test_native_record_packer_equals_the_numpy_one():
    """arp_records_fill (host only) writes the bytes pack_records_buffer(pack_records(...)) writes."""
    from arpeggio_amd import _capi, sharding
    for pc in (synth.proteinlike(n_res=120, n_waters=40), synth.config3(4000, seed=2)):
        rng = np.random.default_rng(1)
        ids = np.sort(rng.choice(pc.n_atoms, pc.n_atoms // 2, replace=False))
        rid, mid = np.arange(0, pc.n_rings, 2), np.arange(1, pc.n_amides, 2)
        for sel in (None, (np.arange(pc.n_atoms) % 3 == 0).astype(np.uint8)):
            a = _capi.pack_records_buffer(sharding.pack_records(pc, ids, rid, mid, sel), pinned=False)
            b = _capi.pack_records_native(pc, ids, rid, mid, sel, pinned=False)
            assert np.array_equal(a, b)
    empty = _capi.pack_records_native(pc, [], [], [], None, pinned=False)
    assert _capi.unpack_records_buffer(empty)['gid'].size == 0
    with pytest.raises(ValueError):
        _capi.pack_records_native(pc, [5, 3], [], [], None, pinned=False)          # ids must ascend


def _build_c_example(tmp_path):
    import shutil
    import subprocess
    if not shutil.which('gcc'):
        pytest.skip('gcc not available')
    exe = str(tmp_path / 'c_abi_smoke')
    lib_dir = os.path.join(ROOT, 'arpeggio_amd', 'csrc')
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'), os.path.join(ROOT, 'examples', 'c_abi_smoke.c'),
                    '-o', exe, '-L', lib_dir, '-larpeggio_hip', f'-Wl,-rpath,{lib_dir}', '-lm'], check=True)
    return exe


def test_header_is_plain_c_and_the_c_example_links_and_runs(tmp_path):
    """include/arpeggio_hip.h is C99 (the boundary is a C ABI, not a C++ one) and examples/c_abi_smoke.c — no Python, no
    C++ — links against the library; without a GPU it runs the host-only entry points and stops at arp_create."""
    import subprocess
    from arpeggio_amd import _capi
    subprocess.run(['gcc', '-std=c99', '-pedantic', '-Wall', '-Werror', '-fsyntax-only', '-x', 'c', os.path.join(ROOT, 'include', 'arpeggio_hip.h')], check=True)
    out = subprocess.run([_build_c_example(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert ('GPU_OK' if _capi.device_count() > 0 else 'NO_GPU') in out.stdout


@pytest.mark.gpu
def test_c_example_on_the_gpu(tmp_path):
    import subprocess
    out = subprocess.run([_build_c_example(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and 'GPU_OK' in out.stdout, out.stdout + out.stderr


def test_c_record_builder_equals_the_python_loop():
    """csrc/arp_pyexport.c (optional CPython helper of get_contacts) builds the records the Python comprehension builds."""
    from arpeggio_amd.core import export
    if export._pyexport() is None:
        pytest.skip('helper not built')
    pc = synth.proteinlike(n_res=40, n_waters=12).ensure_labels()
    rng = np.random.default_rng(9)
    n = 3000
    i = rng.integers(0, pc.n_atoms - 1, n).astype(np.int32)
    bags = {'atom_atom': dict(i=i, j=(i + 1).astype(np.int32), dist=(rng.random(n) * 6).astype(np.float32),
                              sift=rng.integers(0, 1 << 15, n).astype(np.uint16), ctype=rng.integers(0, 6, n).astype(np.uint8))}
    fast = export.contacts_json(pc, bags, pc.component_types)
    saved, export._PYEXPORT = export._PYEXPORT, None
    try:
        slow = export.contacts_json(pc, bags, pc.component_types)
    finally:
        export._PYEXPORT = saved
    assert fast == slow and [list(r) for r in fast] == [list(r) for r in slow]          # same records, same key order
    assert fast[0]['bgn'] is not fast[1]['bgn'] and fast[0]['contact'] is not slow[0]['contact']
    fast[0]['bgn']['auth_atom_id'] = 'changed'                                          # records own their dicts
    assert export.contacts_json(pc, bags, pc.component_types)[0]['bgn']['auth_atom_id'] != 'changed'


def test_contact_table_writer_equals_the_csv_module(tmp_path):
    """write_contact_file assembles '<id>_contacts.csv' from text fragments; the bytes are what csv.writer (QUOTE_MINIMAL,
    '\\r\\n' line ends) writes row by row — atom names that need quoting and awkward float32 distances included."""
    from arpeggio_amd.core import export
    pc = synth.proteinlike(n_res=60, n_waters=20).ensure_labels()
    pc.atom_name[5], pc.atom_name[6], pc.atom_name[7] = 'C,1', 'O"2', "N'3"
    rng = np.random.default_rng(1)
    n = 5000
    i = rng.integers(0, pc.n_atoms - 1, n).astype(np.int32)
    i[:60] = rng.integers(4, 8, 60)
    dist = (rng.random(n) * 5).astype(np.float32)
    dist[:8] = [0.0, 1e-5, 3.0, 4.9999995, 1.5e-7, 2.5, 1e-4, 123456.79]
    bag = dict(i=i, j=(i + 1).astype(np.int32), dist=dist, sift=rng.integers(0, 1 << 15, n).astype(np.uint16),
               ctype=rng.integers(0, 6, n).astype(np.uint8))
    lab = export.Labels(pc, pc.component_types)
    for tag, rows in (('all', None), ('subset', np.nonzero(bag['ctype'] == 2)[0]), ('none', np.zeros(0, np.int64))):
        a, b = str(tmp_path / f'{tag}_a.csv'), str(tmp_path / f'{tag}_b.csv')
        export.write_contact_file(a, pc, lab, bag, rows)
        export.write_contact_file_csv_module(b, pc, lab, bag, rows)
        assert open(a, 'rb').read() == open(b, 'rb').read(), tag


def test_pack_from_reference_objects_round_trip():
    """A PackedComplex -> objects shaped like the reference's InteractionComplex after initialize() (the data holders the
    golden fixtures are made with) -> pack_from_reference_objects: the same arrays come back; rings listed in molecule
    order (as I:1728-1733 stores them) are returned in ring-path order."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import types
    import make_golden_core as G
    from arpeggio_amd.core.interactions import pack_from_reference_objects
    pc = synth.proteinlike(n_res=60, seed=9, n_waters=12)
    pc.ensure_labels()
    chains, residues = {}, []
    for r in range(pc.n_residues):
        chains.setdefault(pc.res_chain[r], G.Chain(pc.res_chain[r]))
        residues.append(G.Residue(r, chains[pc.res_chain[r]], pc.res_name[r], ' ', int(pc.res_seq[r]), pc.res_icode[r]))
    names = config.ATOM_TYPE_NAMES
    atoms = []
    for i in range(pc.n_atoms):
        f = int(pc.flags[i])
        a = G.Atom(i, residues[int(pc.res_id[i])], pc.atom_name[i], pc.element[i], pc.xyz[i].copy(), int(pc.serial[i]), bool(f & config.F_WATER))
        a.atom_types = {names[b] for b in range(12) if (int(pc.type_mask[i]) >> b) & 1}
        a.is_metal, a.is_halogen = bool(f & config.F_METAL), bool(f & config.F_HALOGEN)
        a.vdw_radius, a.cov_radius = float(pc.vdw[i]), float(pc.cov[i])
        a.h_coords = [np.array(pc.h_xyz[k], np.float64) for k in range(pc.h_off[i], pc.h_off[i + 1])]
        atoms.append(a)
    obatoms = [G.OBAtom(1000 + i, 1 if (int(pc.flags[i]) & config.F_HYDROGEN) else 6) for i in range(pc.n_atoms)]
    for i in range(pc.n_atoms):
        me = obatoms[i]
        for k in range(pc.bond_off[i], pc.bond_off[i + 1]):
            me.nbrs.append(obatoms[int(pc.bond_idx[k])])
        me.bonds = [G.OBBond(me, G.OBAtom(-1, 6), 2, False), G.OBBond(me, G.OBAtom(-2, 1), 1, False)]      # decoys: double bond, hydrogen
        if pc.sb_nbr[i] >= 0:
            me.bonds.append(G.OBBond(me, obatoms[int(pc.sb_nbr[i])], 1, False))
    for r in range(pc.n_residues):
        if int(pc.res_flags[r]) & config.R_POLYPEPTIDE:
            residues[r].is_polypeptide = True
        if int(pc.res_flags[r]) & config.R_HAS_SEQ:
            residues[r].prev_residue = residues[pc.res_prev[r]] if pc.res_prev[r] >= 0 else None
            residues[r].next_residue = residues[pc.res_next[r]] if pc.res_next[r] >= 0 else None
    st = G.Structure(residues)
    for r in range(pc.n_rings):      # molecule order = ascending atom index, as OBMolAtomIter delivers the members
        st.rings[r] = {'ring_id': r, 'center': pc.ring_center[r].copy(), 'normal': pc.ring_normal[r].copy(),
                       'atoms': [atoms[int(k)] for k in sorted(pc.ring_atoms[r].tolist())], 'residue': residues[int(pc.ring_res[r])] if pc.ring_res[r] >= 0 else None}
    for e in range(pc.n_amides):
        st.amides[e] = {'amide_id': e, 'center': pc.amide_center[e].copy(), 'normal': pc.amide_normal[e].copy(),
                        'atoms': [atoms[int(k)] for k in pc.amide_atoms[e]], 'residue': residues[int(pc.amide_res[e])]}
    ic = types.SimpleNamespace(id=pc.id, s_atoms=atoms, biopython_str=st, ob_mol=G.OBMol({a.oid: a for a in obatoms}),
                               bio_to_ob={a: obatoms[a.idx].oid for a in atoms}, ob_to_bio={obatoms[a.idx].oid: a for a in atoms},
                               component_types=dict(pc.component_types))
    back = pack_from_reference_objects(ic, ob=G.ob)
    for k in ('xyz', 'vdw', 'cov', 'type_mask', 'flags', 'res_id', 'res_flags', 'res_prev', 'res_next', 'bond_off', 'bond_idx', 'h_off',
              'h_xyz', 'sb_nbr', 'ring_center', 'ring_normal', 'ring_res', 'amide_center', 'amide_normal', 'amide_res', 'amide_atoms'):
        assert np.array_equal(getattr(back, k), getattr(pc, k)), k
    assert back.n_rings > 0 and not back.rings_not_in_path_order()
    for a, b in zip(back.ring_atoms, pc.ring_atoms):
        assert sorted(a.tolist()) == sorted(b.tolist())
