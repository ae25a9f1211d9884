"""InteractionComplex — drop-in counterpart of ``arpeggio.core.InteractionComplex`` for the
``run_arpeggio`` / ``get_contacts`` path (reference: arpeggio/core/interactions.py:36-347, 2063-2113).

Same class name, method names, positional signatures and result attributes; the work is
done by the HIP library behind include/arpeggio_hip.h on a device-resident PackedComplex.
What the reference computes with BioPython/OpenBabel/gemmi before the hot path
(``__init__`` parsing and ``initialize()``'s typing, I:37-105, 288-327) is the *input
contract* here: the constructor takes a PackedComplex (or a ``.npz`` written by
``PackedComplex.save``); ``pack_from_reference_objects`` documents how a reference-side
object maps onto it.

There is no CPU fallback: without the HIP library and a GPU every compute method raises
NativeLibraryError.
"""
from __future__ import annotations

import collections
import logging
import os

import numpy as np

from . import config, export, utils
from .exceptions import AtomSerialError, NativeLibraryError
from .packed import PackedComplex

# result records: same field names as the reference (I:19-33); atoms / residues are packed indices
AtomPlaneContact = collections.namedtuple('AtomPlaneContact',
                                          ['bgn_atom', 'end_res', 'end_res_atoms', 'distance', 'sifts', 'text'])
PlanePlaneContact = collections.namedtuple('PlanePlaneContact',
                                           ['bgn_id', 'bgn_res', 'bgn_res_atoms', 'end_id', 'end_res', 'end_res_atoms',
                                            'distance', 'contact_type', 'text'])
AtomAtomContact = collections.namedtuple('AtomAtomContact', ['bgn_atom', 'end_atom', 'sifts', 'contact_type', 'distance'])
Parameters = collections.namedtuple('Parameters', ['vdw_comp_factor', 'interacting_threshold', 'has_hydrogens', 'ph'])


def ring_path_order(atoms, adj):
    """The atoms of one ring re-ordered so that consecutive ones (and the last and the first) are bonded: a walk over the
    bond graph ``adj`` (list of neighbour lists by packed index) restricted to the ring's atoms, from its first atom.
    Left as it is when the atoms do not form a single cycle in ``adj``."""
    atoms = [int(a) for a in atoms]
    members = set(atoms)
    path, seen = [atoms[0]], {atoms[0]}
    while len(path) < len(atoms):
        nxt = [b for b in adj[path[-1]] if b in members and b not in seen]
        if not nxt:
            return np.array(atoms, np.int32)
        path.append(int(nxt[0]))
        seen.add(path[-1])
    if path[0] not in adj[path[-1]]:
        return np.array(atoms, np.int32)
    return np.array(path, np.int32)


def amide_majority_residue(res_id, amide_atoms):
    """I:1554-1559: the residue of an amide group = ``max(residues, key=residues.count)`` over its four atoms N, C, O,
    C-alpha — the most frequent one, the FIRST such in that order on a tie."""
    r = np.asarray(res_id)[np.asarray(amide_atoms).reshape(-1, 4)]                     # [A, 4]
    cnt = (r[:, :, None] == r[:, None, :]).sum(axis=2)                                  # occurrences of each entry
    return r[np.arange(len(r)), cnt.argmax(axis=1)].astype(np.int32)                    # argmax: first maximum


class InteractionComplex:
    def __init__(self, filename, vdw_comp=0.1, interacting=5.0, ph=7.4, device=0, allow_incomplete=False):
        """Args mirror I:37.  ``filename``: the path of an mmCIF file (I:37-105 — read by ``core.protein_reader.read_mmcif``:
        what the file alone determines; see ``allow_incomplete``), a PackedComplex, or the path of a packed ``.npz``.
        ``allow_incomplete``: a structure read from a file lacks what only the reference's OpenBabel preparation gives
        (``pc.incomplete``); a run that needs a missing part raises ``IncompleteStructureError`` — with True it logs a warning
        and goes on (the records that depend on the missing part are then not the reference's)."""
        self.allow_incomplete = bool(allow_incomplete)
        if isinstance(filename, PackedComplex):
            self.pc = filename
            self.id = filename.id
        elif isinstance(filename, (str, os.PathLike)) and str(filename).endswith('.npz'):
            self.pc = PackedComplex.load(filename)
            self.id = os.path.basename(str(filename)).split('.')[0]        # I:51
        elif isinstance(filename, (str, os.PathLike)) and str(filename).lower().endswith(('.cif', '.mmcif')):
            from . import protein_reader
            self.pc = protein_reader.read_mmcif(str(filename))             # I:54-60, P:258-411
            self.id = os.path.basename(str(filename)).split('.')[0]        # I:51
        else:
            raise NotImplementedError(
                'Only mmCIF files (.cif), PackedComplex objects and packed .npz files are read: the reference converts every other '
                'format to mmCIF with gemmi first (P:258-295), which is outside the accelerated path.')
        self.pc.ensure_labels()
        self.device = device
        self._ctx = None
        self.component_types = self.pc.component_types                     # I:70

        # helper structures (I:73-81) — packed atom / ring / amide indices
        self.selection = []
        self.selection_ring_ids = []
        self.selection_amide_ids = []
        self.selection_plus = []
        self.selection_plus_residues = []
        self.selection_plus_ring_ids = []
        self.selection_plus_amide_ids = []
        # result bags (I:84-88)
        self._bags = {}
        self.params = Parameters(vdw_comp_factor=vdw_comp, interacting_threshold=interacting,
                                 has_hydrogens=bool(np.any(self.pc.flags & config.F_HYDROGEN)) or self.pc.h_xyz.shape[0] > 0,
                                 ph=ph)                                    # I:90-94

    # region public methods
    def structure_checks(self):
        """I:109-118."""
        serials = self.pc.serial
        if len(serials) > len(set(serials.tolist())):
            raise AtomSerialError

    def address_ambiguities(self):
        """I:120-133 strikes the ASN / GLN / HIS side-chain keys from four lists of the typing dictionary BEFORE initialize()
        types the atoms of standard residues from it (I:1966-1983).  Here the types are part of the packed input: the atoms of
        standard residues are re-typed from the same dictionary with those keys struck (``typing.apply_protein_typing``, pinned
        against the executed reference); every other atom keeps the types it was packed with."""
        from . import typing
        self.pc.type_mask = typing.apply_protein_typing(self.pc, use_ambiguities=True)
        if self._ctx is not None:
            self._ctx.set_complex(self.pc)

    def initialize(self):
        """I:288-327 prepares per-atom state; here: create the GPU context and upload the pack.  A structure that came from a
        file (``read_mmcif``) gets the centres / normals / residues of its template rings and amide groups computed on the GPU
        (the geometric part of I:1453-1492, 1531-1589, 1697-1733)."""
        from .. import _capi
        incomplete = getattr(self.pc, 'incomplete', ())
        if 'added hydrogens' in incomplete and not self.params.has_hydrogens and self.pc.n_atoms:
            # I:99-105: a structure without hydrogens gets them from OpenBabel (AddHydrogens at the given pH) — every hbond /
            # weak hbond flag depends on them
            self._incomplete(['hydrogens (the file has none; the reference adds them with OpenBabel, I:99-105)'])
        if self._ctx is None:
            self._ctx = _capi.Context(self.device)
            self._ctx.set_sort_after_pass(True)      # run_arpeggio always fetches the sorted bags next
        self._ctx.set_complex(self.pc)
        if getattr(self.pc, 'plane_geometry_pending', False):
            self.compute_plane_geometry()
            self.pc.plane_geometry_pending = False
        logging.debug('Uploaded packed structure to the GPU.')

    def _incomplete(self, needs):
        """A run on a structure read without OpenBabel needs something the file does not give."""
        from .exceptions import IncompleteStructureError
        if not self.allow_incomplete:
            raise IncompleteStructureError(needs)
        logging.warning('Structure read without OpenBabel, going on as asked (allow_incomplete); missing: %s', '; '.join(needs))

    def _check_incomplete_after_run(self):
        """What of ``pc.incomplete`` the run that has just finished would have needed: atom types, rings and amide groups of the
        non-standard residues inside selection_plus (OpenBabel's SMARTS and ring perception, I:1697-1733, 1531-1589, 1966-1983),
        and the single-bond neighbour of a halogen that takes part in a contact (U:139-141, 173)."""
        pc = self.pc
        incomplete = getattr(pc, 'incomplete', ())
        untyped = getattr(pc, 'untyped_atoms', None)
        if not incomplete or untyped is None or not len(untyped):
            return
        needs = []
        aa = self._bags.get('atom_atom', {})
        if len(aa.get('i', ())):
            hit = untyped[aa['i']] | untyped[aa['j']]
            if hit.any():
                res = sorted({pc.res_name[r].strip() for r in np.unique(pc.res_id[np.concatenate([aa['i'][hit], aa['j'][hit]])]).tolist()
                              if r in set(pc.res_id[untyped].tolist())})
                needs.append(f'atom types of the non-standard residues {", ".join(res[:8])} ({int(hit.sum())} of the contacts found touch '
                             f'their untyped atoms; also their rings and amide groups, if any)')
            hal = (pc.flags & config.F_HALOGEN) != 0
            if hal.any() and (hal[aa['i']] | hal[aa['j']]).any() and (pc.sb_nbr[hal] < 0).any():
                needs.append('bonds inside non-standard residues (the single-bond neighbour of a halogen in contact, U:139-141, 173)')
        if needs:
            self._incomplete(needs)

    def compute_plane_geometry(self, assign_ring_residues=True):
        """The geometric part of the reference's initialize() on the GPU, for packs that carry the perceived rings /
        amide groups only as atom lists (``ring_atoms``, ``amide_atoms``): ring centre + normal (I:1697-1733), amide
        centre + normal (I:1531-1589) and, if asked, the ring -> residue assignment by nearest atom (I:1453-1492).
        Overwrites ``ring_center / ring_normal / ring_res`` and ``amide_center / amide_normal`` of the pack and
        uploads them."""
        if self._ctx is None:
            self.initialize()
        pc, ctx = self.pc, self._ctx
        if pc.n_rings and pc.ring_atoms:
            # the normal is a sum of cross products of CONSECUTIVE centre->atom vectors (OBRing::findCenterAndNormal walks
            # the ring path): atoms listed in another order (e.g. file order CG, CD1, CD2, CE1, CE2, CZ) cancel to noise
            bad = pc.rings_not_in_path_order()
            if bad:
                raise ValueError(f'ring_atoms of ring(s) {bad[:8]} are not in ring-path order (consecutive atoms are not bonded): '
                                 'store the OpenBabel ring path, or keep the ring_center / ring_normal of the pack')
            pc.ring_center, pc.ring_normal = ctx.ring_geometry(pc.ring_atoms)
            if assign_ring_residues:
                pc.ring_res, self.ring_residue_shortest_distance = ctx.ring_residues(pc.ring_center)
        if pc.n_amides and (pc.amide_atoms[:, :3] >= 0).all():
            pc.amide_center, pc.amide_normal = ctx.amide_geometry(pc.amide_atoms)
            if (pc.amide_atoms >= 0).all():
                pc.amide_res = amide_majority_residue(pc.res_id, pc.amide_atoms)
        ctx.set_complex(pc)

    def run_arpeggio(self, user_selections, interacting_cutoff, vdw_comp, include_sequence_adjacent):
        """I:329-347: selection + binding-site expansion, atom, ring and amide contacts (all on the GPU)."""
        if self._ctx is None:
            self.initialize()
        pc, ctx = self.pc, self._ctx
        # I:1395: no selectors -> the whole structure.  Extension: an integer array = the packed indices of the
        # selected atoms (what the parser would have returned), for callers that select by their own means
        if isinstance(user_selections, np.ndarray):
            idx = np.unique(user_selections.astype(np.int64))
        elif user_selections:
            idx = utils.selection_parser(user_selections, pc)
        else:
            idx = np.arange(pc.n_atoms, dtype=np.int64)
        if idx.size == 0:                                                   # I:1399-1401
            logging.error('Selection was empty.')
            raise AttributeError('Selection must not be empty.')
        mask = np.zeros(pc.n_atoms, np.uint8)
        mask[idx] = 1
        ctx.set_selection(mask)
        counts = ctx.run_launch(interacting_cutoff, vdw_comp, include_sequence_adjacent,
                                config.SELECTION_EXPANSION_RADIUS)
        logging.debug('Completed new NeighbourSearch.')
        masks = ctx.make_selection_masks()
        self.selection = idx
        self.selection_plus = np.nonzero(masks['plus'])[0]
        self.selection_plus_residues = np.unique(pc.res_id[self.selection_plus])
        self.selection_ring_ids = set(np.nonzero(masks['ring_sel'])[0].tolist())
        self.selection_plus_ring_ids = set(np.nonzero(masks['ring_plus'])[0].tolist())
        self.selection_amide_ids = set(np.nonzero(masks['amide_sel'])[0].tolist())
        self.selection_plus_amide_ids = set(np.nonzero(masks['amide_plus'])[0].tolist())
        # all five bags with one copy; the atom-atom bag arrives in the canonical (i, j) order, made on the device
        # (the arrays are views into a page-locked buffer of their own: the next run allocates another one)
        self._bags, _ = ctx.fetch_packed()
        self._check_incomplete_after_run()
        self.stats = ctx.stats()

    # ---- result bags as lists of the reference's namedtuples (built on demand) ----
    def _ring_names(self, r):
        return sorted(self.pc.atom_name[a] for a in self.pc.ring_atoms[r]) if self.pc.ring_atoms else []

    def _need_results(self):
        if 'atom_atom' not in self._bags:
            raise AttributeError('no results yet: call run_arpeggio() first')

    def _amide_names(self, a):
        return sorted(self.pc.atom_name[i] for i in self.pc.amide_atoms[a] if i >= 0)

    @property
    def atom_contacts(self):
        b = self._bags.get('atom_atom')
        if b is None:
            return []
        bits = {int(s): [(int(s) >> k) & 1 for k in range(15)] for s in np.unique(b['sift']).tolist()}
        with export.paused_gc():
            return [AtomAtomContact(i, j, list(bits[s]), config.CONTACT_TYPE_NAMES[c], d)
                    for i, j, s, c, d in zip(b['i'].tolist(), b['j'].tolist(), b['sift'].tolist(), b['ctype'].tolist(), b['dist'])]

    @property
    def plane_plane_contacts(self):
        b = self._bags.get('plane_plane')
        if b is None:
            return []
        out = []
        for k in range(len(b['bgn'])):
            types = [config.PLANE_PLANE_NAMES[b['type1'][k]]]
            if b['type2'][k] < config.PP_SAME:
                types.append(config.PLANE_PLANE_NAMES[b['type2'][k]])
            r1, r2 = int(b['bgn'][k]), int(b['end'][k])
            out.append(PlanePlaneContact(r1, int(self.pc.ring_res[r1]), self._ring_names(r1), r2, int(self.pc.ring_res[r2]),
                                         self._ring_names(r2), b['dist'][k], types, config.CONTACT_TYPE_NAMES[b['ctype'][k]]))
        return out

    @property
    def atom_plane_contacts(self):
        b = self._bags.get('atom_plane')
        if b is None:
            return []
        return [AtomPlaneContact(int(a), int(self.pc.ring_res[r]), self._ring_names(int(r)), d,
                                 [n for bit, n in enumerate(config.ATOM_PLANE_NAMES) if (int(m) >> bit) & 1],
                                 config.CONTACT_TYPE_NAMES[c])
                for a, r, d, m, c in zip(b['atom'], b['ring'], b['dist'], b['mask'], b['ctype'])]

    @property
    def group_group_contacts(self):
        b = self._bags.get('group_group')
        if b is None:
            return []
        return [PlanePlaneContact(int(a1), int(self.pc.amide_res[a1]), self._amide_names(int(a1)), int(a2),
                                  int(self.pc.amide_res[a2]), self._amide_names(int(a2)), d, ['AMIDEAMIDE'],
                                  config.CONTACT_TYPE_NAMES[c])
                for a1, a2, d, c in zip(b['bgn'], b['end'], b['dist'], b['ctype'])]

    @property
    def group_plane_contacts(self):
        b = self._bags.get('group_plane')
        if b is None:
            return []
        return [PlanePlaneContact(int(a), int(self.pc.amide_res[a]), self._amide_names(int(a)), int(r),
                                  int(self.pc.ring_res[r]), self._ring_names(int(r)), d, ['AMIDERING'],
                                  config.CONTACT_TYPE_NAMES[c])
                for a, r, d, c in zip(b['amide'], b['ring'], b['dist'], b['ctype'])]

    def residue_plane_sifts(self):
        """Per-residue integer SIFts of the ring / amide loops (I:1040-1057, 1171-1176, 1290-1291, 1371-1373).

        Returns int arrays indexed by residue: ``ring_ring_inter_integer_sift`` [NR,9] (FF..EF),
        ``ring_atom_inter_integer_sift`` / ``atom_ring_inter_integer_sift`` / ``mc_atom_ring_...`` / ``sc_atom_ring_...``
        [NR,5] (CARBONPI..METSULPHURPI), ``amide_amide_inter_integer_sift``, ``amide_ring_inter_integer_sift``,
        ``ring_amide_inter_integer_sift`` [NR,1].  Only contacts of type INTER between different residues count.
        """
        return residue_plane_sifts(self.pc, self._bags)

    # ---- per-atom / per-residue SIFt accumulators ----
    def atom_sifts(self):
        """Per-atom OR-accumulated SIFts of the last run (I:923-934, U:182-221), computed on the GPU from the resident
        contact list: dict of uint8 arrays [n_atoms, 15] ``sift``, ``sift_inter_only``, ``sift_intra_only``,
        ``sift_water_only`` and the int32 [n_atoms, 8] ``counts`` = actual hbonds {all, intra, inter, water} and actual
        polars {all, intra, inter, water} (I:821-852).  Rows of atoms outside ``selection_plus`` are zero."""
        acc = self._ctx.atom_accumulators()
        bits = (acc['sift'][:, :, None] >> np.arange(15, dtype=np.uint16)[None, None, :]) & 1
        names = ('sift', 'sift_inter_only', 'sift_intra_only', 'sift_water_only')
        out = {n: bits[:, k, :].astype(np.uint8) for k, n in enumerate(names)}
        out['counts'] = acc['counts']
        return out

    def atom_integer_sifts(self):
        """``atom.integer_sift`` / ``_inter_only`` / ``_intra_only`` / ``_water_only`` (U:224-242, called at I:924-925) as
        a uint8 array [n_atoms, 4, 15].  The reference's values are decided by the last pair its KD-tree delivers for each
        atom; here the delivery order is the canonical one (contacts sorted by (bgn, end)), see include/arpeggio_hip.h."""
        self._need_results()
        return self._ctx.atom_integer_sifts()

    def residue_sifts(self):
        """`_calc_residue_sifts` (I:471-574): dict name -> int array [n_residues, width] with the reference's attribute
        names (``sift``, ``mc_sift_inter_only``, ``integer_sift``, ``ring_ring_inter_integer_sift`` ...), in the column
        order of `write_residue_sifts`."""
        self._need_results()
        bits = self.atom_sifts()
        return export.residue_sift_table(self.pc, bits, self.atom_integer_sifts(), self.residue_plane_sifts())

    # ---- the legacy CSV tables (I:135-170, 349-366, 405-466) ----
    def write_atom_types(self, wd):
        """I:135-149: '<id>_atomtypes.csv'."""
        export.write_atom_types(wd, self.id, self.pc, self.component_types)
        logging.debug('Typed atoms.')

    def write_contacts(self, selection, wd):
        """I:151-170: '<id>_contacts.csv' (+ '<id>_bs_contacts.csv' when ``selection`` is not empty)."""
        self._need_results()
        export.write_contacts(wd, self.id, self.pc, self._bags['atom_atom'], self.component_types, len(selection) > 0)

    def write_atom_sifts(self, wd):
        """I:349-366, 576-606: '<id>_sifts.csv' (atom, 15 flags) and '<id>_specific_sifts.csv' (atom, inter / intra /
        water flags, 45 columns) for the atoms of selection_plus, in packed atom order (the reference iterates a set).
        The header row is the reference's (17 names whatever the row width)."""
        import csv
        a = self.atom_sifts()
        lab = export.Labels(self.pc, self.component_types)
        header = ['atom'] + list(config.SIFT_NAMES) + ['interacting_entities']
        rows_all, rows_spec = [], []
        for i in self.selection_plus.tolist():
            label = lab.atom_macro(i)
            rows_all.append([label] + a['sift'][i].tolist())
            rows_spec.append([label] + a['sift_inter_only'][i].tolist() + a['sift_intra_only'][i].tolist()
                             + a['sift_water_only'][i].tolist())
        for name, rows in ((self.id + '_sifts.csv', rows_all), (self.id + '_specific_sifts.csv', rows_spec)):
            with open(os.path.join(wd, name), 'w') as f:
                w = csv.writer(f, delimiter=',', quotechar='"', quoting=csv.QUOTE_MINIMAL)
                w.writerow(header)
                w.writerows(rows)

    def write_residue_sifts(self, wd):
        """I:435-466: '<id>_residue_sifts.csv', one row of 426 columns per residue of selection_plus."""
        export.write_residue_sifts(wd, self.id, self.pc, self.component_types, self.selection_plus_residues, self.residue_sifts())

    def write_polar_matching(self, wd):
        """I:405-433: '<id>_polarmatch.csv' and '<id>_specific_polarmatch.csv' for the atoms of selection_plus."""
        export.write_polar_matching(wd, self.id, self.pc, self.component_types, self.selection_plus, self.atom_sifts()['counts'])

    def write_binding_site_sifts(self, wd):
        """I:368-403 cannot complete in the reference itself: its row expression indexes a list with a list (I:401-402,
        a missing ``+``) and ``utils.int3`` needs ``collections.Iterable`` (U:388, gone since Python 3.10), so there is no
        output to reproduce.  ``potential_fsift()`` gives the per-atom potential feature SIFt it would match against."""
        raise NotImplementedError('write_binding_site_sifts raises TypeError in the reference (interactions.py:401-402); '
                                  'see potential_fsift() / atom_sifts() for its inputs')

    def minimize_hydrogens(self, minimisation_forcefield='MMFF94', minimisation_method='ConjugateGradients', minimisation_steps=50):
        """I:214-269 is an OpenBabel force-field run over the hydrogens before ``initialize()`` — SURVEY row 9, outside the
        path this package replaces.  Run it with the reference (or any tool) and hand over the hydrogenated file."""
        raise NotImplementedError('minimize_hydrogens is OpenBabel force-field code (interactions.py:214-269), outside the '
                                  'run_arpeggio path: minimise with OpenBabel first and pass the resulting mmCIF file')

    def write_hydrogenated(self, wd, input_structure):
        """I:271-286 writes OpenBabel's molecule after hydrogen addition; this package adds no hydrogens (see ``read_mmcif``)."""
        raise NotImplementedError('write_hydrogenated writes OpenBabel\'s hydrogenated molecule (interactions.py:271-286); '
                                  'this package does not add hydrogens')

    def potential_fsift(self):
        """``atom.potential_fsift`` (I:1795-1852) as uint8 [n_atoms, 10]."""
        m = export.potential_fsift(self.pc)
        return ((m[:, None] >> np.arange(10, dtype=np.uint16)[None, :]) & 1).astype(np.uint8)

    def get_contacts(self, share_atoms=False):
        """I:172-212: the JSON-able list of contact records, bags in the reference's order (atom-atom, plane-plane,
        atom-plane, group-group, group-plane), canonical order inside a bag ([] before run_arpeggio, like I:84-88).
        ``share_atoms=True`` (not in the reference): records of the same atom share their inner dictionary — same JSON,
        less than half the time on a whole-structure run; see ``export.contacts_json``."""
        return export.contacts_json(self.pc, self._bags, self.component_types, share_atoms)

    def write_json(self, path, indent=4):
        """The file the reference's CLI writes (scripts/process_protein_cli.py:184-188):
        ``json.dump(self.get_contacts(), fh, indent=4, sort_keys=True)``, byte for byte, without building the records in
        Python (native formatter for the atom-atom bag)."""
        export.write_contacts_json(path, self.pc, self._bags, self.component_types, indent)

    # endregion


def residue_plane_sifts(pc, bags):
    """See InteractionComplex.residue_plane_sifts; `bags` = dict of the four ring/amide result bags (arrays)."""
    nr = pc.n_residues
    inter = config.CONTACT_TYPE_NAMES.index('INTER')
    out = {
        'ring_ring_inter_integer_sift': np.zeros((nr, 9), np.int64),
        'ring_atom_inter_integer_sift': np.zeros((nr, 5), np.int64),
        'atom_ring_inter_integer_sift': np.zeros((nr, 5), np.int64),
        'mc_atom_ring_inter_integer_sift': np.zeros((nr, 5), np.int64),
        'sc_atom_ring_inter_integer_sift': np.zeros((nr, 5), np.int64),
        'amide_amide_inter_integer_sift': np.zeros((nr, 1), np.int64),
        'amide_ring_inter_integer_sift': np.zeros((nr, 1), np.int64),
        'ring_amide_inter_integer_sift': np.zeros((nr, 1), np.int64),
    }
    b = bags.get('plane_plane')
    if b is not None and len(b['bgn']):
        r1, r2 = pc.ring_res[b['bgn']], pc.ring_res[b['end']]
        ok = (b['ctype'] == inter) & (r1 != r2)                     # I:1171
        t1, t2 = b['type1'].astype(np.int64), b['type2'].astype(np.int64)
        # the visit that created the record counts for the bgn ring's residue (I:1173-1176) ...
        m = ok & (t1 < 9)
        np.add.at(out['ring_ring_inter_integer_sift'], (r1[m], t1[m]), 1)
        # ... the reverse visit, when it happened, for the end ring's residue with its own class
        rev = np.where(t2 == config.PP_SAME, t1, t2)
        m = ok & (t2 != config.PP_SKIPPED) & (rev < 9)
        np.add.at(out['ring_ring_inter_integer_sift'], (r2[m], rev[m]), 1)
    b = bags.get('atom_plane')
    if b is not None and len(b['atom']):
        ra, rr = pc.res_id[b['atom']], pc.ring_res[b['ring']]
        ok = (b['ctype'] == inter) & (ra != rr)                     # I:1040
        poly = (pc.res_flags[ra] & config.R_POLYPEPTIDE) != 0       # atom.get_parent() in self.polypeptide_residues
        names = np.asarray(pc.atom_name, dtype=object)[b['atom']]
        mc = np.array([n in config.MAINCHAIN_ATOMS for n in names], bool) if len(names) else np.zeros(0, bool)
        for k in range(5):
            m = ok & (((b['mask'] >> k) & 1) == 1)
            np.add.at(out['ring_atom_inter_integer_sift'], (rr[m], k), 1)
            np.add.at(out['atom_ring_inter_integer_sift'], (ra[m], k), 1)
            np.add.at(out['mc_atom_ring_inter_integer_sift'], (ra[m & poly & mc], k), 1)
            np.add.at(out['sc_atom_ring_inter_integer_sift'], (ra[m & poly & ~mc], k), 1)
    b = bags.get('group_group')
    if b is not None and len(b['bgn']):
        r1, r2 = pc.amide_res[b['bgn']], pc.amide_res[b['end']]
        m = (b['ctype'] == inter) & (r1 != r2)                      # I:1290
        np.add.at(out['amide_amide_inter_integer_sift'], (r1[m], 0), 1)
    b = bags.get('group_plane')
    if b is not None and len(b['amide']):
        ra, rr = pc.amide_res[b['amide']], pc.ring_res[b['ring']]
        m = (b['ctype'] == inter) & (ra != rr)                      # I:1371
        np.add.at(out['amide_ring_inter_integer_sift'], (ra[m], 0), 1)
        np.add.at(out['ring_amide_inter_integer_sift'], (rr[m], 0), 1)
    return out


def pack_from_reference_objects(ic, ob=None):
    """Build a PackedComplex from a *reference* ``arpeggio.core.InteractionComplex`` on which ``initialize()`` has run:
    the mapping of I:54-60, 1494-1529, 1591-1733, 1531-1589, 1923-1991 onto the packed arrays.  ``ob``: the OpenBabel
    module (``from openbabel import openbabel``, imported here when not given; the tests pass the iterator holders of
    tests/golden/make_golden_core.py — neither BioPython nor OpenBabel is in this repository's image)."""
    if ob is None:
        from openbabel import openbabel as ob
    atoms = list(ic.s_atoms)
    index = {a: i for i, a in enumerate(atoms)}
    residues, res_index = [], {}
    for a in atoms:
        r = a.get_parent()
        if id(r) not in res_index:
            res_index[id(r)] = len(residues)
            residues.append(r)
    T = config.ATOM_TYPE_BIT
    n = len(atoms)
    xyz = np.array([a.coord for a in atoms], np.float32)
    flags = np.zeros(n, np.uint16)
    tmask = np.zeros(n, np.uint16)
    for i, a in enumerate(atoms):
        tmask[i] = sum(T[t] for t in a.atom_types)
        f = 0
        f |= config.F_METAL if a.is_metal else 0
        f |= config.F_HALOGEN if a.is_halogen else 0
        f |= config.F_WATER if a.get_full_id()[3][0] == 'W' else 0
        f |= config.F_HYDROGEN if a.element.strip() == 'H' else 0
        f |= config.F_ELEM_C if a.element == 'C' else 0
        f |= config.F_ELEM_S if a.element == 'S' else 0
        f |= config.F_RES_MET if a.get_parent().resname == 'MET' else 0
        flags[i] = f
    res_flags = np.zeros(len(residues), np.uint8)
    res_prev = np.full(len(residues), -1, np.int32)
    res_next = np.full(len(residues), -1, np.int32)
    for k, r in enumerate(residues):
        if getattr(r, 'is_polypeptide', False):
            res_flags[k] |= config.R_POLYPEPTIDE
        if hasattr(r, 'prev_residue') and hasattr(r, 'next_residue'):
            res_flags[k] |= config.R_HAS_SEQ
            if r.prev_residue is not None:
                res_prev[k] = res_index[id(r.prev_residue)]
            if r.next_residue is not None:
                res_next[k] = res_index[id(r.next_residue)]
    adj, sb = [[] for _ in range(n)], np.full(n, -1, np.int32)
    for i, a in enumerate(atoms):
        ob_atom = ic.ob_mol.GetAtomById(ic.bio_to_ob[a])
        for nb in ob.OBAtomAtomIter(ob_atom):
            if nb.GetId() in ic.ob_to_bio:
                adj[i].append(index[ic.ob_to_bio[nb.GetId()]])
        for bond in ob.OBAtomBondIter(ob_atom):                           # utils.py:612-635
            if not (bond.GetBondOrder() == 1 and not bond.IsAromatic()):
                continue
            nbr = bond.GetNbrAtom(ob_atom)
            if nbr.GetAtomicNum() == 1:
                continue
            sb[i] = index[ic.ob_to_bio[nbr.GetId()]]
            break
    bond_off = np.concatenate([[0], np.cumsum([len(x) for x in adj])]).astype(np.int32)
    bond_idx = np.array([j for x in adj for j in x], np.int32)
    h_off = np.concatenate([[0], np.cumsum([len(a.h_coords) for a in atoms])]).astype(np.int32)
    h_xyz = np.array([h for a in atoms for h in a.h_coords], np.float64).reshape(-1, 3)
    rings = ic.biopython_str.rings
    amides = ic.biopython_str.amides
    rkeys, akeys = list(rings), list(amides)
    return PackedComplex(
        xyz=xyz, vdw=[a.vdw_radius for a in atoms], cov=[a.cov_radius for a in atoms], type_mask=tmask, flags=flags,
        res_id=[res_index[id(a.get_parent())] for a in atoms], res_flags=res_flags, res_prev=res_prev, res_next=res_next,
        bond_off=bond_off, bond_idx=bond_idx, h_off=h_off, h_xyz=h_xyz, sb_nbr=sb,
        ring_center=np.array([rings[k]['center'] for k in rkeys], np.float64).reshape(-1, 3),
        ring_normal=np.array([rings[k]['normal'] for k in rkeys], np.float64).reshape(-1, 3),
        ring_res=[res_index[id(rings[k]['residue'])] if rings[k].get('residue') is not None else -1 for k in rkeys],
        # (the reference lists a ring's atoms in molecule order, I:1728-1733; the pack holds the ring PATH, which is what a
        # ring normal is computed along: consecutive atoms bonded)
        ring_atoms=[ring_path_order(np.array([index[a] for a in rings[k]['atoms']], np.int32), adj) for k in rkeys],
        amide_center=np.array([amides[k]['center'] for k in akeys], np.float32).reshape(-1, 3),
        amide_normal=np.array([amides[k]['normal'] for k in akeys], np.float32).reshape(-1, 3),
        amide_res=[res_index[id(amides[k]['residue'])] for k in akeys],
        amide_atoms=np.array([[index[a] for a in amides[k]['atoms']] for k in akeys], np.int32).reshape(-1, 4),
        atom_name=[a.name for a in atoms], element=[a.element for a in atoms],
        serial=[a.serial_number for a in atoms], res_name=[r.resname for r in residues],
        res_seq=[r.id[1] for r in residues], res_icode=[r.id[2] for r in residues],
        res_chain=[r.get_parent().id for r in residues], component_types=dict(ic.component_types), id=ic.id)
