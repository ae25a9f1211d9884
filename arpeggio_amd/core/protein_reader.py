"""mmCIF -> the atom / residue part of a PackedComplex without gemmi or BioPython.

Mirrors ``arpeggio/core/protein_reader.py`` of the reference where the reference's own code decides, and restates the
third-party behaviour it relies on where it does not:

====================================  ==========================================================================
reference                             here
====================================  ==========================================================================
gemmi ``get_mmcif_category``          ``_capi.CifCategory`` (native CIF tokenizer, ``csrc/arp_cif.h``): '?' -> None,
                                      '.' -> False, quoted values unquoted — recalled gemmi semantics
``_parse_atom_site_biopython``        ``structure_events``: the same calls, in the same order, with the same arguments,
(P:366-411), ``_init_biopython_atom`` as the reference makes on ``StructureBuilder`` — pinned by executing the
(P:309-363), ``_get_hetero_flag``     reference's functions on a recording builder (tests/golden/reader.json)
``Bio.PDB.StructureBuilder``          ``build_structure``: chains by id, residues by (het flag, seq, icode), alternative
                                      locations collapsed to the highest-occupancy child as ``DisorderedAtom`` does —
                                      recalled; point mutations (two residue names under one id) are refused
``get_component_types`` (P:415-441)   ``get_component_types`` — pinned by executing the reference's function
``Bio.PDB.PPBuilder`` (I:1598-1599)   ``build_peptides``: C–N < 1.8 A between consecutive accepted residues of a chain —
                                      recalled; a residue is accepted if it is a standard amino acid or has a CA atom
                                      (BioPython also accepts the modified residues of its extended table that lack CA)
``_handle_chains_residues_and_breaks``  ``polypeptide_bookkeeping`` — pinned by executing the reference's method on the
(I:1591-1693)                         polypeptides ``build_peptides`` returns
====================================  ==========================================================================

What a file alone does NOT give: the OpenBabel bond graph, hydrogens added by OpenBabel, SMARTS types of ligand atoms,
and OpenBabel's perception of aromatic rings and amide groups.  ``read_mmcif`` therefore returns a pack whose atoms of
standard residues are typed by table (``core/typing.py``) and whose other atoms carry no type; the aromatic rings and the
amide groups of STANDARD residues are taken from residue templates (``template_rings`` / ``template_amides``: what
OpenBabel's SSSR + aromaticity and the SMARTS of C:37 find in a protein — recalled, the order of OpenBabel's lists is not
pinned); ``pc.incomplete`` lists what is missing, and ``InteractionComplex(path)`` refuses a run that needs a missing part
(``IncompleteStructureError``) unless it is told to go on.  The reference also passes the file through gemmi's ``read_structure`` +
``make_mmcif_block`` (first model only, merged chain parts, coordinates re-formatted to three decimals) before reading
the table; this reader takes the first model of the file's own ``_atom_site`` table.
"""
from __future__ import annotations

import json
import os

import numpy as np

from . import config, typing
from .packed import PackedComplex

_HERE = os.path.dirname(os.path.abspath(__file__))
_COMP = None


def _read_text(path):
    if not os.path.isfile(path):
        raise IOError('File {} not found'.format(path))            # P:58-59, 269-270, 417-418
    with open(path, 'rb') as fh:
        return fh.read()


def _category(text, name):
    from .. import _capi
    return _capi.CifCategory(text, name)


# ---------------------------------------------------------------------------------------------------------------------
# _chem_comp -> component types
# ---------------------------------------------------------------------------------------------------------------------
def _comp_table():
    global _COMP
    if _COMP is None:
        _COMP = json.load(open(os.path.join(_HERE, 'data', 'chem_comp_types.json')))
    return _COMP


def component_types_from_columns(chem_comp):
    """P:421-437 on the columns of the ``_chem_comp`` category (item -> list, gemmi conventions)."""
    t = _comp_table()
    letter = {v: k for k, v in t['letters'].items()}
    out = {}
    for i in range(len(chem_comp['id'])):
        ctype = chem_comp['type'][i]
        if ctype:
            comp = t['chem_comp_type'][ctype.upper()]               # KeyError for a type the table does not hold, as the reference
            if comp != letter[5]:
                out[chem_comp['id'][i]] = comp
            elif chem_comp['name'][i].upper() == 'WATER':
                out[chem_comp['id'][i]] = letter[8]
            else:
                out[chem_comp['id'][i]] = letter[7]
        else:
            out[chem_comp['id'][i]] = letter[9]
    return out


def get_component_types(path):
    """``protein_reader.get_component_types`` (P:415-441): residue name -> one-letter component type."""
    text = _read_text(path)
    cat = _category(text, '_chem_comp.')
    if cat.n_blocks != 1:
        raise RuntimeError('single data block expected, got: %d' % cat.n_blocks)      # gemmi's sole_block()
    if cat.rows == 0 and not cat.tags:
        raise ValueError('Missing _chem_comp. category in mmcif')                      # P:440-441
    return component_types_from_columns(cat.columns())


# ---------------------------------------------------------------------------------------------------------------------
# _atom_site -> the calls the reference makes on StructureBuilder
# ---------------------------------------------------------------------------------------------------------------------
def _get_hetero_flag(field, resn):
    """P:292-306."""
    if field == 'HETATM':
        if resn in ('HOH', 'WAT'):
            return 'W'
        return 'H'
    return ' '


def structure_events(atom_sites):
    """The builder calls of ``_parse_atom_site_biopython`` (P:366-411) for an ``_atom_site`` dict of columns, as a list of
    tuples ('model', n) / ('seg', id) / ('chain', id) / ('residue', name, het flag, seq, icode) / ('atom', name, (x, y,
    z), b, occupancy, altloc, fullname, serial, element).  The reference's quirks are kept: a chain change alone does not
    open a residue (``last_chain_id`` is updated before it is compared, P:390-399)."""
    ev = []
    has_label = 'pdbe_label_seq_id' in atom_sites
    has_b = 'B_iso_or_equiv' in atom_sites
    last_model = last_ins = last_chain = last_name = last_id = None
    n = len(atom_sites['id'])
    for i in range(n):
        res_id = int(atom_sites['pdbe_label_seq_id'][i] if has_label else atom_sites['auth_seq_id'][i])       # P:16-21
        name = atom_sites['label_comp_id'][i]
        het = _get_hetero_flag(atom_sites['group_PDB'][i], name)
        ins = ' ' if not atom_sites['pdbx_PDB_ins_code'][i] else atom_sites['pdbx_PDB_ins_code'][i]          # P:30-35
        if last_model != atom_sites['pdbx_PDB_model_num'][i]:
            last_model = atom_sites['pdbx_PDB_model_num'][i]
            ev.append(('model', int(last_model) - 1))
        if i == 0:
            ev.append(('seg', '    '))
        if last_chain != atom_sites['auth_asym_id'][i]:
            last_chain = atom_sites['auth_asym_id'][i]
            ev.append(('chain', last_chain))
        if last_id != res_id or last_name != name or last_ins != ins:
            last_id, last_name, last_ins = res_id, name, ins
            ev.append(('residue', name, het, res_id, ins))
        x, y, z = float(atom_sites['Cartn_x'][i]), float(atom_sites['Cartn_y'][i]), float(atom_sites['Cartn_z'][i])
        b = float(atom_sites['B_iso_or_equiv'][i]) if has_b else 20.0                                         # P:24-27
        alt = ' ' if not atom_sites['label_alt_id'][i] else atom_sites['label_alt_id'][i]
        ev.append(('atom', atom_sites['label_atom_id'][i], (x, y, z), b, float(atom_sites['occupancy'][i]), alt,
                   atom_sites['label_atom_id'][i], int(atom_sites['id'][i]), atom_sites['type_symbol'][i].upper()))
    return ev


# ---------------------------------------------------------------------------------------------------------------------
# StructureBuilder (BioPython; recalled): chains, residues, disordered atoms
# ---------------------------------------------------------------------------------------------------------------------
class _Atom:
    __slots__ = ('name', 'children', 'selected', 'last_occ', 'residue')

    def __init__(self, name, residue):
        self.name, self.children, self.selected, self.last_occ, self.residue = name, [], 0, -1e300, residue

    def add(self, coord, b, occ, alt, serial, element):
        self.children.append(dict(coord=np.array(coord, 'f'), b=b, occ=occ, alt=alt, serial=serial, element=element))
        if occ > self.last_occ:              # DisorderedAtom.disordered_add: the child with the highest occupancy is selected
            self.last_occ, self.selected = occ, len(self.children) - 1

    def __getattr__(self, k):
        return self.children[self.selected][k]


class _Residue:
    def __init__(self, name, het, seq, icode, chain):
        self.name, self.het, self.seq, self.icode, self.chain = name, het, seq, icode, chain
        self.atoms, self.by_name = [], {}


def build_structure(events, first_model_only=True):
    """Chains (by id, in order of first appearance), residues and atoms as ``Bio.PDB.StructureBuilder`` would arrange them
    for the events of ``structure_events`` (first model).  Returns the list of chains, each a list of residues."""
    chains, by_id = [], {}
    chain = residue = None
    models = 0
    for e in events:
        kind = e[0]
        if kind == 'model':
            models += 1
            if models > 1 and first_model_only:          # the reference drops the other models before it reads the table
                break
        elif kind == 'chain':
            if e[1] not in by_id:
                by_id[e[1]] = (e[1], [], {})
                chains.append(by_id[e[1]])
            chain = by_id[e[1]]                          # (a chain seen before: BioPython warns "discontinuous" and goes on)
        elif kind == 'residue':
            _, name, het, seq, icode = e
            key = (het if het != 'H' else 'H_' + name, seq, icode)
            old = chain[2].get(key)
            if old is not None:
                if old.name != name or het != ' ':
                    raise ValueError(f'residue id {key} of chain {chain[0]} is used twice (point mutation / duplicated hetero '
                                     f'residue): not supported by this reader')
                residue = old                            # "Residue redefined": BioPython continues with the existing one
            else:
                residue = _Residue(name, het, seq, icode, chain[0])
                chain[1].append(residue)
                chain[2][key] = residue
        elif kind == 'atom':
            _, name, coord, b, occ, alt, fullname, serial, element = e
            a = residue.by_name.get(name)
            if a is None:
                a = _Atom(name, residue)
                residue.by_name[name] = a
                residue.atoms.append(a)
            elif alt == ' ' or any(c['alt'] == alt for c in a.children):
                raise ValueError(f'atom {name} defined twice in residue {residue.name} {residue.seq} of chain {chain[0]}: '
                                 f'not supported by this reader')
            a.add(coord, b, occ, alt, serial, element)
    return [(cid, res) for cid, res, _ in chains]


# ---------------------------------------------------------------------------------------------------------------------
# PPBuilder (BioPython; recalled) and the reference's bookkeeping on top of it
# ---------------------------------------------------------------------------------------------------------------------
def _accept(res):
    return res.name in typing.table()['std_res'] or 'CA' in res.by_name


def _is_connected(prev, nxt, radius=1.8):
    c, n = prev.by_name.get('C'), nxt.by_name.get('N')
    if c is None or n is None:
        return False
    for ni, nn in enumerate(n.children):                 # every pair of compatible alternative locations; the pair that is
        for ci, cc in enumerate(c.children):             # bonded becomes the selected one (PPBuilder._is_connected)
            if nn['alt'] == cc['alt'] or nn['alt'] == ' ' or cc['alt'] == ' ':
                d = nn['coord'] - cc['coord']
                if np.sqrt(np.dot(d, d)) < radius:       # Atom.__sub__: float32 difference, np.sqrt(np.dot(diff, diff))
                    if len(c.children) > 1:
                        c.selected = ci
                    if len(n.children) > 1:
                        n.selected = ni
                    return True
    return False


def build_peptides(chains):
    """``PPBuilder().build_peptides(structure, aa_only=False)``: lists of consecutive connected residues."""
    pps = []
    for _, residues in chains:
        it = iter(residues)
        prev = next((r for r in it if _accept(r)), None)
        if prev is None:
            continue
        pp = None
        for nxt in it:
            if _accept(prev) and _accept(nxt) and _is_connected(prev, nxt):
                if pp is None:
                    pp = [prev]
                    pps.append(pp)
                pp.append(nxt)
            else:
                pp = None
            prev = nxt
    return pps


def polypeptide_bookkeeping(polypeptides):
    """I:1656-1693: residues of a polypeptide are flagged and linked to their neighbours inside it.  Returns
    {id(residue): (prev residue or None, next residue or None)} for the residues that are part of a polypeptide."""
    links = {}
    for pp in polypeptides:
        last = None
        for r in pp:
            links[id(r)] = [last, None]
            if last is not None:
                links[id(last)][1] = r
            last = r
    return links


# ---------------------------------------------------------------------------------------------------------------------
# what gemmi does to the table before the reference reads it (recalled: gemmi is not in the image)
# ---------------------------------------------------------------------------------------------------------------------
def gemmi_normalised(atom_sites):
    """The reference reads ``gemmi.read_structure(path, merge_chain_parts=True)`` -> first model -> ``make_mmcif_block``
    (P:61-68, 272-277), not the file's own table.  Restated here on the columns: (1) rows of the first model only; (2) the
    parts of a chain (polymer, ligands, waters: same auth_asym_id, apart in the file) follow each other — chains in the
    order of their first row, rows of a chain in file order; (3) coordinates re-written with three decimals
    (``%.3f``, then parsed again)."""
    n = len(atom_sites['id'])
    models = atom_sites['pdbx_PDB_model_num']
    keep = [i for i in range(n) if models[i] == models[0]]
    first = {}
    for i in keep:
        first.setdefault(atom_sites['auth_asym_id'][i], len(first))
    order = sorted(keep, key=lambda i: (first[atom_sites['auth_asym_id'][i]], i))
    out = {k: [v[i] for i in order] for k, v in atom_sites.items()}
    for k in ('Cartn_x', 'Cartn_y', 'Cartn_z'):
        out[k] = ['%.3f' % float(v) for v in out[k]]
    return out


# ---------------------------------------------------------------------------------------------------------------------
# _struct_conn -> bonds (P:139-212)
# ---------------------------------------------------------------------------------------------------------------------
def struct_conn_pairs(atom_sites, struct_conn):
    """``parse_struct_conn_bonds`` / ``__process_struct_conn`` (P:139-194): for every row of ``_struct_conn`` — whatever its
    conn_type_id — the ``_atom_site.id`` of the FIRST row that has partner 1's (auth_asym_id, auth_seq_id, label_atom_id)
    and the same for partner 2 (strings compared as they stand; alternative locations and insertion codes are not
    looked at); rows with a partner that is not found are dropped.  Returns the list of (id_a, id_b)."""
    lookup = {}
    for i in range(len(atom_sites['id'])):
        lookup.setdefault((atom_sites['auth_asym_id'][i], atom_sites['auth_seq_id'][i], atom_sites['label_atom_id'][i]), int(atom_sites['id'][i]))
    pairs = []
    pivot = next(iter(struct_conn))
    for k in range(len(struct_conn[pivot])):
        a = lookup.get((struct_conn['ptnr1_auth_asym_id'][k], struct_conn['ptnr1_auth_seq_id'][k], struct_conn['ptnr1_label_atom_id'][k]), 0)
        b = lookup.get((struct_conn['ptnr2_auth_asym_id'][k], struct_conn['ptnr2_auth_seq_id'][k], struct_conn['ptnr2_label_atom_id'][k]), 0)
        if a != 0 and b != 0:
            pairs.append((a, b))
    return pairs


def add_bonds(neighbours, pairs):
    """``__add_bond_to_openbabel`` (P:197-212) on adjacency lists {atom id: [neighbour ids]}: a bond that exists already
    (in either direction: OBAtomAtomIter walks both) is not added again; a new one is appended to both atoms' lists."""
    for a, b in pairs:
        if b in neighbours.setdefault(a, []):
            continue
        neighbours[a].append(b)
        neighbours.setdefault(b, []).append(a)
    return neighbours


# ---------------------------------------------------------------------------------------------------------------------
# explicit hydrogens of a hydrogenated file -> h_coords of their heavy atom (I:1513-1529)
# ---------------------------------------------------------------------------------------------------------------------
def attach_hydrogens(xyz64, is_h, res_id, max_dist=1.3):
    """The reference finds the hydrogens of an atom through OpenBabel's bond graph (ConnectTheDots, I:1521-1529).  Without
    OpenBabel: a hydrogen belongs to the nearest non-hydrogen atom of its own residue within ``max_dist`` A (any residue
    when its own has none) — for a file hydrogenated at standard bond lengths (X-H <= 1.1 A) the same parent.
    Returns parent index per atom (-1: not a hydrogen / no parent)."""
    n = len(xyz64)
    parent = np.full(n, -1, np.int64)
    heavy = np.nonzero(~is_h)[0]
    if not len(heavy) or not is_h.any():
        return parent
    from scipy.spatial import cKDTree
    tree = cKDTree(xyz64[heavy])
    for h in np.nonzero(is_h)[0]:
        near = tree.query_ball_point(xyz64[h], max_dist)
        if not near:
            continue
        cand = heavy[np.array(near)]
        d = np.linalg.norm(xyz64[cand] - xyz64[h], axis=1)
        same = res_id[cand] == res_id[h]
        pick = np.where(same)[0] if same.any() else np.arange(len(cand))
        parent[h] = cand[pick[np.argmin(d[pick])]]
    return parent


# ---------------------------------------------------------------------------------------------------------------------
# file -> PackedComplex
# ---------------------------------------------------------------------------------------------------------------------
# Aromatic rings of the standard residues, each as the ring PATH (consecutive atoms bonded), and the side-chain amide groups
# (N, C, O, the carbon on C): what OpenBabel's ring perception (GetSSSR + IsAromatic, I:1697-1733) and the SMARTS
# '[NX3][CX3](=[OX1])[#6]' (C:37, I:1531-1589) give for a protein with explicit hydrogens.  Recalled, not pinned: OpenBabel is
# not in the image; in particular the ORDER of its ring list (ring ids) and whether it calls a given histidine aromatic.
RING_TEMPLATES = {
    'PHE': (('CG', 'CD1', 'CE1', 'CZ', 'CE2', 'CD2'),),
    'TYR': (('CG', 'CD1', 'CE1', 'CZ', 'CE2', 'CD2'),),
    'TRP': (('CG', 'CD1', 'NE1', 'CE2', 'CD2'), ('CD2', 'CE2', 'CZ2', 'CH2', 'CZ3', 'CE3')),
    'HIS': (('CG', 'ND1', 'CE1', 'NE2', 'CD2'),),
}
SIDE_CHAIN_AMIDES = {'ASN': ('ND2', 'CG', 'OD1', 'CB'), 'GLN': ('NE2', 'CD', 'OE1', 'CG')}


# Bonds between the heavy atoms of the standard residues (wwPDB chemical component definitions: atom-name pairs, bond order, and
# whether OpenBabel flags the bond aromatic, i.e. both atoms in one of the RING_TEMPLATES rings), in place of OpenBabel's residue
# perception / ConnectTheDots (P:124-131 hands the file to OpenBabel; I:748-757 and U:612-635 walk the result).  What arpeggio
# reads off these bonds: the single-bond heavy neighbour of an atom (U:612-635: first bond of order 1 that is not aromatic) and
# the parents of hydrogens.  The ORDER in which OpenBabel lists an atom's bonds is its own (recalled as: the order of creation,
# residue perception first) — no halogen sits in a standard residue, so no result of the path depends on it.
_BACKBONE = (('N', 'CA', 1), ('CA', 'C', 1), ('C', 'O', 2), ('C', 'OXT', 1))
_SIDE = {
    'ALA': (('CA', 'CB', 1),),
    'ARG': (('CA', 'CB', 1), ('CB', 'CG', 1), ('CG', 'CD', 1), ('CD', 'NE', 1), ('NE', 'CZ', 1), ('CZ', 'NH1', 1), ('CZ', 'NH2', 2)),
    'ASN': (('CA', 'CB', 1), ('CB', 'CG', 1), ('CG', 'OD1', 2), ('CG', 'ND2', 1)),
    'ASP': (('CA', 'CB', 1), ('CB', 'CG', 1), ('CG', 'OD1', 2), ('CG', 'OD2', 1)),
    'CYS': (('CA', 'CB', 1), ('CB', 'SG', 1)),
    'GLN': (('CA', 'CB', 1), ('CB', 'CG', 1), ('CG', 'CD', 1), ('CD', 'OE1', 2), ('CD', 'NE2', 1)),
    'GLU': (('CA', 'CB', 1), ('CB', 'CG', 1), ('CG', 'CD', 1), ('CD', 'OE1', 2), ('CD', 'OE2', 1)),
    'GLY': (),
    'HIS': (('CA', 'CB', 1), ('CB', 'CG', 1), ('CG', 'ND1', 1), ('CG', 'CD2', 2), ('ND1', 'CE1', 2), ('CD2', 'NE2', 1), ('CE1', 'NE2', 1)),
    'ILE': (('CA', 'CB', 1), ('CB', 'CG1', 1), ('CB', 'CG2', 1), ('CG1', 'CD1', 1)),
    'LEU': (('CA', 'CB', 1), ('CB', 'CG', 1), ('CG', 'CD1', 1), ('CG', 'CD2', 1)),
    'LYS': (('CA', 'CB', 1), ('CB', 'CG', 1), ('CG', 'CD', 1), ('CD', 'CE', 1), ('CE', 'NZ', 1)),
    'MET': (('CA', 'CB', 1), ('CB', 'CG', 1), ('CG', 'SD', 1), ('SD', 'CE', 1)),
    'MSE': (('CA', 'CB', 1), ('CB', 'CG', 1), ('CG', 'SE', 1), ('SE', 'CE', 1)),
    'PHE': (('CA', 'CB', 1), ('CB', 'CG', 1), ('CG', 'CD1', 2), ('CG', 'CD2', 1), ('CD1', 'CE1', 1), ('CD2', 'CE2', 2), ('CE1', 'CZ', 2), ('CE2', 'CZ', 1)),
    'PRO': (('CA', 'CB', 1), ('CB', 'CG', 1), ('CG', 'CD', 1), ('CD', 'N', 1)),
    'SER': (('CA', 'CB', 1), ('CB', 'OG', 1)),
    'THR': (('CA', 'CB', 1), ('CB', 'OG1', 1), ('CB', 'CG2', 1)),
    'TRP': (('CA', 'CB', 1), ('CB', 'CG', 1), ('CG', 'CD1', 2), ('CG', 'CD2', 1), ('CD1', 'NE1', 1), ('NE1', 'CE2', 1), ('CD2', 'CE2', 2),
            ('CD2', 'CE3', 1), ('CE2', 'CZ2', 1), ('CE3', 'CZ3', 2), ('CZ2', 'CH2', 2), ('CZ3', 'CH2', 1)),
    'TYR': (('CA', 'CB', 1), ('CB', 'CG', 1), ('CG', 'CD1', 2), ('CG', 'CD2', 1), ('CD1', 'CE1', 1), ('CD2', 'CE2', 2), ('CE1', 'CZ', 2), ('CE2', 'CZ', 1),
            ('CZ', 'OH', 1)),
    'VAL': (('CA', 'CB', 1), ('CB', 'CG1', 1), ('CB', 'CG2', 1)),
}
RESIDUE_BONDS = {k: _BACKBONE + v for k, v in _SIDE.items()}


def template_bonds(residues, atom_index):
    """[(atom, atom, order, aromatic)] of the heavy atoms of the standard residues that are present, residue by residue in
    template order (packed indices).  A bond is aromatic when both atoms lie in one aromatic ring of RING_TEMPLATES."""
    out = []
    for r in residues:
        name = r.name.strip()
        rings = [set(t) for t in RING_TEMPLATES.get(name, ())]
        for a, b, order in RESIDUE_BONDS.get(name, ()):
            pa, pb = r.by_name.get(a), r.by_name.get(b)
            if pa is None or pb is None:
                continue
            arom = int(any(a in t and b in t for t in rings))
            out.append((atom_index[id(pa)], atom_index[id(pb)], order, arom))
    return out


def hydrogen_parent_by_name(hname, residue, coord=None):
    """The heavy atom a hydrogen of a standard residue is named after (wwPDB nomenclature: H + the parent's name without its
    element letter + a counter — HA on CA, HB2 on CB, HG21 on CG2, HD1 on CD1 or ND1, HH on OH, H / H1..H3 / HN on N): the heavy
    atom of the residue whose name, less its first letter, is the longest NON-EMPTY prefix of the hydrogen's name less its 'H'.
    Ties (MSE: SE and CE both reduce to 'E'; HD1 of HIS: CD1 absent, ND1 present — no tie there) go to the candidate that has a
    template bond to a carbon-like parent first (a hydrogen named after a selenium / sulphur that has no hydrogens in the
    template is a methyl hydrogen of the carbon), then to the nearest in space when ``coord`` is given.  The backbone N only
    takes the names H, H1, H2, H3, HN, D: anything else that matches nothing stays unattached (None)."""
    name = hname.strip()
    if name in ('H', 'H1', 'H2', 'H3', 'HN', 'D'):
        for a in residue.atoms:
            if a.name.strip() == 'N':
                return a
        return None
    rem = name[1:]
    cands, best_len = [], 0
    for a in residue.atoms:
        if a.element in ('H', 'D'):
            continue
        suf = a.name.strip()[1:]
        if not suf or not rem.startswith(suf):
            continue
        if len(suf) > best_len:
            cands, best_len = [a], len(suf)
        elif len(suf) == best_len:
            cands.append(a)
    if not cands:
        return None
    if len(cands) == 1:
        return cands[0]
    # heteroatoms that carry no hydrogen in any standard residue / MSE template lose a tie against a carbon or nitrogen
    pref = [a for a in cands if a.element not in ('SE', 'S')] or cands
    if len(pref) > 1 and coord is not None:
        c0 = np.asarray(coord, np.float64)
        pref.sort(key=lambda a: float(np.sum((np.asarray(a.coord, np.float64) - c0) ** 2)))
    return pref[0]


def template_rings(residues, atom_index):
    """Ring atom lists (packed indices, ring-path order) of the standard residues that have all atoms of a template, in the
    order of their first atom in the file."""
    out = []
    for r in residues:
        for names in RING_TEMPLATES.get(r.name.strip(), ()):
            atoms = [r.by_name.get(nm) for nm in names]
            if all(a is not None for a in atoms):
                out.append(np.array([atom_index[id(a)] for a in atoms], np.int32))
    out.sort(key=lambda a: int(a.min()))
    return out


def template_amides(residues, res_next, atom_index):
    """Amide groups [N, C, O, carbon on C] of a protein: one per peptide bond (N of the next residue of the polypeptide; C, O,
    CA of this one) and the side chains of ASN / GLN; ordered by the nitrogen's position in the file (the SMARTS matcher walks
    the atoms in order).  A proline nitrogen (three connections) matches like any other."""
    out = []
    for k, r in enumerate(residues):
        if res_next[k] >= 0:
            n_ = residues[res_next[k]].by_name.get('N')
            grp = [n_, r.by_name.get('C'), r.by_name.get('O'), r.by_name.get('CA')]
            if all(a is not None for a in grp):
                out.append([atom_index[id(a)] for a in grp])
        names = SIDE_CHAIN_AMIDES.get(r.name.strip())
        if names:
            grp = [r.by_name.get(nm) for nm in names]
            if all(a is not None for a in grp):
                out.append([atom_index[id(a)] for a in grp])
    out.sort(key=lambda g: g[0])
    return np.array(out, np.int32).reshape(-1, 4)


def read_mmcif(path, use_ambiguities=False, normalise=True):
    """``initialize()``'s result for an mmCIF file as far as the file alone goes (see the module docstring and
    ``pc.incomplete``): atoms, residues, polypeptide links, table types and radii; bonds from ``_struct_conn``, the peptide
    bonds of the polypeptides and the X-H bonds of explicit hydrogens; the hydrogens' coordinates on their heavy atoms.
    ``normalise``: apply what gemmi does to the table first (``gemmi_normalised``)."""
    text = _read_text(path)
    cat = _category(text, '_atom_site.')
    if cat.rows == 0:
        raise ValueError('The cif file does not contain _atom_site record')           # P:285-287
    cols = cat.columns()
    if normalise:
        cols = gemmi_normalised(cols)
    chains = build_structure(structure_events(cols))
    links = polypeptide_bookkeeping(build_peptides(chains))
    residues = [r for _, rs in chains for r in rs]
    res_index = {id(r): k for k, r in enumerate(residues)}
    atoms = [a for r in residues for a in r.atoms]
    n, nres = len(atoms), len(residues)
    res_flags = np.zeros(nres, np.uint8)
    res_prev, res_next = np.full(nres, -1, np.int32), np.full(nres, -1, np.int32)
    for k, r in enumerate(residues):
        if id(r) in links:
            res_flags[k] |= config.R_POLYPEPTIDE | config.R_HAS_SEQ
            p, q = links[id(r)]
            res_prev[k] = -1 if p is None else res_index[id(p)]
            res_next[k] = -1 if q is None else res_index[id(q)]
    element = [a.element for a in atoms]
    res_id = np.array([res_index[id(a.residue)] for a in atoms], np.int32)
    res_name = [r.name for r in residues]
    flags = typing.element_flags(element, res_name, res_id)
    water = np.array([r.het == 'W' for r in residues], bool)[res_id] if n else np.zeros(0, bool)
    flags[water] |= config.F_WATER
    vdw, cov = typing.element_radii(element)
    try:
        comp = component_types_from_columns(_category(text, '_chem_comp.').columns())
    except (KeyError, ValueError):
        comp = {}
    # ---- bonds the file itself gives: _struct_conn (P:124-131), the C-N bonds of the polypeptides, X-H of explicit hydrogens
    by_serial = {int(a.serial): k for k, a in enumerate(atoms)}
    # OpenBabel's copy of an atom holds the coordinate TEXT as float64 (P:231-235); BioPython's holds float32
    xyz64 = np.zeros((n, 3))
    text_xyz = {int(cols['id'][i]): (float(cols['Cartn_x'][i]), float(cols['Cartn_y'][i]), float(cols['Cartn_z'][i])) for i in range(len(cols['id']))}
    for k, a in enumerate(atoms):
        xyz64[k] = text_xyz[int(a.serial)]
    neighbours = {}
    atom_index = {id(a): k for k, a in enumerate(atoms)}
    # bonds inside the standard residues, from the residue templates (OpenBabel perceives them when it reads the file, before
    # arpeggio adds the _struct_conn ones, P:124-131)
    tbonds = template_bonds(residues, atom_index)
    bond_kind = {}                                                    # (a, b) -> (order, aromatic); anything else: a single bond
    for a_, b_, o_, ar_ in tbonds:
        bond_kind[(a_, b_)] = bond_kind[(b_, a_)] = (o_, ar_)
    add_bonds(neighbours, [(a_, b_) for a_, b_, _, _ in tbonds])
    try:
        sc = _category(text, '_struct_conn.')
        if sc.rows:
            pairs = [(by_serial[a], by_serial[b]) for a, b in struct_conn_pairs(cols, sc.columns()) if a in by_serial and b in by_serial]
            add_bonds(neighbours, pairs)
    except KeyError:
        pass
    pep = []                                                          # peptide bonds: C of a residue - N of its successor in the polypeptide
    for k, r in enumerate(residues):
        if res_next[k] >= 0:
            c_, n_ = r.by_name.get('C'), residues[res_next[k]].by_name.get('N')
            if c_ is not None and n_ is not None:
                pep.append((atom_index[id(c_)], atom_index[id(n_)]))
    add_bonds(neighbours, pep)
    is_h = np.array([e in ('H', 'D') for e in element], bool) if n else np.zeros(0, bool)
    parent = attach_hydrogens(xyz64, is_h, res_id)
    for h_ in np.nonzero(is_h & (parent < 0))[0] if n else ():        # a hydrogen of a standard residue too far from everything: by its name
        r_ = atoms[h_].residue
        if r_.name.strip() in RESIDUE_BONDS:
            p_ = hydrogen_parent_by_name(atoms[h_].name, r_, atoms[h_].coord)
            if p_ is not None:
                parent[h_] = atom_index[id(p_)]
    add_bonds(neighbours, [(int(p_), int(h_)) for h_, p_ in enumerate(parent) if p_ >= 0])
    bond_off = np.zeros(n + 1, np.int32)
    for k in range(n):
        bond_off[k + 1] = bond_off[k] + len(neighbours.get(k, ()))
    bond_idx = np.array([b for k in range(n) for b in neighbours.get(k, ())], np.int32)
    h_lists = [[] for _ in range(n)]
    for h_, p_ in enumerate(parent):
        if p_ >= 0:
            h_lists[p_].append(h_)
    h_off = np.zeros(n + 1, np.int32)
    for k in range(n):
        h_off[k + 1] = h_off[k] + len(h_lists[k])
    h_xyz = np.array([xyz64[h_] for k in range(n) for h_ in h_lists[k]], np.float64).reshape(-1, 3)
    from .packed import single_bond_neighbours
    kinds = [bond_kind.get((k, int(b)), (1, 0)) for k in range(n) for b in neighbours.get(k, ())]
    bond_order = np.array([o_ for o_, _ in kinds], np.int32)
    bond_aromatic = np.array([ar_ for _, ar_ in kinds], np.int32)
    sb_nbr = single_bond_neighbours(bond_off, bond_idx, bond_order, bond_aromatic, is_h)
    ring_atoms = template_rings(residues, atom_index)
    amide_atoms = template_amides(residues, res_next, atom_index)
    pc = PackedComplex(
        xyz=np.array([a.coord for a in atoms], np.float32).reshape(-1, 3), vdw=vdw, cov=cov, type_mask=np.zeros(n, np.uint16), flags=flags,
        res_id=res_id, res_flags=res_flags, res_prev=res_prev, res_next=res_next, bond_off=bond_off,
        bond_idx=bond_idx, h_off=h_off, h_xyz=h_xyz, sb_nbr=sb_nbr,
        # (centres / normals / residues of the template rings and amides: InteractionComplex.compute_plane_geometry, on the GPU)
        ring_center=np.zeros((len(ring_atoms), 3)), ring_normal=np.zeros((len(ring_atoms), 3)), ring_res=np.full(len(ring_atoms), -1, np.int32),
        ring_atoms=ring_atoms,
        amide_center=np.zeros((len(amide_atoms), 3), np.float32), amide_normal=np.zeros((len(amide_atoms), 3), np.float32),
        amide_res=np.full(len(amide_atoms), -1, np.int32), amide_atoms=amide_atoms,
        id=os.path.basename(path).split('.')[0])
    pc.atom_name = [a.name for a in atoms]
    pc.element = element
    pc.serial = np.array([a.serial for a in atoms], np.int64)
    pc.res_name = res_name
    pc.res_chain = [r.chain for r in residues]
    pc.res_seq = np.array([r.seq for r in residues], np.int32)
    pc.res_icode = [r.icode for r in residues]
    pc.res_het = [r.het for r in residues]
    pc.component_types = comp
    pc.type_mask = typing.apply_protein_typing(pc, use_ambiguities=use_ambiguities)
    # what only OpenBabel can add: bonds inside NON-STANDARD residues (ConnectTheDots: they matter for the single-bond
    # neighbour of halogens and for hydrogens further than 1.3 A from any atom — pairs inside a residue are never contacts,
    # I:729), hydrogens of a file that has none (AddHydrogens), SMARTS types of non-standard residues, rings, amides
    # ... and the element radii: OpenBabel's table restated from memory in core/typing.py, not verified against an OpenBabel build
    # (bonds inside the STANDARD residues come from the residue templates above)
    pc.incomplete = ('bonds inside non-standard residues', 'added hydrogens', 'ligand atom types', 'rings of non-standard residues',
                     'amides of non-standard residues', 'ring / amide ids not in OpenBabel\'s order', 'element radii unverified')
    pc.bond_order, pc.bond_aromatic = bond_order, bond_aromatic
    pc.plane_geometry_pending = len(ring_atoms) + len(amide_atoms) > 0      # centres / normals not computed yet
    pc.hydrogen_parent = parent
    # atoms whose types only OpenBabel's SMARTS could give: heavy atoms of residues outside the typing dictionary that are not water
    std = set(typing.table()['std_res'])
    res_untyped = np.array([(rn not in std) and not w_ for rn, w_ in zip(res_name, [r.het == 'W' for r in residues])], bool)
    pc.untyped_atoms = (res_untyped[res_id] & ~is_h) if n else np.zeros(0, bool)
    import logging
    orphans = int((is_h & (parent < 0)).sum()) if n else 0
    logging.info('read_mmcif(%s): %d atoms, %d residues; bonds inferred: %d peptide, %d X-H; %d hydrogens without a heavy atom within 1.3 A; '
                 '%d rings and %d amide groups from residue templates; %d heavy atoms of %d non-standard residues carry no atom type',
                 os.path.basename(path), n, nres, len(pep), int((parent >= 0).sum()) if n else 0, orphans, len(ring_atoms), len(amide_atoms),
                 int(pc.untyped_atoms.sum()), int(res_untyped.sum()))
    if orphans:
        logging.warning('read_mmcif(%s): %d explicit hydrogens have no heavy atom of their residue within 1.3 A and take no part in the '
                        'hydrogen-bond geometry (OpenBabel would attach them by its own bond perception)', os.path.basename(path), orphans)
    return pc
