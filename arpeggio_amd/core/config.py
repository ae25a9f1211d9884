"""Constants of the hot path.

Numeric thresholds mirror the reference's config (arpeggio/core/config.py:23-35,
590-662); they are compiled into the HIP kernels (csrc/arp_kernels.hip) and listed
here for the host side (names, JSON export, selection parser).
"""

# order of the 12 atom-type names = bit position in PackedComplex.type_mask
# (keys of the reference's ATOM_TYPES, config.py:53-145)
ATOM_TYPE_NAMES = (
    'hbond acceptor', 'hbond donor', 'xbond acceptor', 'xbond donor',
    'weak hbond acceptor', 'weak hbond donor', 'pos ionisable', 'neg ionisable',
    'hydrophobe', 'carbonyl oxygen', 'carbonyl carbon', 'aromatic',
)
ATOM_TYPE_BIT = {name: 1 << i for i, name in enumerate(ATOM_TYPE_NAMES)}

# per-atom flag bits (include/arpeggio_hip.h ARP_F_*)
F_METAL, F_HALOGEN, F_WATER, F_HYDROGEN, F_ELEM_C, F_ELEM_S, F_RES_MET = (1 << i for i in range(7))
# per-residue flag bits (ARP_R_*)
R_POLYPEPTIDE, R_HAS_SEQ = 1, 2

# order of the name list in get_contacts (interactions.py:178-180) = SIFt bit position
SIFT_NAMES = ('clash', 'covalent', 'vdw_clash', 'vdw', 'proximal', 'hbond', 'weak_hbond',
              'xbond', 'ionic', 'metal_complex', 'aromatic', 'hydrophobic', 'carbonyl',
              'polar', 'weak_polar')

# interacting-entities codes (ARP_CT_*)
CONTACT_TYPE_NAMES = ('INTRA_NON_SELECTION', 'INTRA_SELECTION', 'INTER', 'SELECTION_WATER',
                      'NON_SELECTION_WATER', 'WATER_WATER', 'INTRA_BINDING_SITE')

# atom-plane interaction bits (ARP_AP_*); alphabetical order == bit order, which is the
# order sorted(list(potential_interactions)) produces (interactions.py:1061)
ATOM_PLANE_NAMES = ('CARBONPI', 'CATIONPI', 'DONORPI', 'HALOGENPI', 'METSULPHURPI')

# plane-plane classes (ARP_PP_*), interactions.py:1127-1148; index 9 is the '' class
PLANE_PLANE_NAMES = ('FF', 'OF', 'EE', 'FT', 'OT', 'ET', 'FE', 'OE', 'EF', '')
PP_SAME = 254      # the reverse visit gave the same class: nothing appended (interactions.py:1184)
PP_SKIPPED = 255   # no reverse visit (dropped by the intra-residue EE rule, interactions.py:1154)

VDW_RADII = {'H': 1.2}

METALS = set(['LI', 'BE', 'NA', 'MG', 'AL', 'K', 'CA', 'SC', 'TI', 'V', 'CR', 'MN', 'FE', 'CO', 'NI',
              'CU', 'ZN', 'GA', 'RB', 'SR', 'Y', 'ZR', 'NB', 'MO', 'TC', 'RU', 'RH', 'PD', 'AG', 'CD',
              'IN', 'SN', 'CS', 'BA', 'LA', 'CE', 'PR', 'ND', 'PM', 'SM', 'EU', 'GD', 'TB', 'DY', 'HO',
              'ER', 'TM', 'YB', 'LU', 'HF', 'TA', 'W', 'RE', 'OS', 'IR', 'PT', 'AU', 'HG', 'TL', 'PB', 'BI',
              'PO', 'FR', 'RA', 'AC', 'TH', 'PA', 'U', 'NP', 'PU', 'AM', 'CM', 'BK', 'CF'])
HALOGENS = set(['F', 'CL', 'BR', 'I', 'AT'])
MAINCHAIN_ATOMS = set(['N', 'C', 'CA', 'O', 'OXT'])
STANDARD_NUCLEOTIDES = set(['A', 'C', 'G', 'I', 'U', 'DA', 'DC', 'DG', 'DI', 'DT', 'DU', 'N'])

CONTACT_TYPES_DIST_MAX = 4.5
CONTACT_TYPES = {
    "hbond": {"polar distance": 3.5, "angle rad": 1.57},
    "weak hbond": {"weak polar distance": 3.5, "angle rad": 2.27,
                   "cx angle min rad": 0.52, "cx angle max rad": 2.62},
    "aromatic": {"distance": 4.0, "centroid_distance": 6.0, "atom_aromatic_distance": 4.5,
                 "met_sulphur_aromatic_distance": 6.0},
    "amide": {"centroid_distance": 6.0},
    "xbond": {"angle theta 1 rad": 2.09},
    "ionic": {"distance": 4.0},
    "hydrophobic": {"distance": 4.5},
    "carbonyl": {"distance": 3.6},
    "metal": {"distance": 2.8},
}
# literal radii in the reference code: selection expansion (interactions.py:1420)
SELECTION_EXPANSION_RADIUS = 6.0


def _load_common_solvents():
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'common_solvents.txt')
    with open(path) as fh:
        return frozenset(line.strip() for line in fh if line.strip() and not line.startswith('#'))


# residue names excluded by the LIGANDS selector (utils.py:453 of the reference)
COMMON_SOLVENTS = _load_common_solvents()
