// arp_host.cpp — libarpeggio_host.so: the host-only entry points of include/arpeggio_hip.h built with g++ (no HIP, no GPU):
// the mmCIF category reader (arp_cif_*) and the JSON writer of the atom-atom records (arp_write_contacts_json).  The same
// sources as in libarpeggio_hip.so (arp_cif.h, arp_cif_api.h, arp_json.h); arpeggio_amd/_capi.py falls back to this library
// for these calls when the HIP library has not been built (a checkout on a machine without hipcc: tests/golden/make_golden*.py).
#include <cstdint>
#include <cstring>

#include "../../include/arpeggio_hip.h"
#include "arp_cif.h"
#include "arp_json.h"

extern "C" const char* arp_host_version(void) { return "arpeggio_host 0.2.0 (host-only subset: arp_cif_*, arp_write_contacts_json)"; }
#include "arp_cif_api.h"      // (C linkage from the prototypes of the public header)
