#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG=..."  ->  .ab/libarp_NAME.so (developer A/B builds; ARP_LIB_PATH selects one)
set -e
name=$1; flags=${2:-}
mkdir -p .ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wall -Wno-unused-function -Wno-unused-variable $flags arpeggio_amd/csrc/arp_api.hip -o .ab/libarp_$name.so 2>&1 | grep -E " error|occupancy" | head -5
ls -la .ab/libarp_$name.so
