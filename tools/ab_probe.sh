#!/bin/bash
# A/B of library builds on ONE box: tools/ab_probe.sh "<lib> <lib> ..." ["pass_probe flags"] [rounds]
# (alternating runs; a lib is a path under the repo, e.g. .ab/libarp_r4.so or arpeggio_amd/csrc/libarpeggio_hip.so)
libs=$1
flags=${2:-}
rounds=${3:-2}
for r in $(seq $rounds); do
  for l in $libs; do
    ARP_LIB_PATH=$PWD/$l timeout 300 python tools/pass_probe.py --tag "$(basename $l)" $flags 2>&1 | tail -1
  done
done
