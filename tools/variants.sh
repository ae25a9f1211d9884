#!/bin/bash
# tools/variants.sh OUT "FLAGS1" "FLAGS2" ...   — rebuild the library with each set of extra hipcc flags and run pass_probe (GPU box)
out=$1; shift
mkdir -p $(dirname $out)
: > $out
for flags in "$@"; do
  ARP_EXTRA_HIPCC_FLAGS="$flags" python -c "from arpeggio_amd import build; build.build(force=True)" > /dev/null 2>> $out.err || { echo "{\"tag\": \"$flags\", \"error\": \"build\"}" >> $out; continue; }
  timeout 300 python tools/pass_probe.py --tag="$flags" $PROBE_ARGS >> $out 2>> $out.err || echo "{\"tag\": \"$flags\", \"error\": \"run\"}" >> $out
done
cat $out
