#!/bin/bash
# tools/pmc_any.sh TAG "COUNTER COUNTER ..." — per-kernel averages of any counter group over the resident pass (GPU box only;
# PROBE_CMD / PROBE_ARGS as in pmc_quick.sh; ARP_LIB_PATH selects the library)
set -u
tag=${1:-q}
ctrs=${2:-SQ_INSTS_VALU}
out=gpurun_out/pmca_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="${PROBE_CMD:-python tools/pass_probe.py --steps 60} ${PROBE_ARGS:-}"
timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$out/c" -o s -- $B > "$out/c.log" 2>&1 < /dev/null
python - "$out" <<'PY'
import csv, collections, sys, glob
out = sys.argv[1]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/c/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        d[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in d.items():
    if max(len(x) for x in v.values()) < 20 or k.startswith('__amd'): continue
    print('%-40s ' % k[:40] + ' '.join('%s %.0f' % (c.replace('SQ_', '').replace('_sum', ''), sum(x) / len(x)) for c, x in sorted(v.items())))
PY
