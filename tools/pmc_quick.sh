#!/bin/bash
# tools/pmc_quick.sh TAG — VALU / SALU / LDS instruction counts and kernel times of the resident pass (GPU box only)
set -u
tag=${1:-q}
out=gpurun_out/pmcq_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="${PROBE_CMD:-python tools/pass_probe.py --steps 60} ${PROBE_ARGS:-}"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- $B > "$out/trace.log" 2>&1 < /dev/null
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d "$out/sq" -o s -- $B > "$out/sq.log" 2>&1 < /dev/null
python - "$out" <<'PY'
import csv, collections, sys, glob
out = sys.argv[1]
for f in glob.glob(out + '/trace/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r['Percentage']) > 1.0: print('%-44s calls %5s avg %8.2f us' % (r['Name'].split('(')[0][:44], r['Calls'], float(r['AverageNs']) / 1e3))
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/sq/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        d[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in d.items():
    if len(v.get('SQ_INSTS_VALU', [])) < 20: continue
    m = {c: sum(x) / len(x) for c, x in v.items()}
    print('%-44s VALU %10.0f SALU %10.0f LDS %8.0f  valu/wavecyc %.2f wait %.2f' % (k[:44], m['SQ_INSTS_VALU'], m['SQ_INSTS_SALU'], m['SQ_INSTS_LDS'], m['SQ_ACTIVE_INST_VALU'] / m['SQ_WAVE_CYCLES'], m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']))
PY
