cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/calg
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/calg/fetch -o f -- python tools/calibrate_pmc_gather.py > gpurun_out/calg/fetch.log 2>&1 < /dev/null
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/calg/write -o w -- python tools/calibrate_pmc_gather.py > gpurun_out/calg/write.log 2>&1 < /dev/null
python - <<'PY'
import csv, glob, collections
for what in ('fetch', 'write'):
    for f in glob.glob(f'gpurun_out/calg/{what}/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r['Kernel_Name'][:70], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k, v in acc.items():
            if len(v) >= 5 and max(v) > 1000:
                print(what, k, len(v), 'median KiB', sorted(v)[len(v)//2])
PY
tail -3 gpurun_out/calg/fetch.log
