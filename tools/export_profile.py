#!/usr/bin/env python3
"""Export a tools/profile.sh run (gpurun_out/prof_<tag>) into profiles/: kernel stats, per-launch PMC
averages and the corrected HBM traffic per launch that bench.py quotes in `roofline.traffic`.

    python tools/export_profile.py gpurun_out/prof_r1c round1_c [traffic.json: profiles/pmc_traffic.json, the one bench.py reads for its default workload]

Correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide
(16 B per lane) coalesced stream, so it is doubled; WRITE_SIZE is used as printed — it was calibrated here
on k_sift, whose output is exactly 15 B per contact (18.76 MB expected, 18.9 MB counted).
Both counters are printed by rocprofv3 in KiB.

Gathers are different (tools/calibrate_pmc_gather.py, profiles/README.md): every 128-byte row of a 256 MiB table read
once in random order shows FETCH_SIZE = 1.000 x the table, every 32-byte row 3.99 x — i.e. line fills are counted in
full, only merged streaming requests are halved.  k_sift reads its atom records by gather (counted in full) and one
coalesced stream, the pair list (8 B per pair, of which the counter shows half): its traffic is
FETCH_SIZE + WRITE_SIZE + 4 B x pairs.  The figure of the guide's rule for everything, 2 x FETCH + WRITE, is kept
beside it as `hbm_bytes_all_streams`.
"""
import collections
import csv
import json
import sys

base, tag = sys.argv[1], sys.argv[2]
traffic_json = sys.argv[3] if len(sys.argv) > 3 else 'profiles/pmc_traffic.json'
rows = list(csv.DictReader(open(f'{base}/trace/t_kernel_stats.csv')))
with open(f'profiles/{tag}_kernel_stats.csv', 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
    for r in rows:
        w.writerow([r['Name'].split('(')[0], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']])


def agg(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        d[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
    return d


out = collections.defaultdict(dict)
import os
for f in ('fetch/f_counter_collection.csv', 'write/w_counter_collection.csv', 'sq/s_counter_collection.csv', 'sq2/s_counter_collection.csv', 'tc/c_counter_collection.csv'):
    if not os.path.exists(f'{base}/{f}'):
        continue
    for k, v in agg(f'{base}/{f}').items():
        for c, x in v.items():
            out[k][c] = round(sum(x) / len(x), 1)
cols = ['FETCH_SIZE', 'WRITE_SIZE', 'SQ_WAVES', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_ACTIVE_INST_VALU',
        'SQ_WAIT_INST_ANY', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR',
        'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_SCA', 'SQ_INST_LEVEL_VMEM', 'TCP_TOTAL_CACHE_ACCESSES_sum',
        'TCP_TCC_READ_REQ_sum', 'TCC_HIT_sum', 'TCC_MISS_sum']
with open(f'profiles/{tag}_pmc_per_launch.csv', 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['Kernel'] + cols)
    for k, v in out.items():
        if not k.startswith('__amd'):
            w.writerow([k] + [v.get(c, '') for c in cols])
avg_ns = {r['Name'].split('(')[0]: float(r['AverageNs']) for r in rows}
traffic = {}
n_pairs = None
try:   # the bench line of the kernel-trace pass tells how many pairs one launch handled
    line = [l for l in open(f'{base}/trace.log') if l.startswith('{')][-1]
    n_pairs = int(json.loads(line)['pairs']['contacts_emitted'])
except (OSError, IndexError, KeyError, ValueError):
    pass
for k, v in out.items():
    if k.startswith('__amd') or 'FETCH_SIZE' not in v or 'WRITE_SIZE' not in v:
        continue
    key = {'void k_search<0>': 'k_search', 'void k_search<0, 1>': 'k_search', 'void k_search<0, 2>': 'k_search_tiles', 'void k_search<2>': 'k_mark_search',
           'void k_search<2, 1>': 'k_mark_search', 'k_compact_atoms': 'k_bin', 'void k_compact_atoms<512>': 'k_bin', 'void k_compact_atoms<1024>': 'k_bin', 'k_sift_planes': 'k_sift',
           'void k_sift_planes<0>': 'k_sift', 'void k_sift<0>': 'k_sift', 'void k_sift_planes<1>': 'k_sift_streaming', 'void k_sift<1>': 'k_sift_streaming',
           'void k_sift_planes<0, 0>': 'k_sift', 'void k_sift_planes<1, 0>': 'k_sift_streaming', 'void k_sift_planes<0, 1>': 'k_sift_gid', 'void k_sift_planes<1, 1>': 'k_sift_streaming_gid'}.get(k, k)
    all_streams = int((2 * v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024)
    if key in ('k_sift', 'k_sift_streaming') and n_pairs is not None:
        hbm = int((v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024 + 4 * n_pairs)
        model = 'gathers counted in full + half of the 8 B/pair list stream added back'
    else:
        hbm, model = all_streams, 'coalesced streams: 2 x FETCH_SIZE + WRITE_SIZE'
    traffic[key] = {'hbm_bytes_per_launch': hbm, 'hbm_bytes_all_streams': all_streams, 'model': model,
                    'fetch_kib_raw': v['FETCH_SIZE'], 'write_kib_raw': v['WRITE_SIZE'],
                    'valu_insts_per_launch': v.get('SQ_INSTS_VALU'),
                    'rocprof_avg_ns': avg_ns.get(k)}
json.dump({'source': f'{tag} (tools/profile.sh: separate --pmc FETCH_SIZE / WRITE_SIZE passes)',
           'correction': 'bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts half of a 16 B/lane stream); k_sift: '
                         'FETCH_SIZE + WRITE_SIZE + 4 B x pairs (line fills of gathers are counted in full: tools/calibrate_pmc_gather.py)',
           'kernels': traffic}, open(traffic_json, 'w'), indent=1)
for k in ('k_search', 'k_sift', 'k_mark_search', 'k_scatter_atoms'):
    if k in traffic:
        print(k, traffic[k])
