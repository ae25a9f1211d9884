#!/usr/bin/env python3
"""First pass over a structure that has just been uploaded (GPU box only): arp_set_blob, then the timed arp_run_launch — static columns,
their spatial order, grid, search, per-pair kernel, candidate lists of the ring / amide loops and their evaluation.  Median and best
of many structures; with --e2e also upload + pass + device sort + one-copy fetch per structure.
    python tools/first_pass_probe.py [--atoms N] [--workload config3|standin] [--reps 60] [--tag x] [--e2e]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arpeggio_amd import synth, _capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--atoms', type=int, default=100_000)
ap.add_argument('--workload', default='config3')
ap.add_argument('--reps', type=int, default=60)
ap.add_argument('--tag', default='')
ap.add_argument('--e2e', action='store_true')
args = ap.parse_args()
fresh = [synth.proteinlike(seed=2 + k) if args.workload == 'standin' else synth.config3(args.atoms, seed=3 + k) for k in range(4)]
blobs = [_capi.pack_blob(f) for f in fresh]
ctx = _capi.Context(0)
for k in range(8):
    ctx.set_blob(blobs[k % 4]); cnt = ctx.run_launch(5.0, 0.1, False, 6.0)
up, fp = [], []
for k in range(args.reps):
    t0 = time.perf_counter(); ctx.set_blob(blobs[k % 4]); t1 = time.perf_counter()
    cnt = ctx.run_launch(5.0, 0.1, False, 6.0); t2 = time.perf_counter()
    up.append((t1 - t0) * 1e3); fp.append((t2 - t1) * 1e3)
out = {'tag': args.tag, 'atoms': int(fresh[0].n_atoms), 'first_pass_ms_median': round(float(np.median(fp)), 4), 'first_pass_ms_min': round(min(fp), 4),
       'upload_validate_ms_median': round(float(np.median(up)), 4), 'contacts': int(cnt['atom_atom']),
       'bags': {k: int(cnt[k]) for k in ('plane_plane', 'atom_plane', 'group_group', 'group_plane')}}
if args.e2e:
    ctx.set_sort_after_pass(os.environ.get('PROBE_SORT_AFTER_PASS', '1') != '0')
    buf = _capi.pinned_empty(int(cnt['atom_atom'] * 1.25) * 16 + (4 << 20), np.uint8)
    for k in range(6):
        ctx.set_blob(blobs[k % 4]); ctx.run_launch(5.0, 0.1, False, 6.0); ctx.fetch_packed(buf)
    n, t0 = 0, time.perf_counter()
    while n < 40 or time.perf_counter() - t0 < 0.5:
        ctx.set_blob(blobs[n % 4]); ctx.run_launch(5.0, 0.1, False, 6.0); ctx.fetch_packed(buf); n += 1
    out['e2e_ms_per_structure'] = round((time.perf_counter() - t0) / n * 1e3, 4)
print(json.dumps(out))
