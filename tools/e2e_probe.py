#!/usr/bin/env python3
"""A new structure every step, end to end (GPU box only): blob upload + validation + per-structure set-up + pass + every result
bag into page-locked host buffers; the loop figure and one structure's breakdown.   python tools/e2e_probe.py [--atoms N] [--tag x]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arpeggio_amd import synth, _capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--atoms', type=int, default=100_000)
ap.add_argument('--workload', default='config3')
ap.add_argument('--tag', default='')
args = ap.parse_args()
fresh = [synth.proteinlike(seed=2 + k) if args.workload == 'standin' else synth.config3(args.atoms, seed=3 + k) for k in range(4)]
blobs = [_capi.pack_blob(f) for f in fresh]
ctx = _capi.Context(0)
ctx.set_blob(blobs[0])
cnt = ctx.run_launch(5.0, 0.1, False, 6.0)
cap = int(cnt['atom_atom'] * 1.2) + 1024
bufs = (ctx.pinned_contact_buffers(cap), {k: ctx.pinned_bag_buffers(k, 4 * max(cnt[k], 256)) for k in ('plane_plane', 'atom_plane', 'group_group', 'group_plane')})


def one(blob):
    ctx.set_blob(blob)
    c = ctx.run_launch(5.0, 0.1, False, 6.0)
    res = ctx.atom_contacts_fetch(c['atom_atom'], sort=False, out=bufs[0])
    bags = {k: ctx.fetch_bag(k, sort=False, out=bufs[1][k]) for k in bufs[1]}
    return c, res, bags


for k in range(8):
    one(blobs[k % 4])
n, t1 = 0, time.perf_counter()
while n < 60 or time.perf_counter() - t1 < 0.5:
    one(blobs[n % 4]); n += 1
e2e = (time.perf_counter() - t1) / n * 1e3
br = {}
for rep in range(5):
    tt = time.perf_counter(); ctx.set_blob(blobs[1]); a = (time.perf_counter() - tt) * 1e3
    tt = time.perf_counter(); c = ctx.run_launch(5.0, 0.1, False, 6.0); b = (time.perf_counter() - tt) * 1e3
    tt = time.perf_counter(); res = ctx.atom_contacts_fetch(c['atom_atom'], sort=False, out=bufs[0]); d = (time.perf_counter() - tt) * 1e3
    tt = time.perf_counter(); bags = {k: ctx.fetch_bag(k, sort=False, out=bufs[1][k]) for k in bufs[1]}; e = (time.perf_counter() - tt) * 1e3
    for k, v in (('upload_validate_ms', a), ('first_pass_ms', b), ('download_contacts_ms', d), ('download_bags_ms', e)):
        br[k] = min(br.get(k, 1e9), v)
print(json.dumps({'tag': args.tag, 'atoms': int(fresh[0].n_atoms), 'e2e_ms_per_structure': round(e2e, 4), 'breakdown_ms_best_of_5': {k: round(v, 4) for k, v in br.items()},
                  'host_times': ctx.host_times(reset=True)}))
