#!/usr/bin/env python3
"""Resident-pass time of a batch of B stand-in structures (GPU box only).  python tools/batch_probe.py --batch 64 [--tag x]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arpeggio_amd import synth, _capi, batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--steps', type=int, default=300)
ap.add_argument('--tag', default='')
args = ap.parse_args()
distinct = [synth.proteinlike(seed=2 + k) for k in range(min(args.batch, 8))]
pcs = [distinct[k % len(distinct)] for k in range(args.batch)]
ctx = _capi.Context(0)
ctx.set_batch(pcs)
for _ in range(8):
    counts = ctx.run_launch(5.0, 0.1, False, 6.0)
t0 = time.perf_counter()
for _ in range(args.steps):
    ctx.run_launch(5.0, 0.1, False, 6.0)
wall = (time.perf_counter() - t0) / args.steps * 1e3
st = ctx.stats()
ctx.set_profiling(True)
ctx.kernel_times(reset=True)
for _ in range(50):
    ctx.run_launch(5.0, 0.1, False, 6.0)
kt = ctx.kernel_times(reset=True)
print(json.dumps({'tag': args.tag, 'batch': args.batch, 'ms_per_pass': round(wall, 4), 'us_per_structure': round(wall / args.batch * 1e3, 3),
                  'cells': int(st['cells']), 'binned': int(st['binned']), 'candidates': int(st['candidates']), 'contacts': int(counts['atom_atom']),
                  'kernel_us': {k: round(v['ms'] / max(v['launches'], 1) * 1e3, 2) for k, v in kt.items() if v['launches']}}))
