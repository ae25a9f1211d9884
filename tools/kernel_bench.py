#!/usr/bin/env python3
"""Isolated per-kernel times of the contact pipeline (bin, scan, scatter, search, sift on ONE stream, nothing
running beside them): the figures to compare kernel variants with.  GPU box only.

    python tools/kernel_bench.py [--atoms 100000] [--steps 200]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arpeggio_amd import synth, _capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--atoms', type=int, default=100_000)
ap.add_argument('--steps', type=int, default=200)
args = ap.parse_args()
pc = synth.config3(args.atoms, seed=3)
ctx = _capi.Context(0)
ctx.set_complex(pc)
ctx.run_launch(5.0, 0.1, False, 6.0)          # sizes every buffer, builds the selection
for _ in range(5):
    ctx.atom_contacts_launch(5.0, 0.1, False)
t0 = time.perf_counter()
for _ in range(args.steps):
    ctx.atom_contacts_launch(5.0, 0.1, False)
wall = (time.perf_counter() - t0) / args.steps * 1e3
ctx.set_profiling(True)
ctx.kernel_times(reset=True)
for _ in range(args.steps):
    ctx.atom_contacts_launch(5.0, 0.1, False)
kt = ctx.kernel_times(reset=True)
print(json.dumps({'atoms': args.atoms, 'contacts_only_ms_per_pass': round(wall, 4),
                  'kernel_us': {k: round(v['ms'] / max(v['launches'], 1) * 1e3, 2) for k, v in kt.items() if v['launches']}}))
