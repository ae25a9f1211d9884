#!/usr/bin/env python3
"""The 2 M-atom structure of BASELINE configs[3] (or an N-th of it: one GPU's slab) on one GPU: resident pass + per-kernel times.
    python tools/config4_probe.py [--slabs 1|2|4|8] [--steps 60]     (ARP_GRID_LONGEST_AXIS_SLOWEST=0: the grid as before round 6)"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arpeggio_amd import synth, _capi  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument('--slabs', type=int, default=1)
ap.add_argument('--steps', type=int, default=60)
ap.add_argument('--tag', default='')
args = ap.parse_args()
pc = synth.slab_config(2_000_000 // 8 // args.slabs * 1, 8, seed=4) if args.slabs == 1 else synth.slab_config(2_000_000 // args.slabs // (8 // args.slabs), 8 // args.slabs, seed=4)
ctx = _capi.Context(0)
ctx.set_complex(pc)
ctx.set_grid_reuse(False)
for _ in range(5):
    counts = ctx.run_launch(5.0, 0.1, False, 6.0)
t0 = time.perf_counter()
for _ in range(args.steps):
    ctx.run_launch(5.0, 0.1, False, 6.0)
wall = (time.perf_counter() - t0) / args.steps * 1e3
ctx.set_profiling(True)
ctx.kernel_times(reset=True)
for _ in range(10):
    ctx.run_launch(5.0, 0.1, False, 6.0)
kt = ctx.kernel_times(reset=True)
st = ctx.stats()
print(json.dumps({'tag': args.tag, 'env': os.environ.get('ARP_GRID_LONGEST_AXIS_SLOWEST', '1'), 'atoms': int(pc.n_atoms), 'ms_per_pass': round(wall, 4), 'contacts': int(counts['atom_atom']),
                  'candidates': int(st['candidates']), 'cells': int(st['cells']),
                  'kernel_us': {k: round(v['ms'] / max(v['launches'], 1) * 1e3, 2) for k, v in kt.items() if v['launches']}}))
