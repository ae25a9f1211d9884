#!/usr/bin/env python3
"""Resident-pass time and per-kernel HIP-event times of one structure (GPU box only): the quick figure to compare
kernel variants with.    python tools/pass_probe.py [--atoms 100000] [--steps 400] [--workload config3|standin|noplanes]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arpeggio_amd import synth, _capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--atoms', type=int, default=100_000)
ap.add_argument('--steps', type=int, default=400)
ap.add_argument('--workload', default='config3')
ap.add_argument('--tag', default='')
args = ap.parse_args()
if args.workload == 'standin':
    pc = synth.proteinlike(seed=1)
elif args.workload == 'noplanes':     # config 3's atoms without rings and amides: the last launch is the per-pair kernel alone
    L = (args.atoms / 0.05) ** (1 / 3)
    pc = synth.make_synthetic(args.atoms, seed=3, box=(L, L, L), n_rings=0, n_amides=0)
else:
    pc = synth.config3(args.atoms, seed=3)
ctx = _capi.Context(0)
ctx.set_complex(pc)
for _ in range(8):
    counts = ctx.run_launch(5.0, 0.1, False, 6.0)
t0 = time.perf_counter()
for _ in range(args.steps):
    ctx.run_launch(5.0, 0.1, False, 6.0)
wall = (time.perf_counter() - t0) / args.steps * 1e3
ctx.set_profiling(True)
ctx.kernel_times(reset=True)
for _ in range(50):
    ctx.run_launch(5.0, 0.1, False, 6.0)
kt = ctx.kernel_times(reset=True)
print(json.dumps({'tag': args.tag, 'atoms': int(pc.n_atoms), 'ms_per_pass': round(wall, 4), 'contacts': int(counts['atom_atom']),
                  'kernel_us': {k: round(v['ms'] / max(v['launches'], 1) * 1e3, 2) for k, v in kt.items() if v['launches']}}))
