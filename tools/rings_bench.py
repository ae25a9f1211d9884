#!/usr/bin/env python3
"""BASELINE configs[4]: 10 k aromatic rings (+ 10 k amides) — wall clock of the ring / amide kernels and of the whole
pass on the ring-rich synthetic set.  GPU box only."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arpeggio_amd import synth, _capi  # noqa: E402

pc = synth.config5()
ctx = _capi.Context(0)
ctx.set_complex(pc)
counts = ctx.run_launch(5.0, 0.1, False, 6.0)
for _ in range(10):
    ctx.run_launch(5.0, 0.1, False, 6.0)
t0 = time.perf_counter()
for _ in range(200):
    ctx.run_launch(5.0, 0.1, False, 6.0)
whole = (time.perf_counter() - t0) / 200 * 1e3
out = {'atoms': int(pc.n_atoms), 'rings': int(pc.n_rings), 'amides': int(pc.n_amides), 'run_arpeggio_ms': round(whole, 4), **counts}
for name in ('plane_plane', 'atom_plane', 'group_group', 'group_plane'):
    for _ in range(5):
        ctx.launch_bag(name)
    t0 = time.perf_counter()
    for _ in range(200):
        ctx.launch_bag(name)
    out[name + '_ms'] = round((time.perf_counter() - t0) / 200 * 1e3, 4)
ctx.set_profiling(True)
for name in ('plane_plane', 'atom_plane', 'group_group', 'group_plane'):
    ctx.kernel_times(reset=True)
    for _ in range(50):
        ctx.launch_bag(name)
    kt = ctx.kernel_times(reset=True)['planes']
    out[name + '_kernel_us'] = round(kt['ms'] / max(kt['launches'], 1) * 1e3, 2)
ctx.set_profiling(False)
R = pc.n_rings
out['ordered_ring_pairs_of_the_reference_loop'] = R * R
out['reference_ring_pairs_per_s'] = round(R * R / (out['plane_plane_ms'] * 1e-3), 1)
print(json.dumps(out))
