#!/usr/bin/env python3
"""Wall clock of run_arpeggio on the 1tqn_h stand-in (BASELINE configs[0]/[1] family: ~5.9 k atoms): whole structure
and ligand selection (residue 508).  Launch-bound sizes; the figure is ms per pass, results resident.  GPU box only."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arpeggio_amd import synth, _capi  # noqa: E402

import argparse
ap = argparse.ArgumentParser()
ap.add_argument('--n-res', type=int, default=480, help='480 = the 1tqn_h stand-in (5.9 k atoms); 8000 = ~96 k atoms of the same chain model')
args = ap.parse_args()
pc = synth.proteinlike(n_res=args.n_res, n_waters=300 if args.n_res == 480 else args.n_res // 2)
ctx = _capi.Context(0)
ctx.set_complex(pc)
out = {'atoms': int(pc.n_atoms), 'hydrogen_atoms': int(((np.asarray(pc.flags) & 8) != 0).sum())}
lig = (np.asarray(pc.res_seq)[np.asarray(pc.res_id)] == 508).astype(np.uint8)
for name, sel in (('whole_structure', np.ones(pc.n_atoms, np.uint8)), ('ligand_508', lig)):
    if sel is None or sel.sum() == 0:
        continue
    ctx.set_selection(sel)
    for _ in range(10):
        counts = ctx.run_launch(5.0, 0.1, False, 6.0)
    t0 = time.perf_counter()
    for _ in range(500):
        counts = ctx.run_launch(5.0, 0.1, False, 6.0)
    st = ctx.stats()
    ctx.set_profiling(True)
    ctx.kernel_times(reset=True)
    for _ in range(50):
        ctx.run_launch(5.0, 0.1, False, 6.0)
    kt = ctx.kernel_times(reset=True)
    ctx.set_profiling(False)
    kernel_us = {k: round(v['ms'] / 50 * 1e3, 1) for k, v in kt.items() if v['launches']}
    out[name] = {'candidate_pairs': st['candidates'], 'ms_per_pass': round((time.perf_counter() - t0) / 500 * 1e3, 4), 'kernel_us_per_pass': kernel_us, 'selected': int(sel.sum()), **counts}
print(json.dumps(out))
