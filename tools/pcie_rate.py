#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) time per structure on config 3: upload + run + download.  GPU box only."""
import sys
import time

sys.path.insert(0, '.')
import numpy as np
from arpeggio_amd import synth, _capi

pc = synth.config3(100_000, seed=3)
ctx = _capi.Context(0)
ts = {'upload': [], 'run': [], 'download': [], 'download_pinned': []}
pinned = None
for it in range(8):
    t0 = time.perf_counter(); ctx.set_complex(pc); t1 = time.perf_counter()
    counts = ctx.run_launch(); t2 = time.perf_counter()
    out = ctx.atom_contacts_fetch(counts['atom_atom'], sort=False)
    for b in ('plane_plane', 'atom_plane', 'group_group', 'group_plane'):
        ctx.fetch_bag(b)
    t3 = time.perf_counter()
    if pinned is None:
        pinned = ctx.pinned_contact_buffers(int(counts['atom_atom'] * 1.1))
    out2 = ctx.atom_contacts_fetch(counts['atom_atom'], sort=False, out=pinned)
    for b in ('plane_plane', 'atom_plane', 'group_group', 'group_plane'):
        ctx.fetch_bag(b)
    t4 = time.perf_counter()
    assert np.array_equal(out2['i'], out['i']) and np.array_equal(out2['sift'], out['sift'])
    if it >= 2:
        ts['upload'].append(t1 - t0); ts['run'].append(t2 - t1); ts['download'].append(t3 - t2); ts['download_pinned'].append(t4 - t3)
med = {k: float(np.median(v)) * 1e3 for k, v in ts.items()}
cand = ctx.stats()['candidates']
tot = med['upload'] + med['run'] + med['download']
tot_pinned = med['upload'] + med['run'] + med['download_pinned']
print({k: round(v, 3) for k, v in med.items()}, 'total_ms', round(tot, 3), 'total_ms_pinned_results', round(tot_pinned, 3), 'pairs/s', f'{cand / (tot * 1e-3):.3e}',
      'upload_MB', round(sum(getattr(pc, k).nbytes for k in pc._ARRAYS[:21]) / 1e6, 1), 'download_MB', round(15 * len(out['i']) / 1e6, 1))
