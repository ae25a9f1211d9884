#!/usr/bin/env python3
"""A fresh batch of B stand-ins every step, end to end (GPU box only): where the time goes.   python tools/batch_e2e_probe.py [B]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from arpeggio_amd import synth, _capi, batch as _batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
distinct = [synth.proteinlike(seed=2 + k, id=f'standin{k}') for k in range(min(B, 8))]
pcs = [distinct[k % len(distinct)] for k in range(B)]
blobs = []
for r in range(2):
    big, off = _batch.concat_complexes(pcs[r:] + pcs[:r])
    blobs.append((_capi.pack_blob(big), off))
ctx = _capi.Context(0)
ctx.set_blob(blobs[0][0]); ctx.declare_batch(blobs[0][1])
cnt = ctx.run_launch(5.0, 0.1, False, 6.0)
buf = _capi.pinned_empty(int(cnt['atom_atom'] * 1.25) * 16 + (8 << 20), np.uint8)
ph = np.zeros(4)
n = 0
for k in range(44):
    b, off = blobs[k % 2]
    t0 = time.perf_counter(); ctx.set_blob(b)
    t1 = time.perf_counter(); ctx.declare_batch(off)
    t2 = time.perf_counter(); c = ctx.run_launch(5.0, 0.1, False, 6.0)
    t3 = time.perf_counter(); bags, buf = ctx.fetch_packed(buf)
    t4 = time.perf_counter()
    if k >= 4:
        ph += (t1 - t0, t2 - t1, t3 - t2, t4 - t3); n += 1
ph = ph / n * 1e3
print(json.dumps({'structures_per_batch': B, 'atoms': int(ctx.n), 'contacts': int(c['atom_atom']), 'upload_bytes': int(blobs[0][0].nbytes),
                  'ms': {'set_blob': round(ph[0], 4), 'declare_batch': round(ph[1], 4), 'first_pass': round(ph[2], 4), 'sort_fetch_packed': round(ph[3], 4),
                         'per_batch': round(float(ph.sum()), 4)}, 'us_per_structure': round(float(ph.sum()) / B * 1e3, 2)}))
