#!/usr/bin/env python3
"""A new structure every step (blob upload + first pass), for a kernel trace of the per-structure work:
    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python tools/fresh_probe.py
    python tools/fresh_timeline.py out/t_kernel_trace.csv"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arpeggio_amd import _capi, synth  # noqa: E402

standin = 'standin' in sys.argv[1:]      # python tools/fresh_probe.py standin: the 5.9 k-atom 1tqn_h stand-in
e2e = 'e2e' in sys.argv[1:]              # ... e2e: every structure also sorted and fetched (arp_fetch_packed)
blobs = [_capi.pack_blob(synth.proteinlike(seed=2 + k) if standin else synth.config3(100_000, seed=3 + k)) for k in range(3)]
ctx = _capi.Context(0)
if e2e:
    ctx.set_sort_after_pass(True)
buf = None
for rep in range(4):
    for b in blobs:
        ctx.set_blob(b)
        cnt = ctx.run_launch(5.0, 0.1, False, 6.0)
        if e2e:
            if buf is None:
                import numpy as np
                buf = _capi.pinned_empty(int(cnt['atom_atom'] * 1.25) * 16 + (4 << 20), np.uint8)
            ctx.fetch_packed(buf)
