#!/usr/bin/env python3
"""Resident-pass time and candidate pairs/s at several structure sizes (GPU box only).

    python tools/size_sweep.py [atoms ...]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arpeggio_amd import synth, _capi  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [5000, 100_000, 250_000, 1_000_000]
for n in sizes:
    pc = synth.config3(n, seed=3)
    ctx = _capi.Context(0)
    ctx.set_complex(pc)
    for _ in range(8):
        ctx.run_launch(5.0, 0.1, False, 6.0)
    reps = max(20, int(4e7 / n))
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.run_launch(5.0, 0.1, False, 6.0)
    ms = (time.perf_counter() - t0) / reps * 1e3
    st = ctx.stats()
    ctx.set_profiling(True)
    ctx.kernel_times(reset=True)
    for _ in range(20):
        ctx.run_launch(5.0, 0.1, False, 6.0)
    kt = ctx.kernel_times(reset=True)
    print(json.dumps({'atoms': n, 'ms_per_pass': round(ms, 4), 'candidate_pairs_per_s': round(st['candidates'] / ms * 1e3, 1),
                      'contacts': st['emitted'], 'kernel_us': {k: round(v['ms'] / max(v['launches'], 1) * 1e3, 1) for k, v in kt.items() if v['launches']}}))
    ctx.close()
