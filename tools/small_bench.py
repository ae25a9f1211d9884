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

pc = synth.proteinlike()
ctx = _capi.Context(0)
ctx.set_complex(pc)
out = {'atoms': int(pc.n_atoms)}
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
    out[name] = {'ms_per_pass': round((time.perf_counter() - t0) / 500 * 1e3, 4), 'selected': int(sel.sum()), **counts}
print(json.dumps(out))
