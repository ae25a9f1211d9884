import sys, time, json, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from arpeggio_amd import synth, _capi
import numpy as np
L = (100000 / 0.05) ** (1 / 3)
for tag, pc in (('with', synth.config3(100000, seed=3)), ('without', synth.make_synthetic(100000, seed=3, box=(L, L, L), n_rings=0, n_amides=0))):
    ctx = _capi.Context(0); ctx.set_complex(pc)
    for _ in range(40): ctx.run_launch(5.0, 0.1, False, 6.0)
    t0 = time.perf_counter()
    for _ in range(300): c = ctx.run_launch(5.0, 0.1, False, 6.0)
    dt = (time.perf_counter() - t0) / 300 * 1e3
    print(tag, round(dt, 4), 'ms/step', c['atom_atom'], ctx.host_times(reset=True))
    ctx.close()
