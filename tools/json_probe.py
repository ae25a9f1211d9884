#!/usr/bin/env python3
"""Time of the native JSON writer on a config-3-sized result (host only): threads x strategy.  python tools/json_probe.py [dir]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arpeggio_amd import synth
from arpeggio_amd.core import export
pc = synth.config3(100000, seed=3); pc.ensure_labels()
rs = np.random.RandomState(0); n = 1250000
i = rs.randint(0, pc.n_atoms - 1, n).astype(np.int32); j = (i + 1 + rs.randint(0, 50, n)).clip(0, pc.n_atoms - 1).astype(np.int32)
vals = np.array([16, 8, 4, 16 | 2048, 16 | 32 | 8192, 8 | 256, 1, 16 | 64 | 16384], np.uint16)
bags = {'atom_atom': dict(i=i, j=j, dist=rs.uniform(1, 5, n).astype(np.float32), sift=vals[rs.randint(0, len(vals), n)], ctype=rs.randint(0, 6, n).astype(np.uint8))}
base = sys.argv[1] if len(sys.argv) > 1 else None
for mm in ('0', '1'):
    for thr in ('1', '4', '8', '16', ''):
        os.environ['ARP_EXPORT_MMAP'] = mm
        if thr: os.environ['ARP_EXPORT_THREADS'] = thr
        else: os.environ.pop('ARP_EXPORT_THREADS', None)
        best = 1e9
        for rep in range(2):
            with tempfile.TemporaryDirectory(dir=base) as td:
                t = time.perf_counter()
                export.write_contacts_json(os.path.join(td, 'o.json'), pc, bags, pc.component_types)
                best = min(best, time.perf_counter() - t)
        print(f'mmap={mm} threads={thr or "auto"}: {best * 1e3:.0f} ms', flush=True)
