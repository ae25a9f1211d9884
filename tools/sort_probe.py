"""Time of the device-side canonical order (arp_atom_contacts_sort) and of the one-copy fetch, config 3.
    python tools/sort_probe.py [atoms]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arpeggio_amd import synth, _capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
pc = synth.config3(n, seed=3)
ctx = _capi.Context(0)
ctx.set_complex(pc)
cnt = ctx.run_launch(5.0, 0.1, False, 6.0)
k = cnt['atom_atom']
ts = []
for _ in range(30):
    ctx.run_launch(5.0, 0.1, False, 6.0)
    ctx.device_synchronize()
    t0 = time.perf_counter()
    ctx.sort_contacts()
    ctx.device_synchronize()
    ts.append(time.perf_counter() - t0)
print(f'{n} atoms, {k} records: sort {np.median(ts) * 1e6:.1f} us (min {min(ts) * 1e6:.1f})')
buf = _capi.pinned_empty(k * 16 + (1 << 20), np.uint8)
cb = ctx.pinned_contact_buffers(k + 1024)
for name, fn in (('five copies, unsorted', lambda: ctx.atom_contacts_fetch(k, sort=False, out=cb)),
                 ('five copies, device-sorted', lambda: ctx.atom_contacts_fetch(k, sort=True, out=cb)),
                 ('one packed copy, device-sorted + bags', lambda: ctx.fetch_packed(buf))):
    ts = []
    for _ in range(20):
        ctx.run_launch(5.0, 0.1, False, 6.0)
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    print(f'  fetch {name}: {np.median(ts) * 1e3:.3f} ms')
raw = ctx.atom_contacts_fetch(k, sort=False)
t0 = time.perf_counter()
o = np.lexsort((raw['j'], raw['i']))
s = {kk: v[o] for kk, v in raw.items()}
print(f'  host np.lexsort + 5 gathers: {(time.perf_counter() - t0) * 1e3:.1f} ms')
# copies alone (the bag is already sorted: no sort kernels in these calls)
ctx.run_launch(5.0, 0.1, False, 6.0)
ctx.sort_contacts(); ctx.device_synchronize()
for name, fn in (('five copies from the sorted slab', lambda: ctx.atom_contacts_fetch(k, sort=False, out=cb)),
                 ('one packed copy (no sort)', lambda: ctx.fetch_packed(buf, sort_bags=False)),
                 ('five copies from the sorted slab', lambda: ctx.atom_contacts_fetch(k, sort=False, out=cb)),
                 ('one packed copy (no sort)', lambda: ctx.fetch_packed(buf, sort_bags=False))):
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    print(f'  {name}: {np.median(ts) * 1e3:.3f} ms (min {min(ts) * 1e3:.3f})')
