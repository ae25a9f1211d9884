"""Per-wave timeline of k_search<MODE_CONTACTS> (developer build: ARP_EXTRA_HIPCC_FLAGS=-DARP_SEARCH_TRACE).  GPU box only.
    ARP_EXTRA_HIPCC_FLAGS=-DARP_SEARCH_TRACE python -c "from arpeggio_amd import build; build.build(force=True)"; python tools/search_trace.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arpeggio_amd import synth, _capi

# python tools/search_trace.py [atoms | chain [residues]]: config 3 of that many atoms, or the chain model of tools/small_bench.py (clumps)
if len(sys.argv) > 1 and sys.argv[1] == 'chain':
    nr = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
    pc = synth.proteinlike(n_res=nr, n_waters=nr // 2)
else:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    pc = synth.config3(n, seed=3)
ctx = _capi.Context(0)
ctx.set_complex(pc)
ctx.set_grid_reuse(False)
for _ in range(5):
    ctx.run_launch(5.0, 0.1, False, 6.0)
L = _capi.load()
NW = 8192 * 8
ptr = C.c_uint64(0)
assert L.arp_debug_alloc(C.c_uint64(NW * 32), C.byref(ptr)) == 0
L.arp_debug_search_trace.argtypes = [C.c_void_p, C.c_uint64]
assert L.arp_debug_search_trace(ctx._h, ptr) == 0
ctx.run_launch(5.0, 0.1, False, 6.0)
ctx.device_synchronize()
buf = np.zeros(NW * 4, np.uint64)
L.arp_debug_read.argtypes = [C.c_uint64, C.c_void_p, C.c_uint64]
assert L.arp_debug_read(ptr, buf.ctypes.data_as(C.c_void_p), C.c_uint64(buf.nbytes)) == 0
t = buf.reshape(-1, 4)
t = t[t[:, 0] != 0]
t0 = t[:, 0].min()
TICK = 0.01   # us per tick of s_memrealtime (100 MHz)
b, l, e = ((t[:, k] - t0).astype(np.float64) * TICK for k in range(3))
xcc = (t[:, 3] & 0xFF).astype(int)
hw = ((t[:, 3] >> 8) & 0xFFFFFFFF).astype(np.int64)
cand = (t[:, 3] >> 40).astype(np.int64)
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7
q = lambda a: np.percentile(a, [0, 10, 50, 90, 100]).round(2).tolist()
print(f'{len(t)} waves; start {q(b)} us; leave the cell loops {q(l)}; end {q(e)}')
print('busy time of a wave (loops end - start):', q(l - b))
blk = e.reshape(-1, 8).max(1) if len(e) % 8 == 0 else e
print('block ends:', q(blk))
print('candidate tests of a wave:', q(cand), 'sum', int(cand.sum()))
print('correlation of a wave\'s tests with its busy time:', round(float(np.corrcoef(cand, l - b)[0, 1]), 3))
# per CU: when does the last wave of the CU leave its loops
key = xcc * 1000 + se * 100 + sh * 20 + cu
cu_end = {k: l[key == k].max() for k in np.unique(key)}
cu_n = {k: int((key == k).sum()) for k in np.unique(key)}
cu_work = {k: int(cand[key == k].sum()) for k in np.unique(key)}
ks = sorted(cu_end)
print('tests per CU:', q(np.array([cu_work[k] for k in ks])), ' correlation with the time its last wave leaves:', round(float(np.corrcoef([cu_work[k] for k in ks], [cu_end[k] for k in ks])[0, 1]), 3))
print('tests per microsecond of a CU (work / end):', q(np.array([cu_work[k] / cu_end[k] for k in ks])))
print(f'{len(cu_end)} distinct (xcd, se, sh, cu); waves per CU {q(np.array(list(cu_n.values())))}; last wave of a CU leaves the loops at {q(np.array(list(cu_end.values())))}')
for x in range(8):
    m = xcc == x
    if m.any():
        print(f'  XCD {x}: {int(m.sum())} waves, loops end {q(l[m])}, tests {int(cand[m].sum())}')
