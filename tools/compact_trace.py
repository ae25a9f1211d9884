"""Per-wave timeline of k_compact_atoms (developer build: -DARP_COMPACT_TRACE).  GPU box only.  ARP_LIB_PATH = the traced library"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arpeggio_amd import synth, _capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
pc = synth.config3(n, seed=3)
ctx = _capi.Context(0)
ctx.set_complex(pc)
ctx.set_grid_reuse(False)
for _ in range(5):
    ctx.run_launch(5.0, 0.1, False, 6.0)
L = _capi.load()
NW = 8192 * 16
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
q = lambda a: np.percentile(a, [0, 10, 50, 90, 100]).round(2).tolist()
r = lambda k: (t[:, k].astype(np.int64) - int(t0)) * 0.01
print(f'{len(t)} waves; start {q(r(0))}; columns loaded + counted {q(r(1))}; base known (look-back done) {q(r(2))}; stores done {q(r(3))}')
