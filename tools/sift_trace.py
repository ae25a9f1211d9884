"""Per-wave timeline of the per-pair kernel (developer build: ARP_EXTRA_HIPCC_FLAGS=-DARP_SIFT_TRACE).  GPU box only.
    python tools/sift_trace.py [atoms]     (ARP_LIB_PATH = the traced library)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arpeggio_amd import synth, _capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
pc = synth.config3(n, seed=3)
ctx = _capi.Context(0)
ctx.set_complex(pc)
for _ in range(5):
    ctx.run_launch(5.0, 0.1, False, 6.0)
L = _capi.load()
NW = 8192 * 4
ptr = C.c_uint64(0)
assert L.arp_debug_alloc(C.c_uint64(NW * 64), C.byref(ptr)) == 0
L.arp_debug_search_trace.argtypes = [C.c_void_p, C.c_uint64]
assert L.arp_debug_search_trace(ctx._h, ptr) == 0
ctx.run_launch(5.0, 0.1, False, 6.0)
ctx.device_synchronize()
buf = np.zeros(NW * 8, np.uint64)
L.arp_debug_read.argtypes = [C.c_uint64, C.c_void_p, C.c_uint64]
assert L.arp_debug_read(ptr, buf.ctypes.data_as(C.c_void_p), C.c_uint64(buf.nbytes)) == 0
t = buf.reshape(-1, 8)
t = t[t[:, 0] != 0]
t0 = t[:, 0].min()
TICK = 0.01   # us per tick of s_memrealtime (100 MHz)
q = lambda a: np.percentile(a, [0, 10, 50, 90, 100]).round(2).tolist()
rel = lambda k: (t[:, k].astype(np.int64) - int(t0)) * TICK
work = t[:, 4] != 0
print(f'{len(t)} waves, {int(work.sum())} with a run of descriptors')
print('start', q(rel(0)), ' tables ready', q(rel(1)))
w = t[work]
r = lambda k: (w[:, k].astype(np.int64) - int(t0)) * TICK
print('first batch landed', q(r(3)))
print('loop ends', q(r(4)), ' wave ends', q(r(5)))
iters = (w[:, 7] & 0xFFFFFFFF).astype(np.int64); lanes = (w[:, 7] >> 32).astype(np.int64)
print('batches per wave', q(iters), ' lanes per batch', round(float(lanes.sum() / max(iters.sum(), 1)), 1), ' pairs', int(lanes.sum()))
print('time in the loop per wave', q(r(4) - r(3)), ' of which waiting at the end of a batch', q(w[:, 6].astype(np.int64) * TICK))
print('loop time per batch', q((r(4) - r(3)) / np.maximum(iters, 1)))
