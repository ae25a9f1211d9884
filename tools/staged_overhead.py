#!/usr/bin/env python3
"""Cost of the three-stage (sharded) protocol versus the single-sync pass on one GPU (no neighbours)."""
import sys, time
sys.path.insert(0, '.')
from arpeggio_amd import synth, _capi
pc = synth.config3(100_000, seed=3)
ctx = _capi.Context(0)
ctx.set_complex(pc)
for _ in range(5):
    ctx.run_launch()
t0 = time.perf_counter()
for _ in range(50):
    ctx.run_launch()
a = (time.perf_counter() - t0) / 50 * 1e3
for _ in range(5):
    ctx.run_stage(0); ctx.run_stage(1); ctx.run_stage(2)
t0 = time.perf_counter()
for _ in range(50):
    ctx.run_stage(0); ctx.run_stage(1); ctx.run_stage(2)
b = (time.perf_counter() - t0) / 50 * 1e3
print(f'run_launch {a:.3f} ms   staged {b:.3f} ms')
