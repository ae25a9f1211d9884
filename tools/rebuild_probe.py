import sys, os, time, json
sys.path.insert(0, os.getcwd())
from arpeggio_amd import synth, _capi
pc = synth.config3(100000, seed=3)
ctx = _capi.Context(0); ctx.set_complex(pc); ctx.set_grid_reuse(False)
for _ in range(8): ctx.run_launch(5.0, 0.1, False, 6.0)
t0 = time.perf_counter()
for _ in range(400): ctx.run_launch(5.0, 0.1, False, 6.0)
wall = (time.perf_counter() - t0) / 400 * 1e3
ctx.set_profiling(True); ctx.kernel_times(reset=True)
for _ in range(50): ctx.run_launch(5.0, 0.1, False, 6.0)
kt = ctx.kernel_times(reset=True)
print(json.dumps({'tag': sys.argv[1] if len(sys.argv) > 1 else '', 'ms_per_pass_grid_built': round(wall, 4), 'kernel_us': {k: round(v['ms'] / max(v['launches'], 1) * 1e3, 2) for k, v in kt.items() if v['launches']}}))
