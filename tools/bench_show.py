"""Print the interesting keys of a bench.py JSON line.  python tools/bench_show.py <file>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "kernel_ms", "wall_clock_per_structure_ms", "run_arpeggio_on_an_unseen_structure_ms", "roofline",
          "roofline_pass", "pass_with_grid_kept", "gpu_legs_s", "end_to_end", "roofline_valu", "throughput_several_in_flight"):
    print(k, '=', d.get(k))
