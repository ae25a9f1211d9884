#!/usr/bin/env python3
"""Lower bound for a pass pipelined in S slabs: S contexts, each holding a structure of N / S atoms (same density), one host
thread: enqueue the pass of every context (each on its own stream: 3 launches), then wait for all.  The time of one round is what
ONE N-atom structure would cost if it could be cut into S independent slabs at no price — no halo, no second layer of the
neighbouring slab, no merged output, no event between the slabs' streams.  A real slab pipeline can only be slower.
    python tools/slab_bound_probe.py [--atoms 100000] [--keep-grid]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arpeggio_amd import _capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--atoms', type=int, default=100_000)
ap.add_argument('--rounds', type=int, default=400)
ap.add_argument('--threads', action='store_true', help='one host thread per slab (a barrier before every round)')
ap.add_argument('--only', type=int, default=0, help='this S only (for a rocprofv3 trace)')
args = ap.parse_args()
import threading
out = {'atoms': args.atoms, 'host_threads': 'one per slab' if args.threads else 1, 'slabs': {}}
for S in ((args.only,) if args.only else (1, 2, 3, 4, 6)):
    pc = synth.config3(args.atoms // S, seed=3)
    ctxs = [_capi.Context(0) for _ in range(S)]
    for c in ctxs:
        c.set_complex(pc)
        c.set_grid_reuse(False)          # every pass builds its grid, as the headline pass does
        for _ in range(6):
            c.run_launch()
    best = 1e9
    for rep in range(3):
        if args.threads and S > 1:
            bar = threading.Barrier(S)

            def work(c):
                for _ in range(args.rounds):
                    bar.wait()
                    c.run_enqueue()
                    c.run_wait()
            ths = [threading.Thread(target=work, args=(c,)) for c in ctxs]
            t0 = time.perf_counter()
            for t in ths:
                t.start()
            for t in ths:
                t.join()
        else:
            t0 = time.perf_counter()
            for _ in range(args.rounds):
                for c in ctxs:
                    c.run_enqueue()
                for c in ctxs:
                    c.run_wait()
        best = min(best, (time.perf_counter() - t0) / args.rounds)
    out['slabs'][S] = round(best * 1e3, 4)
    for c in ctxs:
        c.close()
print(json.dumps(out))
