#!/usr/bin/env python3
"""One host thread, K contexts: enqueue the pass of every context, then wait for each (arp_run_enqueue / arp_run_wait).
    python tools/pipeline_probe.py [--atoms 100000] [--standin]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arpeggio_amd import _capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--atoms', type=int, default=100_000)
ap.add_argument('--standin', action='store_true')
args = ap.parse_args()
pc = synth.proteinlike() if args.standin else synth.config3(args.atoms, seed=3)
for k in (1, 2, 3, 4, 6, 8):
    ctxs = [_capi.Context(0) for _ in range(k)]
    for c in ctxs:
        c.set_complex(pc)
        for _ in range(6):
            c.run_launch()
    rounds = max(20, 600 // k)
    t0 = time.perf_counter()
    for _ in range(rounds):
        for c in ctxs:
            c.run_enqueue()
        for c in ctxs:
            c.run_wait()
    dt = (time.perf_counter() - t0) / (rounds * k)
    print(f'{k} contexts, one thread: {dt * 1e3:.4f} ms per structure')
    for c in ctxs:
        c.close()
