#!/usr/bin/env python3
"""Stress of the end-of-pass publication (GPU box only): many thousand passes over structures of different sizes, from one thread
and from three threads with a context each; every pass must report the counters of the first.  python tools/pub_stress.py [--passes N]"""
import argparse, os, sys, threading, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arpeggio_amd import synth, _capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--passes', type=int, default=20000)
args = ap.parse_args()
cases = [synth.config3(100_000, seed=3), synth.proteinlike(seed=1), synth.config3(30_000, seed=5)]
bad = []

def worker(tag, pcs, n):
    ctxs = []
    for pc in pcs:
        c = _capi.Context(0); c.set_complex(pc); ctxs.append((c, dict(c.run_launch(5.0, 0.1, False, 6.0))))
    for k in range(n):
        c, want = ctxs[k % len(ctxs)]
        got = dict(c.run_launch(5.0, 0.1, False, 6.0))
        if got != want:
            bad.append((tag, k, got, want))
    for c, _ in ctxs:
        c.close()

worker('single', cases, args.passes)
ths = [threading.Thread(target=worker, args=('t%d' % t, cases[t:] + cases[:t], args.passes // 2)) for t in range(3)]
for t in ths: t.start()
for t in ths: t.join()
print(json.dumps({'passes': args.passes + 3 * (args.passes // 2), 'mismatches': len(bad), 'first': [str(b)[:300] for b in bad[:3]]}))
sys.exit(1 if bad else 0)
