#!/usr/bin/env python3
"""Fresh structures end to end on K contexts / host threads (GPU box only): per-structure wall time and where a thread's time goes.
    python tools/inflight_probe.py [--atoms N] [--contexts 1 2 3 4]"""
import argparse, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from arpeggio_amd import synth, _capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--atoms', type=int, default=100_000)
ap.add_argument('--contexts', type=int, nargs='+', default=[1, 2, 3, 4])
ap.add_argument('--per-thread', type=int, default=150)
args = ap.parse_args()
blobs = [_capi.pack_blob(synth.config3(args.atoms, seed=3 + k)) for k in range(4)]


def make(cx):
    cx.set_blob(blobs[0])
    cnt = cx.run_launch(5.0, 0.1, False, 6.0)
    return _capi.pinned_empty(int(cnt['atom_atom'] * 1.25) * 16 + (4 << 20), np.uint8)


for K in args.contexts:
    ctxs = [_capi.Context(0) for _ in range(K)]
    bufs = [make(c) for c in ctxs]
    gate = threading.Barrier(K + 1)
    phase = [[0.0, 0.0, 0.0] for _ in range(K)]

    def worker(k):
        cx, bf = ctxs[k], bufs[k]
        for j in range(4):
            cx.set_blob(blobs[(k + j) % 4]); cx.run_launch(5.0, 0.1, False, 6.0); cx.fetch_packed(bf)
        gate.wait()
        p = phase[k]
        for j in range(args.per_thread):
            t0 = time.perf_counter(); cx.set_blob(blobs[(k + j) % 4])
            t1 = time.perf_counter(); cx.run_launch(5.0, 0.1, False, 6.0)
            t2 = time.perf_counter(); cx.fetch_packed(bf)
            t3 = time.perf_counter()
            p[0] += t1 - t0; p[1] += t2 - t1; p[2] += t3 - t2

    th = [threading.Thread(target=worker, args=(k,)) for k in range(K)]
    for t in th: t.start()
    gate.wait(); t0 = time.perf_counter()
    for t in th: t.join()
    wall = time.perf_counter() - t0
    n = args.per_thread
    print(json.dumps({'contexts': K, 'ms_per_structure': round(wall / (n * K) * 1e3, 4),
                      'per_thread_ms': {'upload_validate': round(sum(p[0] for p in phase) / (n * K) * 1e3, 4),
                                        'first_pass': round(sum(p[1] for p in phase) / (n * K) * 1e3, 4),
                                        'sort_fetch_packed': round(sum(p[2] for p in phase) / (n * K) * 1e3, 4)}}), flush=True)
    for c in ctxs: c.close()
