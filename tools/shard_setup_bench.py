#!/usr/bin/env python3
"""Set-up cost of one shard (middle slab of three: two neighbours) on ONE GPU: host assembly + classic upload against
the device assembly (arp_shard_*), the neighbours' face buffers standing in for what RCCL would deliver.

    python tools/shard_setup_bench.py [--atoms 250000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--atoms', type=int, default=250_000, help='atoms per slab')
    ap.add_argument('--repeat', type=int, default=3)
    args = ap.parse_args()
    from arpeggio_amd import _capi, sharding, synth
    world, rank = 3, 1
    full = synth.slab_config(args.atoms, world, seed=4)
    ctxs = [_capi.Context(0) for _ in range(world)]
    out = {'atoms_per_slab': args.atoms}
    for rep in range(args.repeat):
        # ---- host path: records packed, (exchanged), merged in NumPy, uploaded array by array
        t0 = time.perf_counter()
        sh = sharding.make_shard_local(full, rank, world, None)
        t1 = time.perf_counter()
        sharding.upload_shard(ctxs[rank], sh, whole_structure=True)
        t2 = time.perf_counter()
        n_host = sharding.run_shard_whole_structure(ctxs[rank], sh=sh)
        # ---- device path
        for r in (0, 2):
            nb = sharding.shard_home_to_device(ctxs[r], full, r, world)
            if r == 0:
                left = nb[0][+1]
            else:
                right = nb[0][-1]
        t3 = time.perf_counter()
        halo = sharding.halo_width()
        edges, a_own, r_own, m_own = sharding._partition(full, world, halo)
        home = sharding.pack_records(full, np.nonzero(a_own == rank)[0], np.nonzero(r_own == rank)[0], np.nonzero(m_own == rank)[0], None)
        buf = _capi.pack_records_buffer(home)
        t4 = time.perf_counter()
        ctxs[rank].shard_set_home(buf)
        t5 = time.perf_counter()
        f = {-1: ctxs[rank].shard_pack_face(0, -np.inf, edges[rank] + halo), +1: ctxs[rank].shard_pack_face(1, edges[rank + 1] - halo, np.inf)}
        t6 = time.perf_counter()
        ctxs[rank].shard_assemble(left, right, full.n_residues)
        t7 = time.perf_counter()
        lay = ctxs[rank].shard_layout()
        ctxs[rank].set_whole_structure(True)
        t8 = time.perf_counter()
        n_dev = ctxs[rank].run_launch()
        assert n_dev == n_host, (n_dev, n_host)
        out = {'atoms_per_slab': args.atoms, 'local_atoms': int(ctxs[rank].n), 'halo_atoms': int((lay['origin'] != 0).sum()),
               'face_bytes': [f[-1][1], f[+1][1]],
               'host_path_ms': {'pack_and_merge_numpy': round((t1 - t0) * 1e3, 2), 'upload_setters': round((t2 - t1) * 1e3, 2)},
               'device_path_ms': {'pack_home_records_numpy': round((t4 - t3) * 1e3, 2), 'upload_home': round((t5 - t4) * 1e3, 2),
                                  'cut_two_faces': round((t6 - t5) * 1e3, 2), 'merge_home_and_halos': round((t7 - t6) * 1e3, 2),
                                  'id_maps_to_host': round((t8 - t7) * 1e3, 2)},
               'contacts': n_dev['atom_atom']}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
