#!/usr/bin/env python3
"""Worker of tests/test_sharding.py::test_two_ranks_over_rccl_when_two_gpus_are_visible (launched by torch.distributed.run
with two ranks, one GPU each).  The data path is the library's own RCCL communicator (arp_comm_*); the rendezvous — the
128-byte unique id, the gathered results for the check — is arpeggio_amd.rendezvous (TCP through rank 0): no PyTorch in a
process that holds the library's RCCL."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    import pickle
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    local = int(os.environ.get('LOCAL_RANK', rank))
    from arpeggio_amd import _capi, sharding, synth
    from arpeggio_amd.rendezvous import TcpRendezvous
    rdzv = TcpRendezvous(rank, world)
    uid = np.frombuffer(rdzv.broadcast(_capi.Context.comm_unique_id().tobytes() if rank == 0 else None), np.uint8).copy()
    host_transport = rdzv
    full = synth.slab_config(30_000, world, seed=4)
    sel = (full.res_id % 9 == 2).astype(np.uint8)
    comm_ctx = _capi.Context(local)
    comm_ctx.comm_init(rank, world, uid)                  # one communicator per process, reused by every mode below
    for mode, assembly in (('whole', 'device'), ('staged', 'device'), ('whole', 'host'), ('staged', 'host')):
        ctx = comm_ctx
        if assembly == 'device':    # halo records cut out, exchanged (RCCL on device pointers) and merged in HBM
            shard = sharding.make_shard_device(ctx, full, rank, world, sel=None if mode == 'whole' else sel, whole_structure=(mode == 'whole'))
        else:                       # the same through host buffers (gloo)
            shard = sharding.make_shard_distributed(full, rank, world, host_transport, sel=None if mode == 'whole' else sel)
            sharding.upload_shard(ctx, shard, whole_structure=(mode == 'whole'))
        if mode == 'whole':
            counts = sharding.run_shard_whole_structure(ctx)
        else:
            ex = sharding.DeviceExchange(ctx, shard)      # selection_plus bits and residue sets over RCCL, on the context's stream
            counts = sharding.run_shard_device(ctx, ex)
        mine = ctx.atom_contacts_fetch(counts['atom_atom'], sort=False)
        key = np.sort(mine['i'].astype(np.int64) * full.n_atoms + mine['j'])
        gathered = [pickle.loads(b) for b in rdzv.allgather(pickle.dumps(key))]
        if rank == 0:
            union = np.sort(np.concatenate(gathered))
            one = _capi.Context(local)
            one.set_complex(full)
            if mode == 'staged':
                one.set_selection(sel)
            c1 = one.run_launch()
            ref = one.atom_contacts_fetch(c1['atom_atom'])
            want = ref['i'].astype(np.int64) * full.n_atoms + ref['j']
            assert np.array_equal(union, want), (mode, assembly, len(union), len(want))
            one.close()
        rdzv.barrier()
    comm_ctx.comm_destroy()
    comm_ctx.close()
    if rank == 0:
        print('RCCL_TWO_RANKS_OK')
    rdzv.close()


if __name__ == '__main__':
    main()
