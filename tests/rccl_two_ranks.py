#!/usr/bin/env python3
"""Worker of tests/test_sharding.py::test_two_ranks_over_rccl_when_two_gpus_are_visible (launched by torch.distributed.run
with two ranks, one GPU each, backend nccl = RCCL)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    local = int(os.environ.get('LOCAL_RANK', rank))
    torch.cuda.set_device(local)          # torch touches the device BEFORE the first arpeggio context
    dev = torch.device('cuda', local)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    from arpeggio_amd import _capi, sharding, synth
    full = synth.slab_config(30_000, world, seed=4)
    sel = (full.res_id % 9 == 2).astype(np.uint8)
    for mode, assembly in (('whole', 'device'), ('staged', 'device'), ('whole', 'host'), ('staged', 'host')):
        ctx = _capi.Context(local)
        if assembly == 'device':    # halo records cut out, exchanged (RCCL on device pointers) and merged in HBM
            shard = sharding.make_shard_device(ctx, full, rank, world, dist, dev, sel=None if mode == 'whole' else sel,
                                               whole_structure=(mode == 'whole'))
        else:                       # the same through host buffers
            shard = sharding.make_shard_distributed(full, rank, world, dist, device=dev, sel=None if mode == 'whole' else sel)
            sharding.upload_shard(ctx, shard, whole_structure=(mode == 'whole'))
        if mode == 'whole':
            counts = sharding.run_shard_whole_structure(ctx)
        else:
            ex = sharding.DeviceExchange(ctx, shard, dist, dev)
            counts = sharding.run_shard_device(ctx, ex)
        mine = ctx.atom_contacts_fetch(counts['atom_atom'], sort=False)
        key = mine['i'].astype(np.int64) * full.n_atoms + mine['j']
        t = torch.from_numpy(np.sort(key)).to(dev)
        n_all = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(n_all, torch.tensor([t.numel()], dtype=torch.int64, device=dev))
        cap = int(max(x.item() for x in n_all))
        pad = torch.full((cap,), -1, dtype=torch.int64, device=dev)
        pad[:t.numel()] = t
        gathered = [torch.empty(cap, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(gathered, pad)
        if rank == 0:
            union = np.sort(np.concatenate([g.cpu().numpy()[:int(n.item())] for g, n in zip(gathered, n_all)]))
            one = _capi.Context(local)
            one.set_complex(full)
            if mode == 'staged':
                one.set_selection(sel)
            c1 = one.run_launch()
            ref = one.atom_contacts_fetch(c1['atom_atom'])
            want = ref['i'].astype(np.int64) * full.n_atoms + ref['j']
            assert np.array_equal(union, want), (mode, assembly, len(union), len(want))
            one.close()
        ctx.close()
        dist.barrier()
    if rank == 0:
        print('RCCL_TWO_RANKS_OK')
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
