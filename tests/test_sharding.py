"""Slab sharding on CPU: partition + halo exchange + ownership, world_size 2 and 3 over gloo.

The compute backend in these tests is the oracle (allowed in tests/): each rank evaluates its
shard, keeps what it owns, and the union must equal the unsharded result with global ids.
"""
import os
import socket

import numpy as np
import pytest

import oracle
from arpeggio_amd import sharding, synth
from helpers import GlooTransport


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _eval_shard(sh, masks):
    """Oracle on one shard + ownership rule; ids mapped to global."""
    oc = oracle.OracleComplex(sh.pc, in_sel=masks['sel'], in_plus=masks['plus'])
    oc.ring_sel[:] = masks['ring_sel']; oc.ring_plus[:] = masks['ring_plus']
    oc.amide_sel[:] = masks['amide_sel']; oc.amide_plus[:] = masks['amide_plus']
    # the shard's single-bond neighbours are inline coordinates; the oracle wants an index: append ghost atoms
    r = oc.atom_contacts(use_grid=False) if not sh.sb_has.any() else _contacts_with_inline_sb(sh, masks)
    own = sh.is_home[r['i']] == 1
    aa = np.stack([sh.global_id[r['i'][own]], sh.global_id[r['j'][own]], r['sift'][own].astype(np.int64),
                   r['ctype'][own].astype(np.int64), r['dist'][own].view(np.uint32).astype(np.int64)], axis=1)
    pp = oc.plane_plane()
    lo = np.minimum(pp['bgn'], pp['end'])
    ownp = sh.ring_home[lo] == 1
    ppo = np.stack([sh.ring_gid[pp['bgn'][ownp]], sh.ring_gid[pp['end'][ownp]], pp['type1'][ownp].astype(np.int64),
                    pp['type2'][ownp].astype(np.int64), pp['ctype'][ownp].astype(np.int64)], axis=1)
    ap = oc.atom_plane()
    owna = sh.ring_home[ap['ring']] == 1
    apo = np.stack([sh.ring_gid[ap['ring'][owna]], sh.global_id[ap['atom'][owna]], ap['mask'][owna].astype(np.int64),
                    ap['ctype'][owna].astype(np.int64)], axis=1)
    gg = oc.group_group()
    owng = sh.amide_home[gg['bgn']] == 1
    ggo = np.stack([sh.amide_gid[gg['bgn'][owng]], sh.amide_gid[gg['end'][owng]], gg['ctype'][owng].astype(np.int64)], axis=1)
    gp = oc.group_plane()
    ownq = sh.amide_home[gp['amide']] == 1
    gpo = np.stack([sh.amide_gid[gp['amide'][ownq]], sh.ring_gid[gp['ring'][ownq]], gp['ctype'][ownq].astype(np.int64)], axis=1)
    return dict(aa=aa, pp=ppo, ap=apo, gg=ggo, gp=gpo)


def _contacts_with_inline_sb(sh, masks):
    """Give the oracle an index for every inline single-bond neighbour by appending inert ghost atoms."""
    from arpeggio_amd.core.packed import PackedComplex
    pc = sh.pc
    n = pc.n_atoms
    need = np.nonzero(sh.sb_has)[0]
    g = need.size
    sb = pc.sb_nbr.copy()
    sb[need] = n + np.arange(g)
    ghost_res = pc.n_residues + np.arange(g)
    pc2 = PackedComplex(
        xyz=np.concatenate([pc.xyz, sh.sb_xyz[need]]), vdw=np.concatenate([pc.vdw, np.ones(g)]), cov=np.concatenate([pc.cov, np.ones(g)]),
        type_mask=np.concatenate([pc.type_mask, np.zeros(g, np.uint16)]), flags=np.concatenate([pc.flags, np.zeros(g, np.uint16)]),
        res_id=np.concatenate([pc.res_id, ghost_res]), res_flags=np.concatenate([pc.res_flags, np.zeros(g, np.uint8)]),
        res_prev=np.concatenate([pc.res_prev, np.full(g, -1)]), res_next=np.concatenate([pc.res_next, np.full(g, -1)]),
        bond_off=np.concatenate([pc.bond_off, np.full(g, pc.bond_off[-1])]), bond_idx=pc.bond_idx,
        h_off=np.concatenate([pc.h_off, np.full(g, pc.h_off[-1])]), h_xyz=pc.h_xyz, sb_nbr=np.concatenate([sb, np.full(g, -1)]),
        ring_center=pc.ring_center, ring_normal=pc.ring_normal, ring_res=pc.ring_res,
        amide_center=pc.amide_center, amide_normal=pc.amide_normal, amide_res=pc.amide_res)
    plus = np.concatenate([masks['plus'], np.zeros(g, np.uint8)])     # ghosts are outside the selection_plus tree
    sel = np.concatenate([masks['sel'], np.zeros(g, np.uint8)])
    oc = oracle.OracleComplex(pc2, in_sel=sel, in_plus=plus)
    return oc.atom_contacts(use_grid=False)


def _reference(full, sel):
    oc = oracle.OracleComplex(full)
    oc.make_selection(sel)
    r = oc.atom_contacts()
    aa = np.stack([r['i'], r['j'], r['sift'].astype(np.int64), r['ctype'].astype(np.int64),
                   r['dist'].view(np.uint32).astype(np.int64)], axis=1)
    pp, ap, gg, gp = oc.plane_plane(), oc.atom_plane(), oc.group_group(), oc.group_plane()
    return dict(
        aa=aa,
        pp=np.stack([pp['bgn'], pp['end'], pp['type1'].astype(np.int64), pp['type2'].astype(np.int64), pp['ctype'].astype(np.int64)], axis=1),
        ap=np.stack([ap['ring'], ap['atom'], ap['mask'].astype(np.int64), ap['ctype'].astype(np.int64)], axis=1),
        gg=np.stack([gg['bgn'], gg['end'], gg['ctype'].astype(np.int64)], axis=1),
        gp=np.stack([gp['amide'], gp['ring'], gp['ctype'].astype(np.int64)], axis=1)), oc


def _canon(a):
    a = np.asarray(a, np.int64).reshape(-1, a.shape[1] if a.ndim == 2 else 1)
    return a[np.lexsort(a.T[::-1])] if a.size else a


def _workload():
    full = synth.slab_config(2500, 3, seed=8)      # 7500 atoms, 3 slabs of ~58 A
    rng = np.random.default_rng(3)
    sel = np.zeros(full.n_atoms, np.uint8)
    sel[np.isin(full.res_id, rng.choice(full.n_residues, full.n_residues // 15, replace=False))] = 1
    return full, sel


@pytest.mark.parametrize('world', [2, 3])
def test_local_shards_union_equals_unsharded(world):
    """No communication: shards cut from global knowledge + global selection masks."""
    full, sel = _workload()
    exp, oc = _reference(full, sel)
    got = {k: [] for k in exp}
    for rank in range(world):
        sh = sharding.make_shard_local(full, rank, world, sel)
        assert np.all(np.diff(sh.global_id) > 0) and sh.is_home.sum() > 0 and (sh.is_home == 0).sum() > 0
        masks = dict(sel=sel[sh.global_id], plus=oc.in_plus[sh.global_id], ring_sel=oc.ring_sel[sh.ring_gid],
                     ring_plus=oc.ring_plus[sh.ring_gid], amide_sel=oc.amide_sel[sh.amide_gid], amide_plus=oc.amide_plus[sh.amide_gid])
        # the shard-local expansion + combine (single process: no exchange) reproduces the masks of home atoms
        loc = oracle.OracleComplex(sh.pc)
        plus_local = loc.make_selection(sh.sel, use_grid=False)
        hm = sh.is_home == 1
        assert np.array_equal(plus_local[hm], masks['plus'][hm])
        out = _eval_shard(sh, masks)
        for k in got:
            got[k].append(out[k])
    for k in exp:
        g = np.concatenate(got[k], axis=0)
        assert np.array_equal(_canon(g), _canon(exp[k])), k
    assert len(exp['aa']) > 1000 and len(exp['pp']) > 0 and len(exp['ap']) > 0


@pytest.mark.parametrize('world', [2, 3])
def test_whole_structure_shards_need_no_selection_exchange(world):
    """No selection (I:1395) = everything selected: each shard can fill its masks by itself — selection_plus is every
    local atom and every ring / amide with a residue is in both sets (arp_set_whole_structure) — and the union of what
    the ranks own is the unsharded whole-structure result.  This is the path bench.py --gpus N times by default."""
    full, _ = _workload()
    exp, oc = _reference(full, None)
    assert oc.in_plus.all() and oc.ring_plus[full.ring_res >= 0].all()
    got = {k: [] for k in exp}
    for rank in range(world):
        sh = sharding.make_shard_local(full, rank, world, None)
        assert sh.sel.all()
        ones = lambda n: np.ones(n, np.uint8)
        masks = dict(sel=ones(sh.pc.n_atoms), plus=ones(sh.pc.n_atoms),
                     ring_sel=(sh.pc.ring_res >= 0).astype(np.uint8), ring_plus=(sh.pc.ring_res >= 0).astype(np.uint8),
                     amide_sel=(sh.pc.amide_res >= 0).astype(np.uint8), amide_plus=(sh.pc.amide_res >= 0).astype(np.uint8))
        out = _eval_shard(sh, masks)
        for k in got:
            got[k].append(out[k])
    for k in exp:
        assert np.array_equal(_canon(np.concatenate(got[k], axis=0)), _canon(exp[k])), k
    assert len(exp['aa']) > 10_000 and len(exp['pp']) > 0 and len(exp['ap']) > 0


class _OracleStageContext:
    """Stand-in for the GPU context in the staged (run_stage) protocol: same three stages, NumPy buffers, the oracle
    as compute.  Exercises sharding.DeviceExchange / run_shard_device over gloo."""
    BUF_PLUS, BUF_RES_SETS = 0, 1

    def __init__(self, sh):
        self.sh = sh
        self.plus = np.zeros(sh.pc.n_atoms, np.uint8)
        self.res = np.zeros(2 * sh.n_res_global, np.uint8)

    def host_buffer(self, which):
        return self.plus if which == self.BUF_PLUS else self.res

    def run_stage(self, stage, *a):
        sh = self.sh
        if stage == 0:
            self.plus[:] = oracle.OracleComplex(sh.pc).make_selection(sh.sel, use_grid=False)
        elif stage == 1:
            self.res[:] = 0
            self.res[sh.pc.res_id[sh.sel == 1]] = 1
            self.res[sh.n_res_global + sh.pc.res_id[self.plus == 1]] = 1
        else:
            nr = sh.n_res_global
            rs, rp = self.res[:nr], self.res[nr:]
            pick = lambda r, m: np.where(r >= 0, m[np.maximum(r, 0)], 0).astype(np.uint8)
            self.masks = dict(sel=sh.sel, plus=self.plus.copy(), ring_sel=pick(sh.pc.ring_res, rs), ring_plus=pick(sh.pc.ring_res, rp),
                              amide_sel=pick(sh.pc.amide_res, rs), amide_plus=pick(sh.pc.amide_res, rp))
            return _eval_shard(sh, self.masks)


def _worker_staged(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        full, sel = _workload()
        tr = GlooTransport(dist, rank, world)
        sh = sharding.make_shard_distributed(full, rank, world, tr, sel=sel)
        ctx = _OracleStageContext(sh)
        ex = sharding.DeviceExchange(ctx, sh, tr)
        out = sharding.run_shard_device(ctx, ex)
        gathered = [None] * world
        dist.all_gather_object(gathered, (True, out, ctx.masks['plus'], sh.global_id))
        if rank == 0:
            q.put(gathered)
    finally:
        dist.destroy_process_group()


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        full, sel = _workload()
        tr = GlooTransport(dist, rank, world)
        sh = sharding.make_shard_distributed(full, rank, world, tr, sel=sel)
        ref = sharding.make_shard_local(full, rank, world, sel)
        same = all(np.array_equal(getattr(sh.pc, k), getattr(ref.pc, k)) for k in
                   ('xyz', 'vdw', 'type_mask', 'flags', 'res_id', 'res_prev', 'res_next', 'bond_off', 'bond_idx', 'h_off', 'h_xyz',
                    'ring_center', 'ring_res', 'amide_center', 'amide_res'))
        same = same and np.array_equal(sh.global_id, ref.global_id) and np.array_equal(sh.is_home, ref.is_home) \
            and np.array_equal(sh.sb_xyz, ref.sb_xyz) and np.array_equal(sh.origin, ref.origin)
        loc = oracle.OracleComplex(sh.pc)
        plus_local = loc.make_selection(sh.sel, use_grid=False)
        masks = sharding.combine_selection(sh, plus_local, tr)   # plus-bit halo exchange + residue all-reduce
        out = _eval_shard(sh, masks)
        gathered = [None] * world
        dist.all_gather_object(gathered, (same, out, masks['plus'], sh.global_id))
        if rank == 0:
            q.put(gathered)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,worker', [(2, '_worker'), (3, '_worker'), (2, '_worker_staged'), (3, '_worker_staged')])
def test_gloo_halo_exchange_and_selection_combine(world, worker):
    """One process per rank over gloo: exchanged halos == global-knowledge halos, combined masks == global
    _make_selection, union of owned results == unsharded result."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=globals()[worker], args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full, sel = _workload()
    exp, oc = _reference(full, sel)
    assert all(g[0] for g in gathered), 'exchanged shard differs from the global-knowledge shard'
    for _, _, plus, gid in gathered:
        assert np.array_equal(plus, oc.in_plus[gid]), 'combined selection_plus differs from the global one'
    for k in exp:
        g = np.concatenate([x[1][k] for x in gathered], axis=0)
        assert np.array_equal(_canon(g), _canon(exp[k])), k


@pytest.mark.gpu
def test_two_ranks_over_rccl_when_two_gpus_are_visible():
    """The real thing — two processes, two GPUs, RCCL over xGMI behind the C ABI (arp_comm_init; the 128-byte id travels
    through arpeggio_amd.rendezvous, a few TCP messages): halo records exchanged with grouped ncclSend / ncclRecv, the
    three-stage pass with the selection_plus bits and the residue sets exchanged on the context's stream, and the union
    of what the two ranks own equal to the single-GPU result.  Skipped on a one-GPU box."""
    import subprocess
    import sys
    from arpeggio_amd import _capi      # (not torch: a second HIP / RCCL runtime in this process is what the workers are for)
    if _capi.device_count() < 2:
        pytest.skip('needs two GPUs')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', '29533', os.path.join(root, 'tests', 'rccl_two_ranks.py')]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert 'RCCL_TWO_RANKS_OK' in out.stdout


def _rendezvous_worker(rank, world, port, q):
    from arpeggio_amd.rendezvous import TcpRendezvous
    r = TcpRendezvous(rank, world, '127.0.0.1', port)
    b = r.broadcast(b'id-of-rank-0' if rank == 0 else None)
    r.barrier()
    s = r.allreduce(np.array([rank, 1.0]), 'sum')
    m = r.allreduce_max(np.array([rank * 2], np.uint8))
    pay = {}
    if rank > 0:
        pay[-1] = np.full(3 + rank, rank, np.uint8)
    if rank < world - 1:
        pay[+1] = np.full(5 + rank, 100 + rank, np.uint8)
    got = r.exchange(pay)
    q.put((rank, b, s.tolist(), m.tolist(), {k: (int(v.size), int(v[0])) for k, v in got.items()}))
    r.close()


def test_tcp_rendezvous_three_ranks():
    """arpeggio_amd.rendezvous — what bench.py --gpus N and the two-rank RCCL worker use instead of torch.distributed:
    broadcast, barrier, all-reduce and the neighbour exchange of the sharding transport, three processes."""
    import multiprocessing as mp
    ctxm = mp.get_context('spawn')
    q, W = ctxm.Queue(), 3
    port = 20000 + (os.getpid() * 7) % 20000
    ps = [ctxm.Process(target=_rendezvous_worker, args=(k, W, port, q)) for k in range(W)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(W))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, b, s, m, ex in got:
        assert b == b'id-of-rank-0' and s == [3.0, 3.0] and m == [4]
    assert got[0][4] == {1: (4, 1)}
    assert got[1][4] == {-1: (5, 100), 1: (5, 2)}
    assert got[2][4] == {-1: (6, 101)}


def _launch_bench_dry(nproc, port, extra_env=None, timeout=240):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    env.update(extra_env or {})
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={nproc}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', str(nproc), '--steps', '5', '--warmup', '1', '--dry-run']
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=root)


def _dry_line(out):
    import json
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{') and '"dry_run"' in ln]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-2000:]
    return json.loads(lines[0])


def test_bench_gpus_2_as_the_driver_launches_it_up_to_the_first_hip_call():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2
    ... --dry-run` on CPU: everything of the N > 1 launch before its first HIP call — RANK / LOCAL_RANK / WORLD_SIZE from the
    launcher, one rank per device (LOCAL_RANK -> device), the TCP rendezvous beside the launcher's own store on MASTER_PORT,
    the 128-byte communicator id from rank 0 to everybody, a reduction and the barriers; twice in a row on the SAME port (the
    driver runs N = 1, 2, 4, 8 back to back), once with four ranks and once with eight.  The per-GPU workload of the line is the N = 1 one."""
    import json
    import subprocess
    import sys
    port = 21000 + (os.getpid() * 13) % 20000
    first = _dry_line(_launch_bench_dry(2, port))
    again = _dry_line(_launch_bench_dry(2, port))            # the port (and the rendezvous port derived from it) is free again
    for line in (first, again):
        ranks = sorted(line['ranks'], key=lambda r: r['rank'])
        assert line['n_gpus'] == 2 and [r['rank'] for r in ranks] == [0, 1]
        assert [r['device'] for r in ranks] == [r['local_rank'] for r in ranks] == [0, 1]      # one rank per GPU
        assert len({r['uid_crc32'] for r in ranks}) == 1 and len({r['pid'] for r in ranks}) == 2
        assert line['allreduce_sum'] == 3.0 and all(r['master'] == f'127.0.0.1:{port}' for r in ranks)
    assert first['ranks'][0]['uid_crc32'] != again['ranks'][0]['uid_crc32']                   # (a fresh id per launch)
    four = _dry_line(_launch_bench_dry(4, port + 1))
    assert sorted(r['device'] for r in four['ranks']) == [0, 1, 2, 3] and four['allreduce_sum'] == 10.0
    eight = _dry_line(_launch_bench_dry(8, port + 2))         # the size of the driver's node
    assert sorted(r['device'] for r in eight['ranks']) == list(range(8)) and eight['allreduce_sum'] == 36.0
    assert len({r['uid_crc32'] for r in eight['ranks']}) == 1 and eight['config'] == four['config']
    # the same per-GPU workload at N = 1 (what SCALE divides by) as at N > 1
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    one = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--dry-run'], capture_output=True, text=True, timeout=120, cwd=root)
    one_line = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith('{')][-1])
    assert one_line['config'] == first['config'] == four['config'] and one_line['config']['atoms_per_gpu'] == 100_000
    # the strong-scaling leg (BASELINE configs[3]: ONE 2 M-atom structure cut into N slabs) is part of the line at EVERY N, N = 1 included
    for n, line in ((1, one_line), (2, first), (4, four), (8, eight)):
        leg = line['legs']['config4_strong']
        assert leg['n_gpus'] == n and leg['atoms_per_gpu'] * n == 2_000_000 and leg['scaling'] == 'strong' and '2000000 atoms' in leg['workload']


def test_bench_rendezvous_times_out_when_a_rank_is_missing():
    """A rank whose peers never arrive gives up after ARP_RDZV_TIMEOUT with a message that says who is missing (rank 0: how many
    arrived; another rank: that rank 0 was not found), instead of hanging the launch."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 23000 + (os.getpid() * 17) % 20000
    for rank, needle in ((0, '1 of 2 ranks arrived'), (1, 'rank 0 not found')):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port + rank),
                   ARP_RDZV_TIMEOUT='3')
        t0 = time.time()
        out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--dry-run'], env=env, capture_output=True,
                             text=True, timeout=120, cwd=root)
        assert out.returncode != 0 and needle in out.stderr, out.stderr[-1500:]
        assert time.time() - t0 < 60


def test_rendezvous_ignores_strangers():
    """A connection that never says hello, one that announces a huge message and one with another launch's token do not claim a
    rank and do not use up the deadline."""
    import multiprocessing as mp
    import socket
    import struct
    import time
    from arpeggio_amd import rendezvous
    port = 25000 + (os.getpid() * 19) % 20000
    ctxm = mp.get_context('spawn')
    q = ctxm.Queue()
    ps = [ctxm.Process(target=_rendezvous_worker, args=(k, 2, port, q)) for k in range(1)]      # rank 0 only, for now
    ps[0].start()
    time.sleep(1.0)
    silent = socket.create_connection(('127.0.0.1', port), timeout=5)                             # says nothing
    greedy = socket.create_connection(('127.0.0.1', port), timeout=5)
    greedy.sendall(struct.pack('<Q', 1 << 40))                                                      # "a terabyte follows"
    wrong = socket.create_connection(('127.0.0.1', port), timeout=5)
    hello = rendezvous._MAGIC + struct.pack('<ii', 1, 2) + b'token-of-another-launch'
    wrong.sendall(struct.pack('<Q', len(hello)) + hello)
    p1 = ctxm.Process(target=_rendezvous_worker, args=(1, 2, port, q))
    t0 = time.time()
    p1.start()
    got = sorted(q.get(timeout=60) for _ in range(2))
    assert time.time() - t0 < 30
    for s_ in (silent, greedy, wrong):
        s_.close()
    for p in ps + [p1]:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert [g[0] for g in got] == [0, 1] and all(g[1] == b'id-of-rank-0' for g in got)


@pytest.mark.parametrize('n_per_slab,world,box_slabs', [(5000, 3, None), (3000, 2, None), (3000, 2, 4), (6000, 1, 4)])
def test_a_rank_generates_its_own_slab(n_per_slab, world, box_slabs):
    """synth.slab_home_records: what a rank owns of slab_config(...) WITHOUT the whole structure in its memory — every field of
    the records sharding.pack_records cuts out of the whole structure, same dtypes, same bytes; and the record buffer
    (the C ABI's packer on the records alone) equals the one the native packer makes from the whole structure."""
    from arpeggio_amd import _capi
    full = synth.slab_config(n_per_slab, world, seed=4, box_slabs=box_slabs)
    if box_slabs:      # strong scaling (bench.py's config4_strong): ONE structure whatever the number of slabs it is cut into
        same = synth.slab_config(n_per_slab * world // box_slabs, box_slabs, seed=4)
        assert np.array_equal(same.xyz, full.xyz) and np.array_equal(same.bond_idx, full.bond_idx) and np.array_equal(same.h_xyz, full.h_xyz)
    edges, a_own, r_own, m_own = sharding._partition(full, world, sharding.halo_width())
    for rank in range(world):
        ids = [np.nonzero(o == rank)[0] for o in (a_own, r_own, m_own)]
        want = sharding.pack_records(full, *ids)
        got, book = synth.slab_home_records(n_per_slab, world, rank, seed=4, box_slabs=box_slabs)
        assert np.array_equal(book['edges'], edges) and book['n_res_global'] == full.n_residues and book['n_atoms_global'] == full.n_atoms
        assert np.array_equal(book['home_x'], full.xyz[ids[0], 0].astype(np.float64))
        assert set(got) == set(want)
        for k in want:
            w, g = np.asarray(want[k]), np.asarray(got[k])
            assert w.dtype == g.dtype and w.shape == g.shape, (rank, k, w.dtype, g.dtype, w.shape, g.shape)
            assert np.array_equal(np.ascontiguousarray(w).view(np.uint8), np.ascontiguousarray(g).view(np.uint8)), (rank, k)
        a = _capi.pack_records_buffer(got, pinned=False)
        b = _capi.pack_records_native(full, *ids, pinned=False)
        assert a.size == b.size and np.array_equal(a, b), rank
