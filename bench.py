#!/usr/bin/env python3
"""bench.py — evaluated atom-pairs/s of the run_arpeggio hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--atoms A]

One "step" = one pass of the whole hot path (InteractionComplex.run_arpeggio,
interactions.py:329-347: 6 A selection expansion, 5 A neighbour search + fused 15-flag
SIFt evaluation, ring/amide plane kernels) over the structure resident in HBM; results
stay in HBM.  Unit of work = one candidate atom pair of the 5 A contact search (SURVEY.md
§8d).  N > 1: the box is elongated along x (weak scaling, BASELINE configs[3] family),
sharded into N slabs with a one-cell halo exchanged over RCCL before the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the
dominant kernel (HIP-event time on the context's own stream) and `cpu_baseline` (the C
oracle timed on one host core on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--atoms', type=int, default=100_000, help='atoms per GPU (configs[2]: 100k; configs[3]: 250k x 8)')
    ap.add_argument('--cutoff', type=float, default=5.0)
    ap.add_argument('--vdw-comp', type=float, default=0.1)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--workload', choices=('config3', 'standin'), default='config3',
                    help="config3: BASELINE configs[2] (the headline); standin: the 1tqn_h stand-in of configs[1] (5.9 k atoms incl. "
                         "explicit hydrogens, whole structure) on one GPU beside the CPU restatement")
    ap.add_argument('--inflight', type=int, default=3, help='contexts (host threads) of the extra several-structures-in-flight measurement; 1 = skip')
    ap.add_argument('--staged-exchange', action='store_true',
                    help='N > 1: run the general three-stage protocol (selection_plus bits over P2P + residue-set all-reduce '
                         'every step) although the whole-structure selection of the benchmark does not need it')
    ap.add_argument('--cpu-sample-atoms', type=int, default=100_000)
    return ap.parse_args()


def algorithmic_bytes(kernel, n_binned, ncell, n_pairs):
    """Compulsory HBM bytes of one launch (DESIGN.md 'Kernels and rooflines')."""
    if kernel == 'search':      # read each sorted record once (xyzm 16 B + aux 16 B), the cell table, write the pair list
        return 32 * n_binned + 4 * (ncell + 1) + 8 * n_pairs
    if kernel == 'mark_search':  # same reads, writes one byte per marked atom
        return 32 * n_binned + 4 * (ncell + 1) + n_binned
    if kernel == 'sift':        # pair list + each 32-byte atom record once + 15-byte output record
        return 8 * n_pairs + 32 * n_binned + 15 * n_pairs
    raise KeyError(kernel)


def main():
    args = parse()
    os.environ.setdefault('OMP_WAIT_POLICY', 'passive')   # (cpu_baseline_all_cores: idle OpenMP threads must not spin inside a CPU quota)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')

    import torch
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (no CPU fallback exists)')
    # ARP_BENCH_SHARE_GPU=1 (debug only): every rank uses GPU 0 and the exchange goes over gloo, so the
    # N > 1 code path can be exercised on a one-GPU box.  Never set by the driver.
    share_gpu = os.environ.get('ARP_BENCH_SHARE_GPU') == '1'
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist, comm_device = None, torch.device('cuda', local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if share_gpu:
            dist.init_process_group('gloo', rank=rank, world_size=world)
            comm_device = None
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=comm_device)

    from arpeggio_amd import synth, _capi

    # ---------------- workload ----------------
    t0 = time.perf_counter()
    halo_ms = 0.0
    if world == 1:
        if args.workload == 'standin':
            pc = synth.proteinlike()
            args.atoms = pc.n_atoms
            workload = (f'1tqn_h stand-in (BASELINE configs[1]: the file is not available): synthetic chain of 480 residues + haem-like '
                        f'ligand + waters, {pc.n_atoms} atoms incl. explicit hydrogens, whole structure, 5 A cutoff')
        else:
            pc = synth.config3(args.atoms, seed=3)
            workload = f'synthetic {args.atoms} random-coordinate atoms, rho=0.05/A^3, 5 A cutoff (BASELINE configs[2])'
        ctx = _capi.Context(local_rank)
        ctx.set_complex(pc)
        n_local_home = pc.n_atoms
    else:
        from arpeggio_amd import sharding
        full = synth.slab_config(args.atoms, world, seed=4)
        workload = (f'synthetic {args.atoms * world} atoms in {world} x-slabs of {args.atoms} '
                    f'(BASELINE configs[3] family), one-cell halo over RCCL')
        halo_note = 'records of the one-cell halo exchanged with grouped isend/irecv (RCCL)'
        try:
            shard = sharding.make_shard_distributed(full, rank, world, dist, device=comm_device)
            halo_ms = shard.halo_ms
        except Exception as exc:   # keep the scaling run alive: every rank holds the full synthetic structure anyway
            shard = sharding.make_shard_local(full, rank, world)
            halo_ms = -1.0
            halo_note = f'halo exchange failed ({exc!r}); shards cut from the locally generated structure'
        ctx = _capi.Context(local_rank)
        sharding.upload_shard(ctx, shard, whole_structure=not args.staged_exchange)
        n_local_home = int(shard.is_home.sum())
        pc = shard.pc
    gen_s = time.perf_counter() - t0

    if world == 1:
        def step():
            return ctx.run_launch(args.cutoff, args.vdw_comp, False, 6.0)
    elif not args.staged_exchange:
        # no selection = whole structure (I:1395): selection_plus and the residue sets are known on every rank without
        # asking the neighbours, so the pass needs no exchange; the halo of records was exchanged over RCCL above
        def step():
            return sharding.run_shard_whole_structure(ctx, args.cutoff, args.vdw_comp, False)
    elif comm_device is not None:
        # device-resident exchange: torch tensors alias the context's buffers, RCCL moves the halo bits and reduces
        # the residue sets between the three stages of the pass
        exchange = sharding.DeviceExchange(ctx, shard, dist, comm_device)

        def step():
            return sharding.run_shard_device(ctx, exchange, args.cutoff, args.vdw_comp, False)
    else:
        def step():   # debug path (gloo, host buffers)
            return sharding.run_shard(ctx, shard, dist, comm_device, args.cutoff, args.vdw_comp, False)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, not steps: the first passes size the result buffers (a pass is re-run when one was too small) and the
    # GPU leaves its idle clocks; the W warm-up steps and the K timed steps below then run on a settled context
    for _ in range(30):
        counts = step()
    for _ in range(args.warmup):
        counts = step()
    sync_all()
    ctx.host_times(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        counts = step()          # blocks until the context stream has drained (one sync per step)
    sync_all()
    elapsed = time.perf_counter() - t0
    host_times = ctx.host_times(reset=True)
    st = ctx.stats()
    # Per-kernel durations: the same K steps once more with every launch bracketed by HIP events on the
    # context's streams (recording ~50 events per step costs ~0.1 ms per step, so it is kept out of `value`).
    ctx.set_profiling(True)
    ctx.kernel_times(reset=True)
    sync_all()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed_profiled = time.perf_counter() - t1
    ktimes = ctx.kernel_times(reset=True)
    ctx.set_profiling(False)

    # Throughput with several structures in flight (world == 1, informational, never `value`): one context per host
    # thread, as INTEGRATION.md prescribes; the passes of different contexts overlap on the GPU (each has its own
    # streams), which hides the per-pass launch gaps and the host round trip that bound the single-stream figure.
    in_flight = None
    if world == 1 and args.inflight > 1:
        try:
            import threading
            ctxs = [ctx] + [_capi.Context(local_rank) for _ in range(args.inflight - 1)]
            for c2 in ctxs[1:]:
                c2.set_complex(pc)
                c2.run_launch(args.cutoff, args.vdw_comp, False, 6.0)
            per_thread = max(args.steps // args.inflight, 1)
            gate = threading.Barrier(args.inflight + 1)

            def worker(cx):
                gate.wait()
                for _ in range(per_thread):
                    cx.run_launch(args.cutoff, args.vdw_comp, False, 6.0)

            threads = [threading.Thread(target=worker, args=(cx,)) for cx in ctxs]
            for t in threads:
                t.start()
            torch.cuda.synchronize()
            gate.wait()
            t2 = time.perf_counter()
            for t in threads:
                t.join()
            torch.cuda.synchronize()
            el2 = time.perf_counter() - t2
            in_flight = {'contexts': args.inflight, 'steps': per_thread * args.inflight,
                         'ms_per_step': round(el2 / (per_thread * args.inflight) * 1e3, 4),
                         'value': round(st['candidates'] * per_thread * args.inflight / el2, 1),
                         'note': 'one arp_ctx per host thread, same structure resident in each; not the headline value'}
            for c2 in ctxs[1:]:
                c2.close()
        except Exception as exc:   # never lose the main line over the extra measurement
            in_flight = {'error': repr(exc)}

    # max over ranks of the elapsed time, sum over ranks of the work
    cand, acc, emitted = st['candidates'], st['accepted'], st['emitted']
    if dist is not None:
        rdev = 'cpu' if comm_device is None else comm_device
        t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        w = torch.tensor([cand, acc, emitted, st['expand_candidates']], dtype=torch.float64, device=rdev)
        dist.all_reduce(w, op=dist.ReduceOp.SUM)
        cand_all, acc_all, emitted_all, exp_all = (float(x) for x in w.tolist())
    else:
        cand_all, acc_all, emitted_all, exp_all = float(cand), float(acc), float(emitted), float(st['expand_candidates'])

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = cand_all * args.steps / elapsed

    # ---------------- roofline of the dominant kernel (rank 0) ----------------
    per_kernel = {k: (v['ms'] / max(v['launches'], 1)) for k, v in ktimes.items() if v['launches']}
    candidates_for_dominant = {k: per_kernel[k] for k in ('search', 'sift', 'mark_search') if k in per_kernel}
    dom = max(candidates_for_dominant, key=candidates_for_dominant.get)
    dom_ms = candidates_for_dominant[dom]
    n_binned = st['binned'] if dom != 'mark_search' else pc.n_atoms
    ncell = st['cells']
    b_alg = algorithmic_bytes(dom, n_binned, ncell, emitted)
    achieved = b_alg / (dom_ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC counters of the last committed rocprofv3 run (profiles/pmc_traffic.json,
    # written by tools/export_profile.py; separate --pmc passes, gfx950 FETCH_SIZE correction applied)
    traffic, traffic_src = None, None
    try:
        pj = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
        if args.atoms == 100_000 and world == 1 and f'k_{dom}' in pj['kernels']:
            traffic = pj['kernels'][f'k_{dom}']['hbm_bytes_per_launch']
            traffic_src = pj['source']
    except (OSError, KeyError, ValueError):
        pass
    roofline = {'kernel': f'k_{dom}', 'bound': 'hbm', 'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBS, 6), 'traffic': traffic, 'traffic_source': traffic_src,
                'algorithmic_bytes_per_launch': int(b_alg), 'avg_launch_ms': round(dom_ms, 5),
                'note': 'VALU-bound geometry kernel; HBM fraction is small by construction (SURVEY 8d)'}

    # the same figure for every big kernel of the pass (informational; `roofline` above is the dominant one)
    roofline_all = {}
    for k, ms in candidates_for_dominant.items():
        nb = st['binned'] if k != 'mark_search' else pc.n_atoms
        b = algorithmic_bytes(k, nb, ncell, emitted)
        roofline_all[f'k_{k}'] = {'avg_launch_ms': round(ms, 5), 'algorithmic_bytes_per_launch': int(b),
                                  'achieved_GBps': round(b / (ms * 1e-3) / 1e9, 2), 'frac': round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)}

    # ---------------- CPU baseline: the C oracle on one host core, bounded sample ----------------
    cpu = None
    if not args.no_cpu_baseline and world == 1:   # (rank 0 at N = 1 only, as the bench contract says)
        import oracle
        ns = min(args.cpu_sample_atoms, args.atoms)
        spc = synth.config3(ns, seed=3) if (world > 1 or (ns != args.atoms and args.workload == 'config3')) else pc
        oc = oracle.OracleComplex(spc)
        passes, cpu_s, cand_cpu = 0, 0.0, 0
        while cpu_s < 10.0 and passes < 200:      # ~10 s of single-core work
            t0 = time.perf_counter()
            oc.make_selection(None)
            r = oc.atom_contacts(args.cutoff, args.vdw_comp, False)
            oc.plane_plane(); oc.group_group(); oc.group_plane()
            cpu_s += time.perf_counter() - t0
            cand_cpu += int(r['stats'][0])
            passes += 1
        cpu = {'value': round(cand_cpu / cpu_s, 1), 'unit': 'candidate atom-pairs/s', 'cores': 1, 'kind': 'port',
               'sample': f'{passes} full run_arpeggio passes of the C oracle (oracle/ref_c.c: grid search_all 6 A + 5 A, per-pair '
                         f'SIFt, ring/amide loops; gcc -O2, 1 thread) on the same {spc.n_atoms}-atom synthetic structure, {cpu_s:.1f} s total, '
                         f'{cpu_s / passes * 1e3:.0f} ms per pass; the O(R*N) brute-force atom-plane loop of the oracle is left out',
               'ms_per_structure': round(cpu_s / passes * 1e3, 2), 'host_cores_visible': os.cpu_count()}

    # the same restatement on ALL host cores (OpenMP over the cell loops): a stronger CPU figure than the reference could
    # ever reach (it is single-threaded Python), reported beside the like-for-like one-core baseline
    cpu_mc = None
    if cpu is not None:
        try:
            threads = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
            try:      # a container's CPU quota counts, not the cores the host shows
                quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
                if quota != 'max':
                    threads = max(1, min(threads, int(int(quota) / int(period))))
            except (OSError, ValueError):
                pass
            t0 = time.perf_counter()
            oracle.pass_openmp(oc, args.cutoff, args.vdw_comp, False, 1)
            one_thread_ms = (time.perf_counter() - t0) * 1e3
            oracle.pass_openmp(oc, args.cutoff, args.vdw_comp, False, threads)          # warm-up (thread pool)
            n_mc, s_mc, cand_mc = 0, 0.0, 0
            while s_mc < 3.0 and n_mc < 200:
                t0 = time.perf_counter()
                r_mc = oracle.pass_openmp(oc, args.cutoff, args.vdw_comp, False, threads)
                s_mc += time.perf_counter() - t0
                cand_mc += r_mc['candidates']
                n_mc += 1
            cpu_mc = {'value': round(cand_mc / s_mc, 1), 'unit': 'candidate atom-pairs/s', 'cores': threads, 'kind': 'port',
                      'ms_per_structure': round(s_mc / n_mc * 1e3, 2), 'same_code_one_thread_ms': round(one_thread_ms, 1),
                      'sample': f'{n_mc} passes of oracle.pass_openmp (6 A + 5 A grid searches and the per-pair evaluation spread over '
                                f'{threads} OpenMP threads = the CPU quota of this container; one search per radius where the checker above '
                                f'counts first and fills second; grid builds serial, ring/amide loops not included), {s_mc:.1f} s total'}
        except Exception as exc:
            cpu_mc = {'error': repr(exc)}

    # B1 of SURVEY 8d: the loop-structured Python / NumPy restatement (oracle/ref_py.py: interpreter-bound like the
    # reference itself, which adds BioPython / OpenBabel object traffic on top) on a bounded 3000-atom sample
    cpu_py = None
    if cpu is not None and args.workload == 'config3':
        try:
            from oracle import ref_py
            spy = synth.config3(3000, seed=3)
            ocs = oracle.OracleComplex(spy)
            ocs.make_selection(None)
            cand_s = int(ocs.atom_contacts(args.cutoff, args.vdw_comp, False)['stats'][0])
            rp = ref_py.RefPy(spy)
            n_py, s_py, acc_py = 0, 0.0, 0
            while s_py < 3.0 and n_py < 50:
                t0 = time.perf_counter()
                acc_py += len(rp.atom_contacts(args.cutoff, args.vdw_comp, False)['i'])
                s_py += time.perf_counter() - t0
                n_py += 1
            cpu_py = {'value': round(cand_s * n_py / s_py, 1), 'unit': 'candidate atom-pairs/s', 'cores': 1, 'kind': 'port',
                      'contacts_per_s': round(acc_py / s_py, 1),
                      'sample': f'{n_py} passes of oracle/ref_py.py (Python loops + the reference\'s own NumPy calls per pair; vectorised '
                                f'brute-force pair search) on a 3000-atom structure of the same density, {s_py:.1f} s; candidate pairs '
                                f'counted as the grid search of the C restatement counts them'}
        except Exception as exc:
            cpu_py = {'error': repr(exc)}

    line = {
        'metric': 'evaluated atom-pairs/s', 'value': round(value, 1), 'unit': 'candidate atom-pairs/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64 distance test / f32+f64 SIFt',
        'data': 'synthetic',
        'config': {'workload': workload, 'atoms_per_gpu': args.atoms, 'cutoff_A': args.cutoff, 'vdw_comp': args.vdw_comp,
                   'parallelism': f'slab{world}' if world > 1 else 'single'},
        'wall_clock_per_structure_ms': round(ms_per_step, 4),
        'pairs': {'candidates': cand_all, 'accepted': acc_all, 'contacts_emitted': emitted_all,
                  'expansion_candidates_6A': exp_all, 'bags': counts},
        'accepted_pairs_per_s': round(acc_all * args.steps / elapsed, 1),
        'kernel_ms': {k: round(v, 5) for k, v in per_kernel.items()},
        'kernel_launches_per_step': {k: v['launches'] / args.steps for k, v in ktimes.items() if v['launches']},
        'ms_per_step_profiled_pass': round(elapsed_profiled / args.steps * 1e3, 4),
        'roofline_all_kernels': roofline_all,
        'throughput_several_in_flight': in_flight,
        'cpu_baseline_all_cores': cpu_mc,
        'cpu_baseline_python': cpu_py,
        'host_us_per_step': {k: round(v, 1) for k, v in host_times.items() if k != 'passes'} if world == 1 else None,
        'launch_mode': 'direct launches on two HIP streams, one host wait per step (pinned completion word, bounded spin); kernel_ms from a second pass of the same steps with HIP events (each bracket adds ~4 us to a small kernel; rocprofv3 averages are in profiles/)',
        'per_step_exchange': (None if world == 1 else ('selection_plus halo bits (P2P) + residue sets (all-reduce MAX) over RCCL' if args.staged_exchange else 'none: whole-structure selection, every rank knows selection_plus and the residue sets (DESIGN.md 6)')),
        'halo_exchange_ms': round(halo_ms, 3), 'halo_exchange': (halo_note if world > 1 else None), 'setup_s': round(gen_s, 2), 'home_atoms_rank0': n_local_home,
        'roofline': roofline, 'cpu_baseline': cpu,
    }
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
