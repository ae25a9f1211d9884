#!/usr/bin/env python3
"""bench.py — evaluated atom-pairs/s of the run_arpeggio hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--atoms A]

One "step" = one pass of the whole hot path (InteractionComplex.run_arpeggio,
interactions.py:329-347: selection, the neighbour grid of the pass, 5 A neighbour search + fused
15-flag SIFt evaluation, ring/amide plane kernels) over the structure resident in HBM; results
stay in HBM.  Unit of work = one candidate atom pair of the 5 A contact search (SURVEY.md
§8d).  EVERY timed step builds its contact grid (arp_set_grid_reuse(0)): `value` and
`ms_per_step` are that pass; the pass that keeps the grid of the pass before it
is the extra key `pass_with_grid_kept`, `end_to_end` is a new structure every step with all
five bags fetched in canonical order, and `wall_clock_per_structure_ms` is that figure.
N > 1: the box is elongated along x (weak scaling: every GPU owns one config-3 cube of
--atoms atoms, the same per-GPU workload as N = 1), sharded into N slabs with a one-cell
halo exchanged over RCCL before the timed region; the ranks meet over plain TCP
(arpeggio_amd/rendezvous.py) and the timed region is bracketed by arp_device_synchronize +
barrier — no PyTorch in this process.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the
dominant kernel (HIP-event time on the context's own stream, SURVEY 8d's compulsory bytes of
that kernel) and `cpu_baseline` (the C oracle timed on one host core on a bounded sample of
the same workload).  The GPU legs run back to back; the CPU baselines come last.
"""
import argparse
import json
import math
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
    ap.add_argument('--atoms', type=int, default=100_000,
                    help='atoms per GPU (weak scaling): 100 000 = BASELINE configs[2] on every GPU, at N = 1 and at N > 1 alike; '
                         '--gpus 8 --atoms 250000 is configs[3] (2 M atoms on 8 GPUs)')
    ap.add_argument('--min-seconds', type=float, default=2.0,
                    help='the K timed steps are repeated as a block until the timed region is at least this long')
    ap.add_argument('--no-end-to-end', action='store_true', help='skip the fresh-structure-per-step measurement')
    ap.add_argument('--cutoff', type=float, default=5.0)
    ap.add_argument('--vdw-comp', type=float, default=0.1)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--workload', choices=('config3', 'standin'), default='config3',
                    help="config3: BASELINE configs[2] (the headline); standin: the 1tqn_h stand-in of configs[1] (5.9 k atoms incl. "
                         "explicit hydrogens, whole structure) on one GPU beside the CPU restatement")
    ap.add_argument('--batch', type=int, default=1,
                    help='N = 1 only: B structures of the stand-in family in ONE pass (arp_set_batch); prints the batched line instead')
    ap.add_argument('--inflight', type=int, default=3, help='contexts (host threads) of the extra several-structures-in-flight measurement; 1 = skip')
    ap.add_argument('--host-halo', action='store_true', help='N > 1: pack / merge the halo records on the host (the round-1 path) instead of on the device')
    ap.add_argument('--staged-exchange', action='store_true',
                    help='N > 1: run the general three-stage protocol (selection_plus bits over P2P + residue-set all-reduce '
                         'every step) although the whole-structure selection of the benchmark does not need it')
    ap.add_argument('--cpu-sample-atoms', type=int, default=100_000)
    ap.add_argument('--no-other-configs', action='store_true', help='skip the stand-in (configs[0], [1]) and rings (configs[4]) legs of the default N = 1 run')
    ap.add_argument('--no-config4', action='store_true', help='skip the strong-scaling leg (BASELINE configs[3]: ONE 2 M-atom structure cut into N slabs)')
    ap.add_argument('--dry-run', action='store_true',
                    help='everything a --gpus N launch does up to (not including) the first HIP call: environment, rank -> device mapping, '
                         'rendezvous of the ranks, broadcast of a 128-byte id, barrier, reduction; prints one JSON line on rank 0 (tests/test_sharding.py)')
    return ap.parse_args()


def algorithmic_bytes(kernel, n_binned, ncell, n_pairs, n_h=0):
    """HBM bytes one launch moves AS IMPLEMENTED, the 8 B / pair intermediate list included.  n_h = explicit
    hydrogens of the structure (24-byte float64 coordinates, read by the hydrogen-geometry tests of k_sift)."""
    if kernel == 'bin':         # k_compact_atoms: 52 B of static columns + the row's cell (twice: its own, its neighbour's) read, 52 B of records + h_off + the cell written, the cell table
        return 60 * n_binned + 60 * n_binned + 4 * (ncell + 1)
    if kernel == 'search':      # read each sorted record once (xyzm 16 B + aux 16 B), the cell table, write the pair list
        return 32 * n_binned + 4 * (ncell + 1) + 8 * n_pairs
    if kernel == 'mark_search':  # same reads, writes one byte per marked atom
        return 32 * n_binned + 4 * (ncell + 1) + n_binned
    if kernel == 'sift':        # pair list + each atom's two 16-byte quads and h_off once + its hydrogens + 15-byte output record
        return 8 * n_pairs + 36 * n_binned + 24 * n_h + 15 * n_pairs
    raise KeyError(kernel)


def survey_8d_bytes(kernel, n_binned, ncell, n_pairs):
    """SURVEY.md 8(d)'s COMPULSORY bytes (B_alg = 140 N + 16 P for the whole pipeline), split over the kernels that move them;
    the intermediate pair list of the split design (8 B / pair written by k_search, read by k_sift) is NOT compulsory and not
    counted.  bin + search + sift = 140 N + 16 P (+ the cell table)."""
    if kernel == 'bin':          # bin pass (16 N read, 4 N written) + sort / scatter (36 N read, 36 N written)
        return 92 * n_binned
    if kernel in ('search', 'mark_search'):   # each sorted record once + the cell table
        return 32 * n_binned + 4 * (ncell + 1)
    if kernel == 'sift':         # hydrogen / bond side arrays + the output records
        return 16 * n_binned + 16 * n_pairs
    raise KeyError(kernel)


def _timed_passes(step, min_s=0.25, min_n=50, warm=8):
    for _ in range(warm):
        out = step()
    n, t0 = 0, time.perf_counter()
    while n < min_n or time.perf_counter() - t0 < min_s:
        out = step()
        n += 1
    return (time.perf_counter() - t0) / n * 1e3, n, out


def _kernel_us(ctx, step, n=40):
    ctx.set_profiling(True)
    ctx.kernel_times(reset=True)
    for _ in range(n):
        step()
    kt = ctx.kernel_times(reset=True)
    ctx.set_profiling(False)
    return {k: v['ms'] / max(v['launches'], 1) * 1e3 for k, v in kt.items() if v['launches']}


def _dominant_roofline(kus, st, kernels=('bin', 'search', 'sift', 'mark_search')):
    cands = {k: kus[k] for k in kernels if k in kus}
    dom = max(cands, key=cands.get)
    n_b = st['binned'] if dom != 'mark_search' else st.get('atoms', st['binned'])
    b = survey_8d_bytes(dom, n_b, st['cells'], st['emitted'])
    ach = b / (cands[dom] * 1e-6) / 1e9
    return {'kernel': f'k_{dom}', 'bound': 'hbm', 'achieved': round(ach, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 6),
            'traffic': None, 'algorithmic_bytes_per_launch': int(b), 'avg_launch_us': round(cands[dom], 2),
            'note': 'SURVEY 8d bytes of the longest kernel of the pass over its HIP-event duration in this run'}


def other_single_gpu_configs(args, device):
    """The single-GPU configurations of BASELINE.json beside the headline (configs[2]), each measured the same way — a resident
    pass that builds its grid, the SURVEY 8d roofline of its longest kernel, the C oracle on one host core beside it:
      standin_whole   configs[1]: the 1tqn_h stand-in (the file is not available), whole structure
      standin_ligand  configs[0]: the same structure, `-s /A/508/` (a ligand and its binding site)
      rings           configs[4]: 10 000 aromatic rings + 10 000 amide groups: the centroid + normal plane kernels
    World size 1 only; a few seconds of GPU time and a few seconds of CPU time in all."""
    from arpeggio_amd import synth, _capi
    out = {}
    cx = _capi.Context(device)
    try:
        # ---- configs[1] / configs[0]: the stand-in
        pc = synth.proteinlike()
        cx.set_complex(pc)
        cx.set_grid_reuse(False)
        step = lambda: cx.run_launch(args.cutoff, args.vdw_comp, False, 6.0)
        ms, n, cnt = _timed_passes(step)
        st = cx.stats()
        kus = _kernel_us(cx, step)
        whole = {'workload': f'1tqn_h stand-in (BASELINE configs[1]; the file is not available): {pc.n_atoms} atoms incl. explicit hydrogens, whole structure, 5 A cutoff',
                 'ms_per_step': round(ms, 4), 'steps': n, 'value': round(st['candidates'] / (ms * 1e-3), 1), 'unit': 'candidate atom-pairs/s',
                 'pairs': {'candidates': int(st['candidates']), 'accepted': int(st['accepted']), 'contacts_emitted': int(st['emitted'])},
                 'bags': {k: int(v) for k, v in cnt.items()}, 'kernel_us': {k: round(v, 2) for k, v in kus.items()},
                 'roofline': _dominant_roofline(kus, st)}
        out['standin_whole'] = whole
        lig = (np.asarray(pc.res_seq)[np.asarray(pc.res_id)] == 508).astype(np.uint8)
        cx.set_selection(lig)
        ms, n, cnt = _timed_passes(step)
        st = cx.stats()
        kus = _kernel_us(cx, step)
        ligand = {'workload': 'the same stand-in, selection /A/508/ (BASELINE configs[0]): 6 A expansion of the ligand, selection_plus compacted into the grid of the pass, search, per-pair kernel, ring / amide loops',
                  'selection': '/A/508/', 'selected_atoms': int(lig.sum()), 'ms_per_step': round(ms, 4), 'ms_per_pass': round(ms, 4), 'steps': n,
                  'value': round(st['candidates'] / (ms * 1e-3), 1), 'unit': 'candidate atom-pairs/s (contact search; the 6 A expansion tests are counted beside)',
                  'pairs': {'candidates': int(st['candidates']), 'expansion_candidates_6A': int(st['expand_candidates']), 'accepted': int(st['accepted']),
                            'contacts_emitted': int(st['emitted'])},
                  'candidate_pairs': int(st['candidates']), 'bags': {k: int(v) for k, v in cnt.items()}, 'kernel_us': {k: round(v, 2) for k, v in kus.items()},
                  'roofline': _dominant_roofline(kus, dict(st, atoms=pc.n_atoms))}
        out['standin_ligand'] = ligand
        # ---- configs[4]: the plane kernels on 10 k rings + 10 k amides
        pr = synth.config5()
        cx.set_complex(pr)
        cx.set_grid_reuse(False)
        ms, n, cnt = _timed_passes(step, min_n=30)
        rings = {'workload': f'BASELINE configs[4]: {pr.n_rings} aromatic rings + {pr.n_amides} amide groups ({pr.n_atoms} atoms) in a 100 A cube; whole pass and each plane loop alone',
                 'ms_per_step': round(ms, 4), 'steps': n, 'bags': {k: int(v) for k, v in cnt.items()}, 'loops': {}}
        cx.set_profiling(True)
        R = int(pr.n_rings)
        for name, units, unit_bytes in (('plane_plane', R, 56), ('atom_plane', R, 56), ('group_group', int(pr.n_amides), 32), ('group_plane', int(pr.n_amides), 32)):
            for _ in range(5):
                cx.launch_bag(name)
            cx.kernel_times(reset=True)
            for _ in range(40):
                cx.launch_bag(name)
            kt = cx.kernel_times(reset=True).get('planes')
            us = kt['ms'] / max(kt['launches'], 1) * 1e3 if kt else None
            emitted = int(cnt.get(name, 0))
            b = unit_bytes * units + 32 * emitted          # 56 B per ring (centre + normal f64 + ids; 32 B per amide: f32) + 32 B per emitted contact
            rings['loops'][name] = {'kernel_us': None if us is None else round(us, 2), 'records': emitted, 'algorithmic_bytes': int(b),
                                    'roofline': None if not us else {'bound': 'hbm', 'achieved': round(b / (us * 1e-6) / 1e9, 3), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                                                     'frac': round(b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 7), 'traffic': None}}
        cx.set_profiling(False)
        pp = rings['loops']['plane_plane']
        rings['roofline'] = dict(pp['roofline'] or {}, kernel='plane-plane loop (k_planes, one launch of its own here)', algorithmic_bytes_per_launch=pp['algorithmic_bytes'],
                                 note='SURVEY 8d: 56 B per ring + 32 B per emitted contact; latency-bound list evaluation, the HBM fraction is tiny by construction')
        rings['value'] = round(R * R / (pp['kernel_us'] * 1e-6), 1) if pp['kernel_us'] else None
        rings['unit'] = "ordered ring pairs of the reference's loop per second (R^2 / plane-plane kernel time)"
        out['rings'] = rings
    finally:
        cx.close()
    return out


CONFIG4_ATOMS, CONFIG4_BOX_SLABS = 2_000_000, 8      # BASELINE configs[3]: eight 250 k-atom cubes side by side along x


def config4_workload(world):
    return (f'BASELINE configs[3]: ONE synthetic structure of {CONFIG4_ATOMS} atoms (synth.slab_config(250000, 8): box of eight config-3-density cubes along x, '
            f'rho = 0.05 / A^3, 5 A cutoff), whole structure, cut into {world} x-slab{"s" if world > 1 else ""} of {CONFIG4_ATOMS // world} atoms'
            + (' (+ one-cell halo over RCCL)' if world > 1 else ' on one GPU'))


def config4_single_gpu(args, device):
    """multi_gpu.config4_strong at N = 1 (and configs.config4_2m_atoms): the 2 M-atom structure of BASELINE configs[3] on ONE GPU —
    resident passes that build their grid, wall clock of a fresh upload + first pass + sorted fetch, SURVEY 8d roofline of the
    longest kernel.  The same structure is cut into N x-slabs by `--gpus N` (strong scaling)."""
    from arpeggio_amd import synth, _capi
    t0 = time.perf_counter()
    pc = synth.slab_config(CONFIG4_ATOMS // CONFIG4_BOX_SLABS, CONFIG4_BOX_SLABS, seed=4)
    gen_s = time.perf_counter() - t0
    cx = _capi.Context(device)
    try:
        cx.set_grid_reuse(False)
        blob = _capi.pack_blob(pc)
        walls, buf = [], None
        for _ in range(3):
            cx.device_synchronize()
            tt = time.perf_counter()
            cx.set_blob(blob)
            cnt = cx.run_launch(args.cutoff, args.vdw_comp, False, 6.0)
            bags, buf = cx.fetch_packed(buf)
            walls.append(time.perf_counter() - tt)
        k_aa = bags['atom_atom']
        assert len(k_aa['i']) == cnt['atom_atom']
        step = lambda: cx.run_launch(args.cutoff, args.vdw_comp, False, 6.0)
        ms, n, cnt = _timed_passes(step, min_s=1.0, min_n=20, warm=3)
        st = cx.stats()
        kus = _kernel_us(cx, step, n=10)
        b_pass = 140 * st['binned'] + 16 * st['emitted']
        return {'workload': config4_workload(1), 'n_gpus': 1, 'atoms_per_gpu': CONFIG4_ATOMS,
                'ms_per_step': round(ms, 4), 'steps': n, 'value': round(st['candidates'] / (ms * 1e-3), 1), 'unit': 'candidate atom-pairs/s',
                'step_definition': 'one whole run_arpeggio pass over the resident structure that builds its contact grid',
                'pairs': {'candidates': int(st['candidates']), 'accepted': int(st['accepted']), 'contacts_emitted': int(st['emitted'])},
                'bags': {k: int(v) for k, v in cnt.items()}, 'kernel_us': {k: round(v, 2) for k, v in kus.items()},
                'wall_clock_per_structure_ms': round(min(walls[1:]) * 1e3, 3),
                'wall_clock_per_structure_definition': 'arp_set_blob (one H2D copy + device-side validation) + first pass + device sort + one-copy fetch of all five bags; best of the last two of three',
                'roofline': _dominant_roofline(kus, st),
                'roofline_pass': {'bytes_per_pass': int(b_pass), 'bytes_model': 'SURVEY 8d: B_alg = 140 N + 16 P', 'achieved': round(b_pass / (ms * 1e-3) / 1e9, 2),
                                  'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(b_pass / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)},
                'structure_generation_s_host': round(gen_s, 1)}
    finally:
        cx.close()


def other_configs_cpu_baselines(args, cfg):
    """cpu_baseline of the configurations of other_single_gpu_configs: the C oracle (oracle/ref_c.c) on ONE host core, a couple of
    seconds each, on the same structures."""
    import oracle
    from arpeggio_amd import synth
    pc = synth.proteinlike()
    oc = oracle.OracleComplex(pc)
    oc.make_selection(None)
    n_known = len(oc.atom_contacts(args.cutoff, args.vdw_comp, False)['i'])
    passes, cpu_s, cand_cpu = 0, 0.0, 0
    while cpu_s < 2.0 and passes < 100:
        t0 = time.perf_counter()
        r = oc.atom_contacts(args.cutoff, args.vdw_comp, False, cap_hint=n_known)
        oc.atom_plane(); oc.plane_plane(); oc.group_group(); oc.group_plane()
        cpu_s += time.perf_counter() - t0
        cand_cpu += int(r['stats'][0])
        passes += 1
    cfg['standin_whole']['cpu_baseline'] = {'value': round(cand_cpu / cpu_s, 1), 'unit': 'candidate atom-pairs/s', 'cores': 1, 'kind': 'port',
                                            'ms_per_structure': round(cpu_s / passes * 1e3, 2),
                                            'sample': f'{passes} whole passes of the C oracle (5 A search + per-pair SIFt + the four ring / amide loops) on the same structure, {cpu_s:.1f} s'}
    lig = (np.asarray(pc.res_seq)[np.asarray(pc.res_id)] == 508).astype(np.uint8)
    passes, cpu_s, cand_cpu = 0, 0.0, 0
    while cpu_s < 1.0 and passes < 200:
        t0 = time.perf_counter()
        oc.make_selection(lig)
        r = oc.atom_contacts(args.cutoff, args.vdw_comp, False)
        oc.atom_plane(); oc.plane_plane(); oc.group_group(); oc.group_plane()
        cpu_s += time.perf_counter() - t0
        cand_cpu += int(r['stats'][0])
        passes += 1
    cfg['standin_ligand']['cpu_baseline'] = {'value': round(cand_cpu / cpu_s, 1), 'unit': 'candidate atom-pairs/s', 'cores': 1, 'kind': 'port',
                                             'ms_per_structure': round(cpu_s / passes * 1e3, 2),
                                             'sample': f'{passes} passes of the C oracle with the same selection (6 A expansion + 5 A search + per-pair SIFt + ring / amide loops), {cpu_s:.1f} s'}
    pr = synth.config5()
    ocr = oracle.OracleComplex(pr)
    ocr.make_selection(None)
    R = int(pr.n_rings)
    t0 = time.perf_counter()
    rec = ocr.plane_plane()
    cpu_s = time.perf_counter() - t0
    n_rec = len(next(iter(rec.values()))) if isinstance(rec, dict) and rec else None
    cfg['rings']['cpu_baseline'] = {'value': round(R * R / cpu_s, 1), 'unit': 'ordered ring pairs/s', 'cores': 1, 'kind': 'port', 'ms_per_structure': round(cpu_s * 1e3, 1),
                                    'sample': f'one pass of the plane-plane loop of the C oracle over the same {R} rings (O(R^2) ordered pairs like the reference, I:1064-1194)',
                                    'records': n_rec}


def per_gpu_workload(atoms):
    """config.workload: what ONE GPU works on — the same string at every N (weak scaling)."""
    return f'synthetic {atoms} random-coordinate atoms per GPU, rho=0.05/A^3, 5 A cutoff (BASELINE configs[2] on every GPU)'


def dry_run(args, rank, local_rank, world, timeout):
    """What a `--gpus N` launch does before its first HIP call, without a GPU: the launcher's environment, the rank -> device
    mapping, the rendezvous (arpeggio_amd/rendezvous.py), the broadcast of a 128-byte communicator id, a barrier and a
    reduction.  Rank 0 prints one JSON line."""
    import zlib
    from arpeggio_amd.rendezvous import TcpRendezvous
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    share_gpu = os.environ.get('ARP_BENCH_SHARE_GPU') == '1'
    device = 0 if share_gpu else local_rank
    t0 = time.perf_counter()
    rdzv = TcpRendezvous(rank, world, timeout=timeout) if world > 1 else None
    uid = os.urandom(128) if rank == 0 else None
    if rdzv is not None:
        uid = rdzv.broadcast(uid)
    me = {'rank': rank, 'local_rank': local_rank, 'device': device, 'world': world, 'pid': os.getpid(), 'uid_crc32': zlib.crc32(uid),
          'master': f"{os.environ.get('MASTER_ADDR', '127.0.0.1')}:{os.environ.get('MASTER_PORT', '29500')}", 'rendezvous_s': round(time.perf_counter() - t0, 3)}
    everybody = [json.loads(b) for b in (rdzv.allgather(json.dumps(me).encode()) if rdzv is not None else [json.dumps(me).encode()])]
    total = float(rdzv.allreduce(np.array([rank + 1.0]), 'sum')[0]) if rdzv is not None else 1.0
    if rdzv is not None:
        rdzv.barrier()
    if rank == 0:
        # the legs a real run of this launch adds to its line beside the weak-scaling headline (multi_gpu.*)
        strong = (args.atoms == 100_000 and not args.no_config4 and CONFIG4_ATOMS % world == 0 and (world > 1 or not args.no_other_configs))
        legs = {'config4_strong': {'workload': config4_workload(world), 'n_gpus': world, 'atoms_per_gpu': CONFIG4_ATOMS // world, 'scaling': 'strong'} if strong else None}
        print(json.dumps({'dry_run': True, 'n_gpus': world, 'ranks': everybody, 'allreduce_sum': total, 'legs': legs,
                          'config': {'workload': per_gpu_workload(args.atoms), 'atoms_per_gpu': args.atoms}}), flush=True)
    if rdzv is not None:
        rdzv.barrier()
        rdzv.close()


def bench_batch(args):
    """B protein-sized structures per pass (arp_set_batch): the reference's production use is the weekly PDBe release —
    10^5 entries of a few thousand atoms — and one such structure is four launches of fixed cost.  Prints one JSON line
    with the contract's keys; `value` = candidate pairs of all structures of the batch x steps / time."""
    from arpeggio_amd import synth, _capi, batch as _batch
    if _capi.device_count() < 1:
        raise SystemExit('bench.py needs a GPU (no CPU fallback exists)')
    B = args.batch
    t0 = time.perf_counter()
    distinct = [synth.proteinlike(seed=2 + k, id=f'standin{k}') for k in range(min(B, 8))]
    pcs = [distinct[k % len(distinct)] for k in range(B)]
    big, off = _batch.concat_complexes(pcs)
    gen_s = time.perf_counter() - t0
    ctx = _capi.Context(0)
    ctx.set_complex(big)
    ctx.declare_batch(off)
    ctx.set_grid_reuse(False)      # the timed pass builds its contact grid, as in main(); the pass on a kept grid is an extra key

    def step():
        return ctx.run_launch(args.cutoff, args.vdw_comp, False, 6.0)

    for _ in range(5 + args.warmup):
        counts = step()
    ctx.device_synchronize()
    t_trial = time.perf_counter()
    for _ in range(args.steps):
        counts = step()
    repeats = max(1, int(math.ceil(args.min_seconds / max(time.perf_counter() - t_trial, 1e-6))))
    ctx.device_synchronize()
    t1 = time.perf_counter()
    for _ in range(repeats * args.steps):
        counts = step()
    ctx.device_synchronize()
    elapsed = time.perf_counter() - t1
    timed_steps = repeats * args.steps
    st = ctx.stats()
    ctx.set_profiling(True)
    ctx.kernel_times(reset=True)
    for _ in range(args.steps):
        step()
    ktimes = ctx.kernel_times(reset=True)
    ctx.set_profiling(False)
    per_kernel = {k: (v['ms'] / max(v['launches'], 1)) for k, v in ktimes.items() if v['launches']}
    dom = max((k for k in ('search', 'sift') if k in per_kernel), key=lambda k: per_kernel[k])
    b_alg = algorithmic_bytes(dom, st['binned'], st['cells'], st['emitted'], int(big.h_xyz.shape[0]))
    ms_per_step = elapsed / timed_steps * 1e3
    # the same batch on the grid the pass before it built (see main(): pass_with_grid_kept)
    ctx.set_grid_reuse(True)
    for _ in range(5):
        step()
    ctx.device_synchronize()
    n_rb, t_rb = 0, time.perf_counter()
    while n_rb < args.steps or time.perf_counter() - t_rb < 0.25:
        step()
        n_rb += 1
    ctx.device_synchronize()
    el_rb = time.perf_counter() - t_rb
    grid_kept = {'ms_per_step': round(el_rb / n_rb * 1e3, 4), 'us_per_structure': round(el_rb / n_rb / B * 1e6, 3), 'steps': n_rb,
                 'note': 'the same pass on the contact grid of the pass before it (a cache across identical passes: not the headline)'}
    # the same structures one at a time on the same context (resident pass of each distinct structure)
    single_ms = []
    for pc in distinct:
        ctx.set_complex(pc)
        for _ in range(8):
            ctx.run_launch(args.cutoff, args.vdw_comp, False, 6.0)
        tt = time.perf_counter()
        for _ in range(200):
            ctx.run_launch(args.cutoff, args.vdw_comp, False, 6.0)
        single_ms.append((time.perf_counter() - tt) / 200 * 1e3)
    # end to end: a FRESH batch every step (blob H2D + partition + static columns + lists + pass + five bags D2H), two batches alternating
    e2e = None
    try:
        pcs2 = pcs[1:] + pcs[:1]
        big2, off2 = _batch.concat_complexes(pcs2)
        blobs = [(_capi.pack_blob(big), off), (_capi.pack_blob(big2), off2)]
        ctx.set_blob(blobs[0][0]); ctx.declare_batch(blobs[0][1])
        cnt = ctx.run_launch(args.cutoff, args.vdw_comp, False, 6.0)
        pbuf = [_capi.pinned_empty(int(cnt['atom_atom'] * 1.25) * 16 + (8 << 20), np.uint8)]

        def one(k):
            ctx.set_blob(blobs[k % 2][0]); ctx.declare_batch(blobs[k % 2][1])
            ctx.run_launch(args.cutoff, args.vdw_comp, False, 6.0)
            _, pbuf[0] = ctx.fetch_packed(pbuf[0])      # canonical order (the records of a structure are contiguous in it), one copy
        for k in range(4):
            one(k)
        n_e, tt = 0, time.perf_counter()
        while n_e < 10 or time.perf_counter() - tt < 0.5:
            one(n_e); n_e += 1
        e2e_ms = (time.perf_counter() - tt) / n_e * 1e3
        e2e = {'ms_per_batch': round(e2e_ms, 4), 'us_per_structure': round(e2e_ms / B * 1e3, 2), 'batches': n_e,
               'upload_bytes': int(blobs[0][0].nbytes),
               'note': 'fresh batch per step: arp_set_blob (one H2D copy + device validation) + arp_set_batch + static columns + '
                       'ring / amide lists + pass + device sort of the atom-atom bag (canonical order: the records of a structure are '
                       'contiguous) and of the ring / amide bags + all five bags in one copy (arp_fetch_packed)'}
    except Exception as exc:
        e2e = {'error': repr(exc)}
    cpu = None
    if not args.no_cpu_baseline:
        import oracle
        tt = time.perf_counter()
        for pc in distinct:
            oc = oracle.OracleComplex(pc)
            oc.make_selection(None)
            oc.atom_contacts(args.cutoff, args.vdw_comp, False)
            oc.plane_plane(); oc.atom_plane(); oc.group_group(); oc.group_plane()
        cpu_s = time.perf_counter() - tt
        cpu = {'value': round(st['candidates'] / B * len(distinct) / cpu_s, 1), 'unit': 'candidate atom-pairs/s', 'cores': 1, 'kind': 'port',
               'ms_per_structure': round(cpu_s / len(distinct) * 1e3, 2),
               'sample': f'one full run_arpeggio pass of the C oracle (oracle/ref_c.c, gcc -O2, 1 thread) over each of the {len(distinct)} distinct '
                         f'structures of the batch, {cpu_s:.2f} s; pairs counted as the GPU counts them'}
    line = {
        'metric': 'evaluated atom-pairs/s', 'value': round(st['candidates'] * timed_steps / elapsed, 1), 'unit': 'candidate atom-pairs/s',
        'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4), 'timed_steps': timed_steps,
        'timed_region_s': round(elapsed, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f64 distance test / f32+f64 SIFt', 'data': 'synthetic',
        'config': {'workload': f'{B} structures of the 1tqn_h stand-in family (BASELINE configs[1]; {len(distinct)} distinct synthetic chains of 480 residues + '
                               f'haem-like ligand + waters, ~5.9 k atoms each incl. explicit hydrogens, repeated) in ONE pass (arp_set_batch), whole structures, 5 A cutoff',
                   'structures_per_pass': B, 'atoms_per_pass': int(big.n_atoms), 'cutoff_A': args.cutoff, 'vdw_comp': args.vdw_comp, 'parallelism': 'single'},
        'structures_per_s': round(B * timed_steps / elapsed, 1), 'us_per_structure': round(ms_per_step / B * 1e3, 3),
        'single_structure_pass_ms': {'mean': round(float(np.mean(single_ms)), 4), 'min': round(float(np.min(single_ms)), 4), 'max': round(float(np.max(single_ms)), 4),
                                     'note': 'resident pass of each distinct structure alone on the same context'},
        'speedup_vs_one_at_a_time': round(float(np.mean(single_ms)) / (ms_per_step / B), 2),
        'pairs': {'candidates': float(st['candidates']), 'accepted': float(st['accepted']), 'contacts_emitted': float(st['emitted']),
                  'bags': {k: int(v) for k, v in counts.items()}},
        'kernel_ms': {k: round(v, 5) for k, v in per_kernel.items()}, 'pass_with_grid_kept': grid_kept,
        'roofline': {'kernel': f'k_{dom}', 'bound': 'hbm', 'achieved': round(b_alg / (per_kernel[dom] * 1e-3) / 1e9, 2), 'peak': 8000.0, 'unit': 'GB/s',
                     'frac': round(b_alg / (per_kernel[dom] * 1e-3) / 1e9 / 8000.0, 6), 'traffic': None,
                     'algorithmic_bytes_per_launch': int(b_alg), 'avg_launch_ms': round(per_kernel[dom], 5),
                     'note': 'HIP-event duration of the dominant kernel in this run; no PMC run of this workload is committed, hence traffic = null'},
        'cpu_baseline': cpu, 'end_to_end': e2e, 'setup_s': round(gen_s, 2),
    }
    print(json.dumps(line))
    ctx.close()


def main():
    args = parse()
    if args.batch > 1:
        if args.gpus != 1 or int(os.environ.get('WORLD_SIZE', '1')) != 1:
            raise SystemExit('--batch is a one-GPU measurement')
        return bench_batch(args)
    os.environ.setdefault('OMP_WAIT_POLICY', 'passive')   # (cpu_baseline_all_cores: idle OpenMP threads must not spin inside a CPU quota)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')

    rdzv_timeout = float(os.environ.get('ARP_RDZV_TIMEOUT', '180'))
    if args.dry_run:
        return dry_run(args, rank, local_rank, world, rdzv_timeout)
    from arpeggio_amd import synth, _capi
    from arpeggio_amd.rendezvous import TcpRendezvous
    if _capi.device_count() < 1:
        raise SystemExit('bench.py needs a GPU (no CPU fallback exists)')
    # ARP_BENCH_SHARE_GPU=1 (debug only): every rank uses GPU 0 and the exchange goes through host buffers (RCCL refuses two
    # ranks on one device), so the N > 1 code path can be exercised on a one-GPU box.  Never set by the driver.
    share_gpu = os.environ.get('ARP_BENCH_SHARE_GPU') == '1'
    if share_gpu:
        local_rank = 0
    # Data path: the library's own RCCL communicator (arp_comm_*: ncclSend / ncclRecv / ncclAllReduce on the context's stream).
    # The rendezvous of the ranks — the 128-byte communicator id, the barriers around the timed region, the reduction of the
    # per-rank timings — is a few TCP messages through rank 0 (arpeggio_amd/rendezvous.py).  No PyTorch in this process: its
    # ROCm build brings a HIP runtime and an RCCL of its own, and two of each in one process do not end well.
    comm_device = None if share_gpu else local_rank
    rdzv = TcpRendezvous(rank, world, timeout=rdzv_timeout) if world > 1 else None
    transport = rdzv

    # ---------------- workload ----------------
    t0 = time.perf_counter()
    halo_ms, halo_bytes, shard_setup_ms, shard_timings = 0.0, 0, 0.0, None
    if world == 1:
        if args.workload == 'standin':
            pc = synth.proteinlike()
            args.atoms = pc.n_atoms
            workload = (f'1tqn_h stand-in (BASELINE configs[1]: the file is not available): synthetic chain of 480 residues + haem-like '
                        f'ligand + waters, {pc.n_atoms} atoms incl. explicit hydrogens, whole structure, 5 A cutoff')
        else:
            pc = synth.config3(args.atoms, seed=3)
            workload = per_gpu_workload(args.atoms)
        ctx = _capi.Context(local_rank)
        ctx.set_complex(pc)
        n_local_home = pc.n_atoms
    else:
        from arpeggio_amd import sharding
        full = None      # (only the host-buffer debug modes build the whole structure: a rank generates its own slab)
        workload = per_gpu_workload(args.atoms)      # (the same per-GPU workload as N = 1; the slabs are in config.sharding)
        # no fallback: if the exchange over RCCL fails, the run fails (a scaling figure must not be printed without it)
        ctx = _capi.Context(local_rank)
        if comm_device is not None:
            uid = np.frombuffer(rdzv.broadcast(_capi.Context.comm_unique_id().tobytes() if rank == 0 else None), np.uint8).copy()
            ctx.comm_init(rank, world, uid)
        t_sh = time.perf_counter()
        if comm_device is None:
            args.host_halo = True      # (one-GPU debug mode: no RCCL between ranks that share a device)
        if args.host_halo:
            halo_note = 'records of the one-cell halo packed on the host, exchanged through host buffers (TCP rendezvous), merged on the host'
            full = synth.slab_config(args.atoms, world, seed=4)
            shard = sharding.make_shard_distributed(full, rank, world, transport)
            sharding.upload_shard(ctx, shard, whole_structure=not args.staged_exchange)
            n_local = shard.pc.n_atoms
        else:
            halo_note = ('home records uploaded once; the one-cell halo cut out on the device, exchanged with grouped ncclSend / ncclRecv on the '
                         'device buffers (RCCL behind the C ABI: arp_shard_exchange_faces), merged into the resident structure on the device (arp_shard_*); '
                         'every rank generates the records of its own slab only (synth.slab_home_records)')
            t_gen = time.perf_counter()
            home = synth.slab_home_records(args.atoms, world, rank, seed=4)      # this rank's part of slab_config(atoms, world), nothing else
            gen_home_ms = (time.perf_counter() - t_gen) * 1e3
            shard = sharding.make_shard_device(ctx, home, rank, world, whole_structure=not args.staged_exchange)
            shard.timings_ms['generate_home_records'] = round(gen_home_ms, 3)
            n_local = shard.n_atoms
            shard_timings = shard.timings_ms
        shard_setup_ms = (time.perf_counter() - t_sh) * 1e3
        halo_ms, halo_bytes = shard.halo_ms, shard.halo_bytes
        n_local_home = int(shard.is_home.sum())
        import types
        pc = types.SimpleNamespace(n_atoms=n_local)      # the rest of the script needs the atom count of the local structure only
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
        # device-resident exchange: the library gathers the halo bits, moves them (ncclSend / ncclRecv) and reduces the
        # residue sets (ncclAllReduce) on the context's stream between the three stages of the pass
        exchange = sharding.DeviceExchange(ctx, shard)

        def step():
            return sharding.run_shard_device(ctx, exchange, args.cutoff, args.vdw_comp, False)
    else:
        def step():   # debug path (host buffers through the rendezvous)
            return sharding.run_shard(ctx, shard, transport, args.cutoff, args.vdw_comp, False)

    t_gpu_legs_begin = time.perf_counter()

    def sync_all():      # device idle, every rank here, device idle (the bracket of the timed region)
        ctx.device_synchronize()
        if rdzv is not None:
            rdzv.barrier()
        ctx.device_synchronize()

    # EVERY step of the headline builds its contact grid (the bin / sort passes are part of the pipeline; no caller
    # runs the same pass twice on one structure, so nothing a pass made may be reused by the next one).
    ctx.set_grid_reuse(False)
    # set-up, not steps: the first passes size the result buffers (a pass is re-run when one was too small); reported
    # in the line as `setup_passes`.  Then the W warm-up steps and the timed region: the K steps, repeated as a block
    # until the region is at least --min-seconds long (K x repeats steps are timed; `steps` stays K).
    SETUP_PASSES = 5
    for _ in range(SETUP_PASSES):
        counts = step()
    for _ in range(args.warmup):
        counts = step()
    sync_all()
    ctx.host_times(reset=True)
    # How many blocks of K steps fill --min-seconds is settled ONCE, before the timed region, from an untimed trial block (all
    # ranks take the largest answer): the timed region itself holds steps only — at N > 1 no collective that N = 1 would not have.
    t_trial = time.perf_counter()
    for _ in range(args.steps):
        counts = step()
    trial_s = max(time.perf_counter() - t_trial, 1e-6)
    repeats = max(1, int(math.ceil(args.min_seconds / trial_s)))
    if rdzv is not None:
        repeats = int(rdzv.allreduce(np.array([float(repeats)]), 'max')[0])
    sync_all()
    ctx.host_times(reset=True)
    t0 = time.perf_counter()
    for _ in range(repeats):
        for _ in range(args.steps):
            counts = step()          # blocks until the pass has published its counters (one host wait per step)
    sync_all()
    elapsed = time.perf_counter() - t0
    timed_steps = args.steps * repeats
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

    # ---- extra GPU legs (world == 1; informational, never `value`), back to back before any CPU baseline ----
    # (a) the pass that KEEPS the grid of the pass before it: a whole-structure pass over a resident structure whose grid
    #     the library caches — two launches instead of three
    grid_kept = None
    if world == 1:
        try:
            ctx.set_grid_reuse(True)
            for _ in range(5):
                step()
            sync_all()
            n_gk, t_gk = 0, time.perf_counter()
            while n_gk < args.steps or time.perf_counter() - t_gk < 0.25:
                step()
                n_gk += 1
            sync_all()
            el_gk = time.perf_counter() - t_gk
            st_gk = ctx.stats()
            b_gk = 48 * st_gk['binned'] + 16 * st_gk['emitted']
            grid_kept = {'ms_per_step': round(el_gk / n_gk * 1e3, 4), 'steps': n_gk, 'value': round(st_gk['candidates'] * n_gk / el_gk, 1),
                         'roofline_pass': {'bytes_per_pass': int(b_gk), 'bytes_model': '48 N + 16 P (SURVEY 8d without its bin / sort passes)',
                                           'achieved_GBps': round(b_gk / (el_gk / n_gk) / 1e9, 2), 'frac': round(b_gk / (el_gk / n_gk) / 1e9 / HBM_PEAK_GBS, 6)},
                         'note': 'the same pass on the grid the pass before it built (arp_set_grid_reuse(1), the library default for whole-structure '
                                 'passes over a resident structure): no k_compact_atoms launch.  Not the headline: a cache across identical passes'}
        except Exception as exc:   # never lose the main line over the extra measurement
            grid_kept = {'error': repr(exc)}
        ctx.set_grid_reuse(False)

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
                c2.set_grid_reuse(False)
                c2.run_launch(args.cutoff, args.vdw_comp, False, 6.0)
            per_thread = max(max(args.steps, 600) // args.inflight, 1)      # (a stable figure also when the harness asks for few steps)
            gate = threading.Barrier(args.inflight + 1)

            def worker(cx):
                gate.wait()
                for _ in range(per_thread):
                    cx.run_launch(args.cutoff, args.vdw_comp, False, 6.0)

            threads = [threading.Thread(target=worker, args=(cx,)) for cx in ctxs]
            for t in threads:
                t.start()
            ctx.device_synchronize()
            gate.wait()
            t2 = time.perf_counter()
            for t in threads:
                t.join()
            ctx.device_synchronize()
            el2 = time.perf_counter() - t2
            in_flight = {'contexts': args.inflight, 'steps': per_thread * args.inflight,
                         'ms_per_step': round(el2 / (per_thread * args.inflight) * 1e3, 4),
                         'value': round(st['candidates'] * per_thread * args.inflight / el2, 1),
                         'note': 'one arp_ctx per host thread, same structure resident in each, every pass builds its grid; not the headline value'}
            # the same contexts driven by ONE host thread: enqueue the pass of each (arp_run_enqueue), then wait for each
            rounds = max(per_thread, 1)
            ctx.device_synchronize()
            t3 = time.perf_counter()
            for _ in range(rounds):
                for cx in ctxs:
                    cx.run_enqueue(args.cutoff, args.vdw_comp, False, 6.0)
                for cx in ctxs:
                    cx.run_wait()
            el3 = time.perf_counter() - t3
            in_flight['one_thread_enqueue_wait'] = {'contexts': args.inflight, 'steps': rounds * args.inflight,
                                                     'ms_per_step': round(el3 / (rounds * args.inflight) * 1e3, 4),
                                                     'value': round(st['candidates'] * rounds * args.inflight / el3, 1)}
            for c2 in ctxs[1:]:
                c2.close()
        except Exception as exc:   # never lose the main line over the extra measurement
            in_flight = {'error': repr(exc)}

    # BASELINE configs[0]'s use of the reference (`-s /A/508/`: a ligand and its binding site) on the stand-in: the selection is
    # expanded (k_expand_small), selection_plus compacted into the pass's grid (k_compact_atoms), then search + per-pair kernel
    ligand_pass = None
    if world == 1 and args.workload == 'standin':
        try:
            lig = (np.asarray(pc.res_seq)[np.asarray(pc.res_id)] == 508).astype(np.uint8)
            if lig.sum() > 0:
                ctx.set_selection(lig)
                for _ in range(10):
                    cl = ctx.run_launch(args.cutoff, args.vdw_comp, False, 6.0)
                tl = time.perf_counter()
                for _ in range(500):
                    cl = ctx.run_launch(args.cutoff, args.vdw_comp, False, 6.0)
                ligand_pass = {'selection': '/A/508/', 'selected_atoms': int(lig.sum()), 'ms_per_pass': round((time.perf_counter() - tl) / 500 * 1e3, 4),
                               'candidate_pairs': int(ctx.stats()['candidates']), 'bags': {k: int(v) for k, v in cl.items()}}
            ctx.set_selection(np.ones(pc.n_atoms, np.uint8))
            for _ in range(3):
                step()
        except Exception as exc:   # never lose the main line over the extra measurement
            ligand_pass = {'error': repr(exc)}

    # ---- wall clock per structure, end to end (world == 1; never `value`): a FRESH structure every step — one packed
    # upload from page-locked memory (arp_set_blob: copy + device-side validation), the static record columns and the
    # ring / amide grids rebuilt, the pass, the atom-atom bag put into the canonical (i, j) order ON THE DEVICE and all
    # five bags copied into one page-locked host buffer with one copy (arp_fetch_packed).  The structures cycle through
    # four different seeds of the same workload.  Measured with one context (latency of one structure) and with three
    # contexts on three host threads (upload / pass / download of consecutive structures overlap).
    end_to_end, e2e_host = None, None
    if world == 1 and not args.no_end_to_end:
        try:
            import threading
            if args.workload == 'standin':
                fresh = [synth.proteinlike(seed=2 + k) for k in range(4)]
            else:
                fresh = [synth.config3(args.atoms, seed=3 + k) for k in range(4)]
            t_pack = time.perf_counter()
            blobs = [_capi.pack_blob(f) for f in fresh]
            pack_ms = (time.perf_counter() - t_pack) / len(blobs) * 1e3

            def make_buffer(cx):
                cx.set_blob(blobs[0])
                cnt = cx.run_launch(args.cutoff, args.vdw_comp, False, 6.0)
                return _capi.pinned_empty(int(cnt['atom_atom'] * 1.25) * 16 + (4 << 20), np.uint8)

            def one_structure(cx, blob, buf):
                cx.set_blob(blob)
                cnt = cx.run_launch(args.cutoff, args.vdw_comp, False, 6.0)
                bags, _ = cx.fetch_packed(buf)          # device sort of the atom-atom bag + one D2H copy of everything
                return cnt, bags

            # (a caller that fetches the sorted bags after every pass says so: the sort is enqueued inside run_launch, as soon as the
            # pass has reported its record count, not a host round trip later)
            ctx.set_sort_after_pass(True)
            ctx.set_packed_layout(rows=True)        # (N + 1 row offsets instead of k bgn ids: 4 bytes per record less over PCIe; bag['i'] expands them on the host)
            buf = make_buffer(ctx)
            for k in range(8):
                one_structure(ctx, blobs[k % 4], buf)
            n_e2e, t1 = 0, time.perf_counter()
            while n_e2e < 40 or time.perf_counter() - t1 < 0.5:
                cnt_e, bags_e = one_structure(ctx, blobs[n_e2e % 4], buf)
                n_e2e += 1
            e2e_ms = (time.perf_counter() - t1) / n_e2e * 1e3
            # breakdown of one structure (separately timed calls; their sum is a little above the loop figure)
            br = {}
            ctx.set_sort_after_pass(False)      # (the breakdown times the sort as a call of its own)
            tt = time.perf_counter(); ctx.set_blob(blobs[1]); br['upload_validate_ms'] = (time.perf_counter() - tt) * 1e3
            tt = time.perf_counter(); cnt_e = ctx.run_launch(args.cutoff, args.vdw_comp, False, 6.0); br['first_pass_ms'] = (time.perf_counter() - tt) * 1e3
            tt = time.perf_counter(); ctx.sort_contacts(); ctx.device_synchronize(); br['device_sort_ms'] = (time.perf_counter() - tt) * 1e3
            tt = time.perf_counter(); bags_e, _ = ctx.fetch_packed(buf); br['download_all_bags_one_copy_ms'] = (time.perf_counter() - tt) * 1e3
            nbytes_up = int(blobs[1].nbytes)
            nbytes_down = int(sum(v.nbytes for b in bags_e.values() for v in b.values()))
            k_aa = bags_e['atom_atom']
            key = k_aa['i'].astype(np.int64) << 32 | k_aa['j'].astype(np.int64)
            assert len(key) == cnt_e['atom_atom'] and bool(np.all(np.diff(key) > 0)), 'end_to_end: the atom-atom bag is not in canonical order'
            e2e_host = (fresh[1], {k: {kk: vv.copy() for kk, vv in v.items() if kk != 'row'} for k, v in bags_e.items()})      # (records for the exporter: 'i' was expanded by the assertion above)
            # pipelined: three contexts, three host threads
            pipelined = None
            if args.inflight > 1:
                ctxs = [_capi.Context(local_rank) for _ in range(args.inflight)]
                for cx in ctxs:
                    cx.set_sort_after_pass(True)
                    cx.set_packed_layout(rows=True)
                bb = [make_buffer(cx) for cx in ctxs]
                per_thread = max(30, n_e2e // args.inflight)
                gate = threading.Barrier(args.inflight + 1)

                def worker(cx, bf, off):
                    for k in range(4):
                        one_structure(cx, blobs[(off + k) % 4], bf)
                    gate.wait()
                    for k in range(per_thread):
                        one_structure(cx, blobs[(off + k) % 4], bf)

                th = [threading.Thread(target=worker, args=(cx, bf, k)) for k, (cx, bf) in enumerate(zip(ctxs, bb))]
                for t in th:
                    t.start()
                gate.wait()
                t2 = time.perf_counter()
                for t in th:
                    t.join()
                pipelined = round((time.perf_counter() - t2) / (per_thread * args.inflight) * 1e3, 4)
                for cx in ctxs:
                    cx.close()
            end_to_end = {'ms_per_structure': round(e2e_ms, 4), 'structures': n_e2e, 'canonical_order': 'atom-atom bag sorted by (i, j) on the device (arp_atom_contacts_sort: radix passes over i — one block for a bag of up to 32 768 records — + per-run rank by j), ring / amide bags by their two ids on the device as well (k_bag_order: one block up to 8192 records; the radix passes beyond)',
                          'ms_per_structure_%d_contexts_in_flight' % max(args.inflight, 1): pipelined,
                          'breakdown_ms': {k: round(v, 4) for k, v in br.items()},
                          'run_arpeggio_on_an_unseen_structure_ms': round(br['first_pass_ms'], 4),
                          'upload_bytes': nbytes_up, 'download_bytes': nbytes_down,
                          'candidate_pairs_per_s': round(st['candidates'] / (e2e_ms * 1e-3), 1),
                          'pack_blob_ms_host': round(pack_ms, 3),
                          'note': 'fresh structure per step: arp_set_blob (one H2D copy from page-locked memory + device-side validation; the ring / amide '
                                  'centre grids and candidate lists are made beside it on a second stream) + static columns + contact grid + search + '
                                  'per-pair kernel with the ring / amide loops (six launches, one stream) + device sort of the atom-atom bag + all five result bags into one '
                                  'page-locked host buffer with one copy, the atom-atom bag in ROWS layout (arp_set_packed_layout: N + 1 row offsets instead of a bgn id per record); pack_blob_ms_host = NumPy packing of a PackedComplex into the blob (done '
                                  'by the producer of the structure, outside the figure)'}
            ctx.set_packed_layout(rows=False)
            ctx.set_complex(pc)     # back to the resident benchmark structure
            ctx.set_grid_reuse(False)
            for _ in range(3):
                step()
        except Exception as exc:   # never lose the main line over the extra measurement
            end_to_end = {'error': repr(exc)}
    # the other single-GPU configurations of BASELINE.json (stand-in whole / ligand, rings): GPU parts here, CPU parts with the baselines below
    other_configs = None
    if world == 1 and args.workload == 'config3' and not args.no_other_configs:
        try:
            other_configs = other_single_gpu_configs(args, local_rank)
        except Exception as exc:   # never lose the main line over the extra measurement
            other_configs = {'error': repr(exc)}
    ctx.device_synchronize()
    t_gpu_legs_done = time.perf_counter()

    # host-only: the exporter of the drop-in class on the result fetched above (JSON records as the reference's get_contacts builds them)
    if end_to_end is not None and e2e_host is not None and 'error' not in end_to_end:
        try:
            from arpeggio_amd.core import export as _export
            pc_e, bag_sorted = e2e_host
            pc_e.ensure_labels()
            tt = time.perf_counter()
            recs = _export.contacts_json(pc_e, bag_sorted, pc_e.component_types)
            end_to_end['get_contacts_ms'] = round((time.perf_counter() - tt) * 1e3, 2)
            end_to_end['get_contacts_records'] = len(recs)
            del recs
            tt = time.perf_counter()
            recs = _export.contacts_json(pc_e, bag_sorted, pc_e.component_types, share_atoms=True)
            end_to_end['get_contacts_share_atoms_ms'] = round((time.perf_counter() - tt) * 1e3, 2)
            del recs
            import tempfile
            with tempfile.TemporaryDirectory() as td:     # the CLI's output file, written by the native formatter
                tt = time.perf_counter()
                _export.write_contacts_json(os.path.join(td, 'out.json'), pc_e, bag_sorted, pc_e.component_types)
                end_to_end['write_json_ms'] = round((time.perf_counter() - tt) * 1e3, 2)
                end_to_end['write_json_bytes'] = os.path.getsize(os.path.join(td, 'out.json'))
            end_to_end['host_export_note'] = 'get_contacts_ms / write_json_ms: host Python / IO on the fetched arrays, outside every GPU figure'
        except Exception as exc:
            end_to_end['host_export_error'] = repr(exc)
    e2e_host = None

    # ---- N > 1: what a scaling record should also say (never `value`) ----------------------------------------------------------
    # (a) wall clock per DISTRIBUTED structure: shard set-up from this rank's home records — upload, faces cut on the device,
    #     arp_shard_exchange_faces over RCCL, merge — + the first pass + the fetch of this rank's five bags; max over ranks
    # (b) the general three-stage pass with a SELECTION (every 40th residue, by global id): arp_shard_exchange_plus and the
    #     all-reduce of the residue sets are inside every timed step
    # (c) at N = 8 with the default workload: BASELINE configs[3] itself, 250 000 atoms per GPU = 2 M atoms
    multi = None
    if world > 1:
        multi = {}
        try:
            from arpeggio_amd import sharding

            def build(atoms, whole, home_records=None):
                if comm_device is None or args.host_halo:      # (debug transports: host buffers through the rendezvous)
                    full_ = synth.slab_config(atoms, world, seed=4)
                    sh_ = sharding.make_shard_distributed(full_, rank, world, transport)
                    sharding.upload_shard(ctx, sh_, whole_structure=whole)
                    return sh_
                return sharding.make_shard_device(ctx, home_records if home_records is not None else synth.slab_home_records(atoms, world, rank, seed=4),
                                                  rank, world, whole_structure=whole)

            def red_max(x):
                return float(rdzv.allreduce(np.array([x], np.float64), 'max')[0])

            def red_sum(x):
                return float(rdzv.allreduce(np.array([x], np.float64), 'sum')[0])

            home_again = None if (comm_device is None or args.host_halo) else home
            ts = []
            for k in range(3):
                sync_all()
                tt = time.perf_counter()
                sh_w = build(args.atoms, True, home_again)
                cnt_w = sharding.run_shard_whole_structure(ctx, args.cutoff, args.vdw_comp, False)
                bags_w, _ = ctx.fetch_packed()
                ts.append(time.perf_counter() - tt)
            multi['wall_clock_per_structure_ms'] = round(red_max(min(ts[1:])) * 1e3, 3)
            multi['wall_clock_per_structure_definition'] = ('shard set-up from resident home records (upload + faces cut on the device + arp_shard_exchange_faces over RCCL + merge) + '
                                                            'first pass + fetch of the rank\'s five bags (device sort, one copy); best of the last two of three, max over ranks')
            multi['shard_setup_breakdown_ms_rank0'] = getattr(sh_w, 'timings_ms', None)
            # (b) staged pass with a selection
            sh_s = build(args.atoms, False, home_again)
            gid = np.asarray(sh_s.global_id, np.int64)
            sel = (((gid // 8) % 40) == 0).astype(np.uint8)
            ctx.set_selection(sel)
            sh_s.sel = sel
            if comm_device is not None and not args.host_halo:
                ex_s = sharding.DeviceExchange(ctx, sh_s)
                step_s = lambda: sharding.run_shard_device(ctx, ex_s, args.cutoff, args.vdw_comp, False)
            else:
                step_s = lambda: sharding.run_shard(ctx, sh_s, transport, args.cutoff, args.vdw_comp, False)
            for _ in range(5):
                cnt_s = step_s()
            sync_all()
            n_s = max(20, min(args.steps, 200))
            tt = time.perf_counter()
            for _ in range(n_s):
                cnt_s = step_s()
            sync_all()
            el_s = red_max(time.perf_counter() - tt)
            st_s = ctx.stats()
            multi['staged_selection_pass'] = {'ms_per_step': round(el_s / n_s * 1e3, 4), 'steps': n_s, 'selection': 'every 40th run of 8 atoms by global id (both atoms of a boundary pair agree on every rank)',
                                              'selected_atoms_all_ranks': int(red_sum(float((sel * np.asarray(sh_s.is_home)).sum()))),
                                              'candidates_all_ranks': red_sum(float(st_s['candidates'])), 'contacts_emitted_all_ranks': red_sum(float(st_s['emitted'])),
                                              'per_step_exchange': 'arp_shard_exchange_plus (grouped ncclSend / ncclRecv of the halo atoms\' selection_plus bits) + arp_shard_reduce_residue_sets (ncclAllReduce MAX) between the three stages of every step'
                                                                   if (comm_device is not None and not args.host_halo) else 'host buffers through the rendezvous (debug transport)'}
            # (c) BASELINE configs[3]
            if world == 8 and args.atoms == 100_000:
                sync_all()
                tt = time.perf_counter()
                sh_4 = build(250_000, True)
                setup4 = red_max(time.perf_counter() - tt)
                step4 = lambda: sharding.run_shard_whole_structure(ctx, args.cutoff, args.vdw_comp, False)
                for _ in range(5):
                    step4()
                sync_all()
                n_4 = max(20, min(args.steps, 100))
                tt = time.perf_counter()
                for _ in range(n_4):
                    step4()
                sync_all()
                el_4 = red_max(time.perf_counter() - tt)
                st_4 = ctx.stats()
                c4 = red_sum(float(st_4['candidates']))
                multi['config4'] = {'workload': 'BASELINE configs[3]: 2 000 000 atoms, 8 x-slabs of 250 000 (+ one-cell halo over RCCL), whole structure',
                                    'ms_per_step': round(el_4 / n_4 * 1e3, 4), 'steps': n_4, 'value': round(c4 * n_4 / el_4, 1), 'unit': 'candidate atom-pairs/s',
                                    'contacts_emitted': red_sum(float(st_4['emitted'])), 'shard_setup_ms_incl_record_generation': round(setup4 * 1e3, 2),
                                    'halo_exchange_ms_rank0': round(getattr(sh_4, 'halo_ms', 0.0), 3)}
            # (d) strong scaling: BASELINE configs[3] itself — ONE 2 M-atom structure — cut into `world` slabs, at every N
            if args.atoms == 100_000 and not args.no_config4 and CONFIG4_ATOMS % world == 0:
                per = CONFIG4_ATOMS // world
                ts4, sh_4s = [], None
                home4 = None if (comm_device is None or args.host_halo) else synth.slab_home_records(per, world, rank, seed=4, box_slabs=CONFIG4_BOX_SLABS)
                full4 = synth.slab_config(per, world, seed=4, box_slabs=CONFIG4_BOX_SLABS) if home4 is None else None      # (debug transports only)
                for k in range(2):
                    sync_all()
                    tt = time.perf_counter()
                    if home4 is None:
                        sh_4s = sharding.make_shard_distributed(full4, rank, world, transport)
                        sharding.upload_shard(ctx, sh_4s, whole_structure=True)
                    else:
                        sh_4s = sharding.make_shard_device(ctx, home4, rank, world, whole_structure=True)
                    sharding.run_shard_whole_structure(ctx, args.cutoff, args.vdw_comp, False)
                    ctx.fetch_packed()
                    ts4.append(time.perf_counter() - tt)
                wall4 = red_max(ts4[-1])
                step4s = lambda: sharding.run_shard_whole_structure(ctx, args.cutoff, args.vdw_comp, False)
                for _ in range(3):
                    step4s()
                sync_all()
                n_4s = max(10, min(args.steps, 50))
                tt = time.perf_counter()
                for _ in range(n_4s):
                    step4s()
                sync_all()
                el_4s = red_max(time.perf_counter() - tt)
                st_4s = ctx.stats()
                c4s = red_sum(float(st_4s['candidates']))
                multi['config4_strong'] = {'workload': config4_workload(world), 'n_gpus': world, 'atoms_per_gpu': per, 'scaling': 'strong',
                                           'ms_per_step': round(el_4s / n_4s * 1e3, 4), 'steps': n_4s, 'value': round(c4s * n_4s / el_4s, 1), 'unit': 'candidate atom-pairs/s',
                                           'step_definition': 'one whole-structure pass of every rank over its resident slab + halo that builds its contact grid; max over ranks',
                                           'contacts_emitted': red_sum(float(st_4s['emitted'])),
                                           'wall_clock_per_structure_ms': round(wall4 * 1e3, 3),
                                           'wall_clock_per_structure_definition': 'shard set-up from resident home records (upload + faces cut on the device + arp_shard_exchange_faces over RCCL + merge) + first pass + fetch of the rank\'s five bags; second of two, max over ranks',
                                           'halo_exchange_ms_rank0': round(getattr(sh_4s, 'halo_ms', 0.0), 3)}
            # back to the headline's shard (nothing below reads device state, but a later leg might)
        except Exception as exc:   # never lose the main line over the extra measurements
            multi['error'] = repr(exc)      # (a rank that fails alone leaves the others in a collective: the rendezvous' time-out ends the run)

    if world == 1 and args.workload == 'config3' and args.atoms == 100_000 and not args.no_other_configs and not args.no_config4:
        try:
            c4 = config4_single_gpu(args, local_rank)
            multi = {'config4_strong': dict(c4, scaling='strong')}
            if other_configs is not None and 'error' not in other_configs:
                other_configs['config4_2m_atoms'] = c4
        except Exception as exc:   # never lose the main line over the extra measurement
            multi = {'config4_strong': {'error': repr(exc)}}
        t_gpu_legs_done = time.perf_counter()

    # max over ranks of the elapsed time, sum over ranks of the work
    cand, acc, emitted = st['candidates'], st['accepted'], st['emitted']
    rccl_ranks_seen = None
    if rdzv is not None:
        elapsed = float(rdzv.allreduce(np.array([elapsed], np.float64), 'max')[0])
        cand_all, acc_all, emitted_all, exp_all = (float(x) for x in rdzv.allreduce(np.array([cand, acc, emitted, st['expand_candidates']], np.float64), 'sum'))
        if comm_device is not None:
            rccl_ranks_seen = ctx.comm_info()[1]
    else:
        cand_all, acc_all, emitted_all, exp_all = float(cand), float(acc), float(emitted), float(st['expand_candidates'])

    if rank != 0:
        rdzv.barrier()      # (rank 0 prints its line before anybody leaves)
        ctx.comm_destroy()
        rdzv.close()
        return

    ms_per_step = elapsed / timed_steps * 1e3
    value = cand_all * timed_steps / elapsed

    # ---------------- roofline of the dominant kernel (rank 0) ----------------
    # The dominant kernel is the one with the longest HIP-event duration in this run; its bytes are SURVEY 8d's compulsory
    # bytes of that stage (survey_8d_bytes: no intermediate pair list).  roofline_all_kernels gives every kernel of the pass
    # under both byte models, roofline_pass the whole pass (140 N + 16 P over ms_per_step), roofline_valu the issue-rate view.
    per_kernel = {k: (v['ms'] / max(v['launches'], 1)) for k, v in ktimes.items() if v['launches']}
    candidates_for_dominant = {k: per_kernel[k] for k in ('bin', 'search', 'sift', 'mark_search') if k in per_kernel}
    dom = max(candidates_for_dominant, key=candidates_for_dominant.get)
    dom_tie = None
    try:      # two kernels within 3 % of each other by HIP events (search and per-pair kernel are): the committed rocprofv3 averages decide
        pj0 = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
        if args.atoms == 100_000 and world == 1 and args.workload == 'config3':
            order = sorted(candidates_for_dominant, key=candidates_for_dominant.get, reverse=True)
            if len(order) > 1 and candidates_for_dominant[order[1]] > 0.97 * candidates_for_dominant[order[0]]:
                ns = {k: pj0['kernels'].get(f'k_{k}', {}).get('rocprof_avg_ns') for k in order[:2]}
                if all(ns.values()):
                    dom = max(order[:2], key=lambda k: ns[k])
                    dom_tie = {k: round(candidates_for_dominant[k], 5) for k in order[:2]}
    except (OSError, ValueError, KeyError):
        pass
    dom_ms = candidates_for_dominant[dom]
    ncell = st['cells']
    n_h = int(pc.h_xyz.shape[0]) if getattr(pc, 'h_xyz', None) is not None else 0

    def n_of(k):
        return st['binned'] if k != 'mark_search' else pc.n_atoms
    b_alg = survey_8d_bytes(dom, n_of(dom), ncell, emitted)
    achieved = b_alg / (dom_ms * 1e-3) / 1e9
    # HBM bytes and VALU instructions per launch from the PMC counters of the last committed rocprofv3 run
    # (profiles/pmc_traffic.json, written by tools/export_profile.py; separate --pmc passes, gfx950 FETCH_SIZE correction applied)
    pj = None
    try:
        pj = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
        if not (args.atoms == 100_000 and world == 1 and args.workload == 'config3'):
            pj = None
    except (OSError, ValueError):
        pj = None
    traffic = traffic_src = None
    committed = None
    if pj is not None and f'k_{dom}' in pj.get('kernels', {}):
        traffic = pj['kernels'][f'k_{dom}'].get('hbm_bytes_per_launch')
        traffic_src = pj.get('source')
        ns = pj['kernels'][f'k_{dom}'].get('rocprof_avg_ns')
        if ns:      # the same figure with rocprofv3's average duration of the committed profile (HIP-event brackets read 1 - 2 us long)
            committed = {'avg_launch_ms': round(ns * 1e-6, 5), 'achieved': round(b_alg / (ns * 1e-9) / 1e9, 2),
                         'frac': round(b_alg / (ns * 1e-9) / 1e9 / HBM_PEAK_GBS, 6), 'source': pj.get('source')}
    # The number that actually bounds these kernels: VALU issue.  SQ_INSTS_VALU of the committed PMC run (per launch) / launch duration
    # against the MEASURED issue rates of this chip (tools/micro/valu_issue.hip -> profiles/round6_valu_issue.json): a wave64
    # instruction of the plain VOP2 / f32-FMA kind issues every ~2.4 cycles per SIMD (SIMD-32 x 2 passes: 1.0e9 /s/SIMD), but packed
    # f32, f64, compares, carries, three-operand integer forms and v_readlane — what the distance loop of k_search and the float64
    # geometry of k_sift are made of — every ~4.2 cycles (0.58e9 /s/SIMD).  `frac` is against the full rate (a lower bound of how busy
    # the VALUs are), `frac_half_rate` against the half rate (an upper bound); through round 5 an assumed 1024 x 2.4 GHz / 4 stood here.
    VALU_PEAK, VALU_PEAK_HALF, valu_src = 1024 * 2.4e9 / 2.0, 1024 * 2.4e9 / 4.0, 'assumed: no profiles/round6_valu_issue.json'
    try:
        vj = json.load(open(os.path.join(ROOT, 'profiles', 'round6_valu_issue.json')))
        VALU_PEAK = float(vj['valu_issue_peak_per_s_chip'])
        VALU_PEAK_HALF = float(vj['valu_issue_half_rate_per_s_per_simd']) * int(vj['simds'])
        valu_src = 'measured: profiles/round6_valu_issue.json (tools/micro/valu_issue.hip, all 1024 SIMDs, 8 waves per SIMD)'
    except (OSError, ValueError, KeyError):
        pass
    roofline_valu = None
    if pj is not None:
        roofline_valu = {'issue_peak_per_s': VALU_PEAK, 'issue_peak_half_rate_per_s': VALU_PEAK_HALF, 'peak_source': valu_src,
                         'source': f'instruction counts from the committed profile, NOT this run ({pj.get("source")}); durations of this run', 'kernels': {}}
        total_insts = 0
        for k_, ms_ in candidates_for_dominant.items():
            pv = pj['kernels'].get(f'k_{k_}', {})
            if 'valu_insts_per_launch' in pv:
                total_insts += pv['valu_insts_per_launch']
                rate = pv['valu_insts_per_launch'] / (ms_ * 1e-3)
                roofline_valu['kernels'][f'k_{k_}'] = {'valu_wave_instructions_per_launch': pv['valu_insts_per_launch'],
                                                       'frac': round(rate / VALU_PEAK, 4), 'frac_half_rate': round(rate / VALU_PEAK_HALF, 4)}
        if total_insts:
            roofline_valu['pass'] = {'valu_wave_instructions_per_pass': total_insts, 'ms_per_step': round(ms_per_step, 4),
                                     'frac': round(total_insts / (ms_per_step * 1e-3) / VALU_PEAK, 4),
                                     'frac_half_rate': round(total_insts / (ms_per_step * 1e-3) / VALU_PEAK_HALF, 4)}
    roofline = {'kernel': f'k_{dom}', 'bound': 'hbm', 'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBS, 6), 'traffic': traffic,
                'traffic_source': (None if traffic_src is None else f'committed profile, NOT this run: {traffic_src}'),
                'algorithmic_bytes_per_launch': int(b_alg),
                'bytes_model': {'bin': 'SURVEY 8d bin + sort / scatter passes: 92 N', 'search': 'SURVEY 8d: each sorted record once, 32 N + 4 (C + 1); the 8 B / pair list it writes is an intermediate of the split design, not counted',
                                'sift': 'SURVEY 8d: side arrays + output records, 16 N + 16 P; the pair list and the record gathers are not counted',
                                'mark_search': '32 N + 4 (C + 1)'}[dom],
                'avg_launch_ms': round(dom_ms, 5), 'with_rocprofv3_duration_of_the_committed_profile': committed,
                'dominant_kernel_rule': 'longest average HIP-event duration among the kernels of the pass in this run (kernel_ms); two kernels within 3 % of each other: the one '
                                        'with the longer rocprofv3 average in the committed profile' + (' — applied here: %s' % dom_tie if dom_tie else ''),
                'note': 'VALU-issue-bound geometry kernels (roofline_valu); the HBM fraction is small by construction: register-tiled pair tests '
                        'move ~35 MB per 100 k-atom pass (SURVEY 8d)'}

    # the same figure for every big kernel of the pass (informational; `roofline` above is the dominant one)
    roofline_all = {}
    for k, ms in candidates_for_dominant.items():
        b8 = survey_8d_bytes(k, n_of(k), ncell, emitted)
        bi = algorithmic_bytes(k, n_of(k), ncell, emitted, n_h)
        roofline_all[f'k_{k}'] = {'avg_launch_ms': round(ms, 5), 'survey_8d_bytes_per_launch': int(b8),
                                  'frac': round(b8 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                                  'bytes_as_implemented': int(bi), 'frac_as_implemented': round(bi / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)}
    b_pass = 140 * st['binned'] + 16 * emitted
    roofline_pass = {'bytes_per_pass': int(b_pass), 'bytes_model': 'SURVEY 8d: B_alg = 140 N + 16 P (grid build included)', 'ms_per_step': round(ms_per_step, 4),
                     'achieved': round(b_pass / (ms_per_step * 1e-3) / 1e9, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': round(b_pass / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)} if world == 1 else None

    # ---------------- CPU baseline: the C oracle on one host core, bounded sample ----------------
    gpu_legs_s = t_gpu_legs_done - t_gpu_legs_begin
    cpu = None
    if not args.no_cpu_baseline and world == 1:   # (rank 0 at N = 1 only, as the bench contract says)
        import oracle
        ns = min(args.cpu_sample_atoms, args.atoms)
        spc = synth.config3(ns, seed=3) if (world > 1 or (ns != args.atoms and args.workload == 'config3')) else pc
        oc = oracle.OracleComplex(spc)
        # The GPU step of a whole-structure run is the 5 A contact search, the per-pair evaluation and the ring / amide loops;
        # selection_plus = selection needs no 6 A search there.  The baseline times exactly that work (one search, one
        # evaluation per pass: the capacity comes from an untimed first call); what the reference's own CPU path does on top —
        # the 6 A search_all of _make_selection, I:1420, which it runs whatever the selection is — is timed beside it.
        oc.make_selection(None)
        n_known = len(oc.atom_contacts(args.cutoff, args.vdw_comp, False)['i'])
        passes, cpu_s, sel_s, cand_cpu = 0, 0.0, 0.0, 0
        while cpu_s < 10.0 and passes < 200:      # ~10 s of single-core work
            t0 = time.perf_counter()
            r = oc.atom_contacts(args.cutoff, args.vdw_comp, False, cap_hint=n_known)
            oc.plane_plane(); oc.group_group(); oc.group_plane()
            cpu_s += time.perf_counter() - t0
            cand_cpu += int(r['stats'][0])
            passes += 1
            if passes <= 3:
                t0 = time.perf_counter()
                oc.make_selection(None)
                sel_s += time.perf_counter() - t0
        sel_ms = sel_s / min(passes, 3) * 1e3
        cpu = {'value': round(cand_cpu / cpu_s, 1), 'unit': 'candidate atom-pairs/s', 'cores': 1, 'kind': 'port',
               'sample': f'{passes} passes of the C oracle (oracle/ref_c.c: ONE 5 A grid search_all + per-pair SIFt + plane-plane / group loops = the '
                         f'work of the GPU step; gcc -O2, 1 thread) on the same {spc.n_atoms}-atom synthetic structure, {cpu_s:.1f} s total, '
                         f'{cpu_s / passes * 1e3:.0f} ms per pass; the O(R*N) brute-force atom-plane loop of the oracle is left out',
               'ms_per_structure': round(cpu_s / passes * 1e3, 2), 'host_cores_visible': os.cpu_count(),
               'with_expansion_search_6A': {'ms_per_structure': round(cpu_s / passes * 1e3 + sel_ms, 2),
                                            'value': round(cand_cpu / passes / (cpu_s / passes + sel_ms * 1e-3), 1),
                                            'note': 'plus the 6 A search_all of _make_selection (I:1420), which the reference runs also for a whole-structure '
                                                    'selection and the GPU pass skips (selection_plus = selection)'}}

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

    if other_configs is not None and 'error' not in other_configs and not args.no_cpu_baseline:
        try:
            other_configs_cpu_baselines(args, other_configs)
        except Exception as exc:
            other_configs['cpu_baseline_error'] = repr(exc)
    if cpu is not None and multi and 'config4_strong' in multi and 'pairs' in multi['config4_strong'] and world == 1:
        c4 = multi['config4_strong']
        cb = {'value': cpu['value'], 'unit': 'candidate atom-pairs/s', 'cores': 1, 'kind': 'port',
              'ms_per_structure_extrapolated': round(c4['pairs']['candidates'] / cpu['value'] * 1e3, 1),
              'sample': 'SAMPLED: the C oracle on one host core over the headline\'s 100 000-atom structure of the same density and cutoff (cpu_baseline of this line: '
                        f"{cpu['ms_per_structure']} ms per pass); a whole 2 M-atom pass at that rate would take the extrapolated figure — it was not run"}
        c4['cpu_baseline'] = cb
        if other_configs is not None and 'config4_2m_atoms' in other_configs:
            other_configs['config4_2m_atoms']['cpu_baseline'] = cb
    if ligand_pass is None and other_configs is not None:
        ligand_pass = other_configs.get('standin_ligand')
    e2e_ms_sorted = (end_to_end or {}).get('ms_per_structure')
    line = {
        'metric': 'evaluated atom-pairs/s', 'value': round(value, 1), 'unit': 'candidate atom-pairs/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4),
        'timed_steps': timed_steps, 'timed_region_s': round(elapsed, 3), 'setup_passes': SETUP_PASSES, 'untimed_trial_steps': args.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64 distance test / f32+f64 SIFt',
        'data': 'synthetic',
        'config': {'workload': workload, 'atoms_per_gpu': args.atoms, 'cutoff_A': args.cutoff, 'vdw_comp': args.vdw_comp,
                   'parallelism': f'slab{world}' if world > 1 else 'single',
                   'sharding': (None if world == 1 else f'{args.atoms * world} atoms, box elongated along x, {world} x-slabs of {args.atoms} atoms (one config-3 cube each'
                                                        f'{"; BASELINE configs[3]" if args.atoms * world == 2_000_000 else ""}), one-cell halo over RCCL')},
        'step_definition': 'one whole run_arpeggio pass over the resident structure that BUILDS its contact grid (k_compact_atoms + k_search + k_sift_planes, '
                           'arp_set_grid_reuse(0)); nothing a pass made is reused by the next one',
        # wall clock per structure = a FRESH structure end to end, canonical order included (end_to_end); null when that leg was skipped
        'wall_clock_per_structure_ms': e2e_ms_sorted,
        'wall_clock_per_structure_definition': 'end_to_end.ms_per_structure: upload + first pass + device sort + one-copy fetch of all five bags, new structure every step',
        'run_arpeggio_on_an_unseen_structure_ms': (end_to_end or {}).get('run_arpeggio_on_an_unseen_structure_ms'),
        'pairs': {'candidates': cand_all, 'accepted': acc_all, 'contacts_emitted': emitted_all,
                  'expansion_candidates_6A': exp_all, 'bags': counts},
        'accepted_pairs_per_s': round(acc_all * timed_steps / elapsed, 1),
        'kernel_ms': {k: round(v, 5) for k, v in per_kernel.items()},
        'kernel_launches_per_step': {k: v['launches'] / args.steps for k, v in ktimes.items() if v['launches']},
        'ms_per_step_profiled_pass': round(elapsed_profiled / args.steps * 1e3, 4),
        'roofline_all_kernels': roofline_all,
        'throughput_several_in_flight': in_flight,
        'cpu_baseline_all_cores': cpu_mc,
        'cpu_baseline_python': cpu_py,
        'host_us_per_step': {k: round(v, 1) for k, v in host_times.items() if k != 'passes'} if world == 1 else None,
        'launch_mode': 'three launches on one HIP stream per pass: k_compact_atoms (kernel_ms.bin: the contact grid of the pass), k_search, k_sift_planes (per-pair evaluation + ring/amide loops; kernel_ms: search / sift); the last launch publishes the counters; one host wait per step (pinned completion word, bounded spin); kernel_ms from a second pass of the same steps with HIP events (each bracket adds ~4 us to a small kernel; rocprofv3 averages are in profiles/)',
        'per_step_exchange': (None if world == 1 else ('selection_plus halo bits (P2P) + residue sets (all-reduce MAX) over RCCL' if args.staged_exchange else 'none: whole-structure selection, every rank knows selection_plus and the residue sets (DESIGN.md 6)')),
        'rccl_ranks_seen': rccl_ranks_seen,
        'multi_gpu': multi,
        'halo_exchange_ms': round(halo_ms, 3), 'halo_exchange_bytes_sent_rank0': halo_bytes, 'shard_setup_ms': round(shard_setup_ms, 2), 'shard_setup_breakdown_ms': shard_timings,
        'halo_exchange': (halo_note if world > 1 else None),
        'scaling_note': (None if world == 1 else f'weak scaling: every GPU owns one config-3 cube of {args.atoms} atoms (+ a one-cell halo) — the workload of the N = 1 line with the same --atoms'),
        'setup_s': round(gen_s, 2), 'home_atoms_rank0': n_local_home,
        'gpu_legs_s': round(gpu_legs_s, 2),
        'end_to_end': end_to_end,
        'end_to_end_ms_per_structure': e2e_ms_sorted,
        'get_contacts_ms': (end_to_end or {}).get('get_contacts_ms'),
        'roofline_valu': roofline_valu,
        'roofline_pass': roofline_pass,
        'pass_with_grid_kept': grid_kept, 'ligand_selection_pass': ligand_pass,
        'configs': other_configs,
        'roofline': roofline, 'cpu_baseline': cpu,
    }
    if world > 1 and comm_device is not None and rccl_ranks_seen != world:
        line['error'] = f'the RCCL communicator holds {rccl_ranks_seen} ranks, not {world}: this is not an {world}-GPU measurement'
    print(json.dumps(line), flush=True)
    if rdzv is not None:
        rdzv.barrier()
        ctx.comm_destroy()
        rdzv.close()


if __name__ == '__main__':
    main()
