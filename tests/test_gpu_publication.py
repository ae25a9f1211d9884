"""The end-of-pass publication under stress (round 4 found an intermittent `arp_run_launch -3` on a fresh box by luck: a
cross-kernel publication race, fixed in 51a34f9).  The hand-rolled protocol — relaxed system-scope stores of the counter block to
pinned memory ordered by s_waitcnt vmcnt(0), list counts handed between the kernels of two streams, the per-pair kernel's counted
vmcnt waits — is exercised here the way tools/pub_stress.py did by hand: many thousand passes alternating structures of
different sizes, a selection and a batch, on two contexts driven by two host threads, every counter of every pass compared
with the first pass of its kind."""
import threading

import numpy as np
import pytest

from arpeggio_amd import synth

pytestmark = pytest.mark.gpu


def _kinds(_capi):
    """(name, make context) for the kinds of pass that alternate; a context per kind and thread."""
    big = synth.config3(60_000, seed=3)
    prot = synth.proteinlike(seed=1)
    small = [synth.proteinlike(n_res=60, seed=12, n_waters=20, id='p1'), synth.make_synthetic(800, seed=7, box=(60, 20, 20), n_rings=0, n_amides=0),
             synth.make_synthetic(0, seed=6, box=(25, 25, 25), n_rings=40, n_amides=30)]
    lig = (np.asarray(prot.res_seq)[np.asarray(prot.res_id)] == 508).astype(np.uint8)

    def whole(pc, reuse):
        def make():
            c = _capi.Context(0)
            c.set_complex(pc)
            c.set_grid_reuse(reuse)
            return c, (lambda: (dict(c.run_launch(5.0, 0.1, False, 6.0)), c.stats()))
        return make

    def ligand():
        c = _capi.Context(0)
        c.set_complex(prot)
        c.set_selection(lig)
        return c, (lambda: (dict(c.run_launch(5.0, 0.1, False, 6.0)), c.stats()))

    def batch():
        c = _capi.Context(0)
        c.set_batch(small)
        return c, (lambda: (dict(c.run_launch(5.0, 0.1, False, 6.0)), c.stats()))
    return [('config3-60k, grid rebuilt', whole(big, False)), ('stand-in, grid kept', whole(prot, True)), ('stand-in, ligand selection', ligand),
            ('batch of three', batch), ('stand-in, grid rebuilt', whole(prot, False))]


def test_every_pass_publishes_the_counters_of_its_kind():
    from arpeggio_amd import _capi
    kinds = _kinds(_capi)
    bad, done = [], [0, 0]

    def worker(t, n):
        ctxs = []
        try:
            for name, make in kinds[t:] + kinds[:t]:
                c, run = make()
                for _ in range(3):
                    want = run()
                ctxs.append((name, c, run, want))
            for k in range(n):
                name, c, run, want = ctxs[k % len(ctxs)]
                got = run()
                if got != want:
                    bad.append((t, k, name, got, want))
                    if len(bad) > 5:
                        break
                done[t] += 1
        except Exception as exc:      # (an error code of the library is a failure of the publication as well: -3 was the symptom)
            bad.append((t, -1, repr(exc)))
        finally:
            for _, c, _, _ in ctxs:
                c.close()

    n = 10_000
    ths = [threading.Thread(target=worker, args=(t, n)) for t in range(2)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not bad, bad[:2]
    assert done == [n, n]
