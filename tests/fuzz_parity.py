"""Long-running differential test (not collected by pytest): the body of test_random_parameters_and_selections and of
test_random_shards_device_vs_host for as many fresh seeds as fit in the given time.

    python tests/fuzz_parity.py [seconds] [first_seed] [blob]

`blob`: the parameters body uploads every structure as ONE blob (arp_set_blob: candidate lists, static columns and their
spatial order made with the upload; the sort enqueued by the pass itself) instead of the classic setters.

Prints one line per failing seed (with the assertion) and a summary; exit code 1 if any seed failed."""
import os
import sys
import time
import traceback

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
    from arpeggio_amd import _capi
    import test_gpu_parity as tp
    import test_gpu_shard_device as ts
    blob = 'blob' in sys.argv[3:]

    def make_ctx():
        c = _capi.Context(0)
        if blob:
            c.set_complex = lambda pc, c=c: c.set_blob(_capi.pack_blob(pc))
            c.set_sort_after_pass(True)
        return c
    ctx = make_ctx()
    t0, done, failed = time.time(), 0, []
    while time.time() - t0 < budget:
        for name, fn, arg in (('parameters', tp.test_random_parameters_and_selections, ctx),
                              ('shards', ts.test_random_shards_device_vs_host, _capi)):
            try:
                fn(arg, seed)
            except Exception as e:      # noqa: BLE001 — every kind of failure is a finding
                failed.append((name, seed))
                print('FAIL', name, seed, repr(e)[:300], flush=True)
                traceback.print_exc(limit=3)
                ctx = make_ctx()
        done += 1
        seed += 1
    print('fuzz: %d seeds x 2 bodies in %.0f s, %d failures %s' % (done, time.time() - t0, len(failed), failed[:20]))
    return 1 if failed else 0


if __name__ == '__main__':
    sys.exit(main())
