import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arpeggio_amd import _capi, synth, sharding
mode = sys.argv[1]
c = _capi.Context(0)
c.comm_init(0, 1, _capi.Context.comm_unique_id())
if mode in ('ops', 'ops_nodestroy'):
    full = synth.slab_config(3000, 2, seed=6)
    ds = sharding.make_shard_device(c, full, 0, 1, sel=(full.res_id % 7 == 3).astype(np.uint8))
    ex = sharding.DeviceExchange(c, ds)
    print(sharding.run_shard_device(c, ex))
if mode in ('plain', 'ops'):
    c.comm_destroy()
c.close()
print('DONE', mode)
