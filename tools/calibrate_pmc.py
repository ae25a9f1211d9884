#!/usr/bin/env python3
"""Known-byte streams for calibrating FETCH_SIZE / WRITE_SIZE (run under `rocprofv3 --pmc ...`): a device-to-device
copy of NBYTES through torch's vectorised elementwise kernel (16 bytes per lane) and a float32 read-only reduction."""
import torch

NBYTES = 256 * 1024 * 1024
a = torch.empty(NBYTES // 4, dtype=torch.float32, device='cuda').normal_()
b = torch.empty_like(a)
torch.cuda.synchronize()
for _ in range(5):
    b.copy_(a)            # reads NBYTES, writes NBYTES
torch.cuda.synchronize()
for _ in range(5):
    s = a.sum()           # reads NBYTES
torch.cuda.synchronize()
print('bytes per launch', NBYTES)
