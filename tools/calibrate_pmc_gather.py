#!/usr/bin/env python3
"""Known-byte GATHERS for calibrating FETCH_SIZE (run under `rocprofv3 --pmc FETCH_SIZE`): every 128-byte row (one L2
line) / every 32-byte row (a sift record) of a 256 MiB table read exactly once in random order (torch index_select).
True HBM reads per launch: the table once + the int64 index."""
import torch

NBYTES = 256 * 1024 * 1024
for row_floats in (32, 8):
    n = NBYTES // (4 * row_floats)
    src = torch.empty((n, row_floats), dtype=torch.float32, device='cuda').normal_()
    idx = torch.randperm(n, device='cuda')
    torch.cuda.synchronize()
    for _ in range(5):
        out = torch.index_select(src, 0, idx)      # reads NBYTES + 8 n, writes NBYTES
    torch.cuda.synchronize()
    print('row bytes', 4 * row_floats, 'table bytes', NBYTES, 'index bytes', 8 * n)
    del src, idx, out
