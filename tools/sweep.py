#!/usr/bin/env python3
"""Run bench.py under several values of one environment variable (tuning aid, GPU box only)."""
import json
import os
import subprocess
import sys

var, values = sys.argv[1], sys.argv[2].split(',')
extra = sys.argv[3:]
for v in values:
    env = dict(os.environ)
    env[var] = v
    out = subprocess.run([sys.executable, 'bench.py', '--no-cpu-baseline', '--steps', '200', '--warmup', '5'] + extra,
                         env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith('{')]
    if not line:
        print(var, v, 'FAILED', out.stderr[-400:])
        continue
    d = json.loads(line[-1])
    print(f'{var}={v}: value={d["value"]:.4g} ms/step={d["ms_per_step"]} kernels={d["kernel_ms"]}')
