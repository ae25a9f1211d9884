#!/bin/bash
# Kernel-trace + PMC passes of bench.py under rocprofv3 (GPU box only).  Usage: tools/profile.sh <tag> ["extra bench flags"]
# Counters are collected in their own runs (one --pmc group per pass), as the MI355X guide prescribes; the profiled
# command is the resident-pass part of bench.py (no end-to-end leg, no CPU baseline, one context).
set -u
tag=${1:-run}
extra=${2:-}
out=gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end --no-other-configs --inflight 1 --min-seconds 0 $extra"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- $B > "$out/trace.log" 2>&1 < /dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/fetch" -o f -- $B > "$out/fetch.log" 2>&1 < /dev/null
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/write" -o w -- $B > "$out/write.log" 2>&1 < /dev/null
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d "$out/sq" -o s -- $B > "$out/sq.log" 2>&1 < /dev/null
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d "$out/sq2" -o s -- $B > "$out/sq2.log" 2>&1 < /dev/null
timeout 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$out/tc" -o c -- $B > "$out/tc.log" 2>&1 < /dev/null
find "$out" -name "*.csv" | head -40
grep -h '^{' "$out"/trace.log | tail -1 | cut -c1-300
