cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for a in 250000 1000000; do
  out=gpurun_out/prof_size_$a; mkdir -p $out
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python bench.py --atoms $a --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end --inflight 1 --min-seconds 0 > $out/trace.log 2>&1 < /dev/null
done
out=gpurun_out/prof_fresh; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python tools/fresh_probe.py > $out/trace.log 2>&1 < /dev/null
out=gpurun_out/prof_standin; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python bench.py --workload standin --steps 200 --warmup 3 --no-cpu-baseline --no-end-to-end --inflight 1 --min-seconds 0 > $out/trace.log 2>&1 < /dev/null
find gpurun_out/prof_size_250000 gpurun_out/prof_size_1000000 gpurun_out/prof_fresh gpurun_out/prof_standin -name "*kernel_stats.csv"
