set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest.log 2>&1; grep -E "passed|failed|error" gpurun_out/final/pytest.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
timeout 400 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; tail -c 400 gpurun_out/final/bench.json | head -c 10 >/dev/null
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench_driver.json 2>> gpurun_out/final/bench.err
timeout 300 python bench.py --atoms 250000 --no-other-configs > gpurun_out/final/bench_250k.json 2>> gpurun_out/final/bench.err
timeout 300 python bench.py --atoms 1000000 --steps 100 --no-other-configs > gpurun_out/final/bench_1m.json 2>> gpurun_out/final/bench.err
timeout 300 python bench.py --workload standin --no-other-configs > gpurun_out/final/standin_bench.json 2>> gpurun_out/final/bench.err
timeout 300 python bench.py --workload standin --batch 64 --no-other-configs > gpurun_out/final/batch64_bench.json 2>> gpurun_out/final/bench.err
timeout 300 python bench.py --workload standin --batch 8 --no-other-configs > gpurun_out/final/batch8_bench.json 2>> gpurun_out/final/bench.err
timeout 300 python tools/rings_bench.py > gpurun_out/final/rings_bench.json 2>> gpurun_out/final/bench.err
timeout 300 python tools/small_bench.py --n-res 8000 > gpurun_out/final/chain_bench.json 2>> gpurun_out/final/bench.err
for k in 1 2 3; do python tools/first_pass_probe.py --tag final --e2e; done > gpurun_out/final/first_pass.jsonl 2>> gpurun_out/final/bench.err
python tools/first_pass_probe.py --tag final_standin --workload standin --e2e >> gpurun_out/final/first_pass.jsonl 2>> gpurun_out/final/bench.err
bash tools/profile.sh r6a > gpurun_out/final/profile.log 2>&1
bash tools/profile.sh r6a_1m "--atoms 1000000 --steps 10" > gpurun_out/final/profile_1m.log 2>&1
bash tools/profile.sh r6a_250k "--atoms 250000 --steps 20" > gpurun_out/final/profile_250k.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fresh_final; timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d gpurun_out/fresh_final -o t -- python tools/fresh_probe.py e2e > gpurun_out/fresh_final/log 2>&1
python tools/fresh_timeline.py $(find gpurun_out/fresh_final -name "t_kernel_trace.csv") $(find gpurun_out/fresh_final -name "t_memory_copy_trace.csv") > gpurun_out/final/e2e_timeline.txt
for f in gpurun_out/final/*bench*.json; do echo "$f: $(tail -1 $f | cut -c1-160)"; done
cat gpurun_out/final/first_pass.jsonl | cut -c1-200
tail -5 gpurun_out/final/bench.err
