set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/pytest_gpu.txt; cat gpurun_out/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
export TMPDIR=/tmp
for w in deterministic ensemble spectrum; do
  extra=""; [ $w != deterministic ] && extra="--workload $w"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$w -o run -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline $extra > $GRAFT_REPO_ROOT/gpurun_out/prof_$w.log 2>&1)
  f=$(find gpurun_out/prof_$w -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -12 "$f" > gpurun_out/stats_$w.csv
  tail -1 gpurun_out/prof_$w.log | cut -c1-300
  rm -rf gpurun_out/prof_$w
done
timeout 200 python bench.py --workload ensemble --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_ens.json
timeout 200 python bench.py --workload spectrum --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_spec.json
timeout 200 python bench.py --workload spectrum_mean --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_spec_mean.json
timeout 300 python bench.py --pcie 2>/dev/null | tail -1 > gpurun_out/bench_det.json
timeout 200 python tools/api_throughput.py 2>&1 | grep -v amdgpu | head -5 > gpurun_out/api_throughput.txt
timeout 200 python tools/axis_bench.py 2>&1 | grep -v amdgpu > gpurun_out/axis_bench.txt
cat gpurun_out/bench_det.json | cut -c1-300; cat gpurun_out/api_throughput.txt; cat gpurun_out/bench_spec_mean.json | cut -c1-200; cat gpurun_out/bench_ens.json | cut -c1-200; cat gpurun_out/bench_spec.json | cut -c1-200; cat gpurun_out/axis_bench.txt
