cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/k2
O=$GRAFT_REPO_ROOT/gpurun_out/k2
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_det_gpu.py tests/test_ens_gpu.py tests/test_bench_launch_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt
for w in deterministic ensemble; do
  extra="--no-full-suite --no-api"; [ $w != deterministic ] && extra="--workload $w"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -o run -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline $extra > $O/prof_$w.log 2>&1)
  f=$(find $O/prof_$w -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -6 "$f" > $O/r02_${w}_kernel_stats.csv
  tail -1 $O/prof_$w.log | cut -c1-250
  rm -rf $O/prof_$w
  cut -d, -f1-4 $O/r02_${w}_kernel_stats.csv | cut -c1-60,150-260 | head -4
done
