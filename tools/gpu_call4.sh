# Round 2, GPU call 4: K4f layout / store variants, the rewritten bench.py (N = 1
# full line, self-launched N = 2 smoke), host profile of the drop-in API loop.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c4
O=gpurun_out/c4
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_spectrum_gpu.py tests/test_bench_gpu.py tests/test_eval_gpu.py -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.txt; tail -15 $O/pytest_gpu.txt
run() {  # name workload lib
  local lib=""; [ -n "$3" ] && lib="$GRAFT_REPO_ROOT/build/variants/libwb2hip_$3.so"
  WB2HIP_LIB=$lib timeout 200 python bench.py --workload $2 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $O/$1.json
  python - <<PY
import json
try:
  d = json.load(open('$O/$1.json'))
  r = d['roofline']
  print('%-26s %-14s kernel_ms %.4f  GB/s %.0f  frac %.3f  value %.4g' % ('$1', '$2', r['kernel_ms'], r['achieved'], r['frac'], d['value']))
except Exception as e:
  print('$1 FAILED', e)
PY
}
{
for rep in 1 2; do
for v in "" nopad nowide pf0 pf3 blk1536 blk4096; do
  run specmean_${v:-main}_$rep spectrum_mean "$v"
  run spec_${v:-main}_$rep spectrum "$v"
done
done
} 2>&1 | tee $O/summary.txt
timeout 600 python bench.py --pcie > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json; tail -5 $O/bench_default.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2>/dev/null; cut -c1-400 $O/bench_driver.json
timeout 300 python tools/api_profile.py 2>&1 | grep -v amdgpu | head -60 > $O/api_profile.txt; head -50 $O/api_profile.txt
