# Round 3, A/B 1: K4f LDS reads as single ds_read_b64 (default now) vs the
# compiler's ds_read2_b64 pairs (read2) vs single reads + 4 waves/SIMD (w4);
# then the GPU tests of the new host code (baseline substitutions, dtypes).
#   gpurun --timeout 900 -- 'bash tools/ab_round3a.sh'
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab3a
mkdir -p $O
: > $O/summary.txt
V=$GRAFT_REPO_ROOT/build/variants
run() {
  local name=$1 wl=$2; shift 2
  local lib=""; [ "$name" != default ] && lib=$V/libwb2hip_$name.so
  WB2HIP_LIB=$lib timeout 120 python bench.py --workload $wl --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('$name', '$wl', 'step_ms=%.4f kernel_ms=%.4f value=%.4g frac=%.3f' % (d['ms_per_step'], r['kernel_ms'], d['value'], r['frac']))
" | tee -a $O/summary.txt
}
for rep in 1 2; do
  for wl in spectrum spectrum_mean spectrum_materialized; do
    for n in read2 default w4; do run $n $wl; done
  done
done
timeout 400 python -m pytest -x -q -m gpu tests/test_evalall.py tests/test_spectrum_gpu.py tests/test_threads_gpu.py "tests/test_bench_launch_gpu.py" tests/test_reference_vectors.py tests/test_eval_gpu.py 2>&1 | tail -15 | tee $O/pytest.txt
