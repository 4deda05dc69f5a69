# A/B of the K2 / time-accumulate / K4f-epilogue variants built by
# tools/build_variant.py (k2m16, k2m8, sf1, all) against the default library,
# same box, interleaved; then the parity tests through the combined variant.
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab2b
mkdir -p $O
V=$GRAFT_REPO_ROOT/build/variants
run() {  # lib-name workload extra-args...
  local name=$1 wl=$2; shift 2
  local lib=""; [ "$name" != default ] && lib=$V/libwb2hip_$name.so
  WB2HIP_LIB=$lib timeout 300 python bench.py --workload $wl "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('$name', '$wl', 'step_ms=%.4f kernel_ms=%.4f value=%.4g frac=%.3f' % (d['ms_per_step'], r['kernel_ms'], d['value'], r['frac']))
" | tee -a $O/summary.txt
}
for rep in 1 2; do
  for n in default k2m16 k2m8; do run $n ensemble --steps 60 --warmup 10; done
  for n in default sf1; do
    run $n spectrum_mean --steps 100 --warmup 10
    run $n spectrum --steps 100 --warmup 10
  done
  for n in default k2m16; do
    run $n deterministic --steps 200 --warmup 20 --no-cpu-baseline --no-api --no-full-suite
  done
done
run k2m8 deterministic --steps 200 --warmup 20 --no-cpu-baseline --no-api --no-full-suite
WB2HIP_LIB=$V/libwb2hip_all.so timeout 600 python -m pytest -x -q -m gpu tests/test_det_gpu.py tests/test_ens_gpu.py tests/test_spectrum_gpu.py tests/test_eval_gpu.py tests/test_golden_fixtures.py tests/test_bench_launch_gpu.py tests/test_edge_gpu.py 2>&1 | tail -4 | tee $O/pytest_all_variant.txt
