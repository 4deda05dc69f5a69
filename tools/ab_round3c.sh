# Round 3, A/B 3: K4f with derived twiddle powers (120 VGPRs = 4 waves per SIMD, no
# spills) now that the LDS accesses are single instructions
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab3c
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
WB2HIP_LIB=$V/libwb2hip_twp.so timeout 300 python -m pytest -x -q -m gpu tests/test_spectrum_gpu.py tests/test_reference_vectors.py -k "spectrum or Spectrum" 2>&1 | grep -E "passed|failed" | tee $O/pytest.txt
for rep in 1 2; do
  for wl in spectrum spectrum_mean spectrum_materialized; do
    for n in default twp twp4; do run $n $wl; done
  done
done
