# Round 3, A/B 4: LATSEG with interleaved latitudes per segment (adjacent rows read
# concurrently) vs contiguous segments
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab3d
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
WB2HIP_LIB=$V/libwb2hip_ileave.so timeout 300 python -m pytest -x -q -m gpu tests/test_spectrum_gpu.py 2>&1 | grep -E "passed|failed" | tee $O/pytest.txt
for rep in 1 2 3; do
  for n in default ileave; do run $n spectrum; done
done
