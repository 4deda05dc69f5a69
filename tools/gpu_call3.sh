# Round 2, GPU call 3: full parity suite (new bench-launch / f32-latitude tests,
# tightened fuzz tolerances) + K4f scheduling / register sweep + lean K3.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c3
O=gpurun_out/c3
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/pytest_gpu.txt; tail -25 $O/pytest_gpu.txt
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
for v in "" k3_r01; do
  run ens_${v:-main}_$rep ensemble "$v"
done
for v in "" static_main pf0_tw0 static_pf0_tw0 pf0_tw12 pf0_tw15 pf3_tw12 pf3_tw15 pf0_tw15_mw5; do
  run specmean_${v:-main}_$rep spectrum_mean "$v"
  run spec_${v:-main}_$rep spectrum "$v"
done
done
} 2>&1 | tee $O/summary.txt
