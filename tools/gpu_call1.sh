# Round 2, GPU call 1: parity of the rewritten K4f / dieted K3, then A/B of the
# kernel variants (build/variants/*.so, selected with WB2HIP_LIB).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c1
O=gpurun_out/c1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
run() {  # name workload lib
  local lib=""; [ -n "$3" ] && lib="$GRAFT_REPO_ROOT/build/variants/libwb2hip_$3.so"
  WB2HIP_LIB=$lib timeout 200 python bench.py --workload $2 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $O/$1.json
  python - <<PY
import json
try:
  d = json.load(open('$O/$1.json'))
  r = d['roofline']
  print('%-22s %-14s kernel_ms %.4f  GB/s %.0f  frac %.3f  value %.4g' % ('$1', '$2', r['kernel_ms'], r['achieved'], r['frac'], d['value']))
except Exception as e:
  print('$1 FAILED', e)
PY
}
for rep in 1 2; do
for v in "" k4_r01 k4_nopf k4_pf_tw0 k4_pf_tw12 k4_pf_tw15 k4_noasm k4_blk768 k4_blk1024 k4_blk4096; do
  run spec_${v:-main}_$rep spectrum "$v"
  run specmean_${v:-main}_$rep spectrum_mean "$v"
done
for v in "" k3_r01 k3_mw4; do
  run ens_${v:-main}_$rep ensemble "$v"
done
done 2>&1 | tee $O/summary.txt
