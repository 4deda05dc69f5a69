# K1 ring form (LDS-DMA) against the batch form: parity tests, then timings
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ring
export WB2HIP_K1_RING=3
timeout 900 python -m pytest tests/test_det_gpu.py tests/test_bench_launch_gpu.py tests/test_suite_step_gpu.py tests/test_pairs_gpu.py -x -q -m gpu 2>&1 | tail -4
unset WB2HIP_K1_RING
for rep in 1 2; do
for cfg in "0 0" "2 0" "3 0" "4 0" "3 3" "4 3" "5 3" "4 2"; do
  set -- $cfg
  echo "== ring=$1 waves=$2 rep=$rep"
  WB2HIP_K1_RING=$1 WB2HIP_K1_RING_WAVES=$2 timeout 300 python tools/pair_bench.py --window 8 --chunks 2 --reps 5 2>&1 | tail -3
  WB2HIP_K1_RING=$1 WB2HIP_K1_RING_WAVES=$2 timeout 300 python bench.py --no-pmc --no-cpu-baseline --no-secondary --no-api --no-pcie --no-full-suite 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline', round(d['value']/1e9,1), 'G', d['ms_per_step'], d.get('roofline',{}).get('frac'))"
done; done 2>&1 | tee gpurun_out/ring/ab.txt
