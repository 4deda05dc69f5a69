# Round 3: the SGPR-addressed K1 instantiation for unaligned float32 rows
# (lon-lat layout): parity + A/B (WB2HIP_SGPR_UNALIGNED=0 = plain addressing),
# and the variants leg with its ramp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q
mkdir -p $O
: > $O/summary.txt
timeout 900 python -m pytest -x -q -m gpu tests/test_det_gpu.py tests/test_fuzz_gpu.py tests/test_reductions.py tests/test_eval_gpu.py tests/test_bench_launch_gpu.py > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -2 | tee -a $O/summary.txt
for rep in 1 2 3; do
  for sg in 0 1; do
    WB2HIP_SGPR_UNALIGNED=$sg timeout 100 python bench.py --variants-only 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('SGPR_UNALIGNED=$sg ' + ' '.join('%s=%.4f' % (k[:10], v['kernel_ms']) for k, v in d.items()))" | tee -a $O/summary.txt
  done
done
