# Round 3: lean skipna path of the exact-50 K3 kernel -- parity subset + variants
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3t
mkdir -p $O
timeout 600 python -m pytest -x -q -m gpu tests/test_ens_gpu.py tests/test_evalall.py tests/test_fuzz_gpu.py tests/test_reference_vectors.py tests/test_tier2_gpu.py tests/test_eval_gpu.py tests/test_edge_gpu.py tests/test_bench_launch_gpu.py tests/test_threads_gpu.py > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -2
grep -E "^(FAILED|ERROR)" $O/pytest.txt | head
timeout 200 python tools/k3_variants.py 2>/dev/null | tail -1 | tee $O/k3_variants.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items(): print('%-22s %.4f ms  frac %.3f' % (k, v['kernel_ms'], v['frac']))"
