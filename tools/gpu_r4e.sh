cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
timeout 1200 python -m pytest -x -q -m gpu tests/test_ens_exact_gpu.py tests/test_ens_gpu.py tests/test_bench_launch_gpu.py tests/test_evalall.py > gpurun_out/r4e/pytest.txt 2>&1; tail -12 gpurun_out/r4e/pytest.txt
timeout 900 python tools/k3_variants.py > gpurun_out/r4e/k3.json 2>gpurun_out/r4e/k3.err; tail -3 gpurun_out/r4e/k3.err
python - <<'PY'
import json
a=json.load(open('gpurun_out/r4e/k3.json'))
for k,v in a.items(): print(f"{k:24s} {v['kernel_ms']:.4f} ms  frac {v['frac']:.3f} [{v['frac_min']:.3f}, {v['frac_max']:.3f}]")
PY
