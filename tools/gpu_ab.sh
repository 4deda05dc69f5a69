# One-off GPU checks of a kernel change (edit freely; not part of the product):
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
timeout 1500 python -m pytest -x -q -m gpu tests/test_ens_gpu.py tests/test_ens_exact_gpu.py tests/test_evalall.py tests/test_reference_vectors.py tests/test_bench_launch_gpu.py > gpurun_out/ab/pytest.txt 2>&1; tail -4 gpurun_out/ab/pytest.txt
timeout 900 python tools/k3_variants.py > gpurun_out/ab/k3.json 2>gpurun_out/ab/k3.err; tail -2 gpurun_out/ab/k3.err
python - <<'PY'
import json
a=json.load(open('gpurun_out/ab/k3.json'))
for k,v in a.items(): print(f"{k:24s} {v['kernel_ms']:.4f} ms  frac {v['frac']:.3f} [{v['frac_min']:.3f}, {v['frac_max']:.3f}]")
PY
