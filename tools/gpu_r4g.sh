cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
timeout 1200 python -m pytest -x -q -m gpu tests/test_ens_exact_gpu.py tests/test_ens_gpu.py > gpurun_out/r4g/pytest.txt 2>&1; tail -5 gpurun_out/r4g/pytest.txt
timeout 600 python - > gpurun_out/r4g/k3.json 2>gpurun_out/r4g/k3.err <<'PY'
import sys, json, torch
sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import k3_variants
print(json.dumps(k3_variants.variants(torch.device('cuda',0), reps=3, only=('skipna','skipna_nan_patches','headline_slice13','members51_skipna'))))
PY
python - <<PY
import json
a=json.load(open('gpurun_out/r4g/k3.json'))
print({k: (round(x['kernel_ms'],4), round(x['frac'],3)) for k,x in a.items()})
PY
