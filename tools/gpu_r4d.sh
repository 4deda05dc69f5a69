cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
timeout 600 python tools/k3_variants.py > gpurun_out/r4d/k3_base.json 2>gpurun_out/r4d/err1.txt
WB2HIP_ENS_EXACT_TEST=1 timeout 600 python tools/k3_variants.py > gpurun_out/r4d/k3_exact.json 2>gpurun_out/r4d/err2.txt
python - <<'PY'
import json
a=json.load(open('gpurun_out/r4d/k3_base.json')); b=json.load(open('gpurun_out/r4d/k3_exact.json'))
for k in a: print(k, round(a[k]['kernel_ms'],4), round(a[k]['frac'],3), '->', round(b[k]['kernel_ms'],4), round(b[k]['frac'],3))
PY
