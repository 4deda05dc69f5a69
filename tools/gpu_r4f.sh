cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
for v in default skw3 skw4 nofast; do
  if [ $v = default ]; then unset WB2HIP_LIB; else export WB2HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libwb2hip_$v.so; fi
  timeout 600 python - > gpurun_out/r4f/$v.json 2>gpurun_out/r4f/$v.err <<'PY'
import sys, json, torch
sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import k3_variants
print(json.dumps(k3_variants.variants(torch.device('cuda',0), reps=3, only=('skipna','skipna_nan_patches','headline_slice13'))))
PY
  python - <<PY
import json
a=json.load(open('gpurun_out/r4f/$v.json'))
print('$v', {k: (round(x['kernel_ms'],4), round(x['frac'],3)) for k,x in a.items()})
PY
done
