cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/s13
for rep in 1 2; do
for v in default k1_noskip; do
  if [ $v = default ]; then unset WB2HIP_LIB; else export WB2HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libwb2hip_$v.so; fi
  timeout 300 python bench.py --variants-only 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$v', {k: round(v['frac'],3) for k,v in d.items()})"
done; done
unset WB2HIP_LIB
timeout 300 python -m pytest tests/test_det_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -2
