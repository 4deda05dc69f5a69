cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4j

for rep in 1 2; do
for v in default abreast1 r3spec; do
  if [ $v = default ]; then unset WB2HIP_LIB; else export WB2HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libwb2hip_$v.so; fi
  for w in spectrum spectrum_mean spectrum_materialized; do
    timeout 300 python bench.py --workload $w --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', '$w', round(r['kernel_ms'],4), round(r['frac'],3))"
  done
done
done
