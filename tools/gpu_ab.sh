# One-off GPU checks of a kernel change (edit freely; not part of the product):
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in default wg3; do
  if [ $v = default ]; then unset WB2HIP_LIB; else export WB2HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libwb2hip_$v.so; fi
  timeout 300 python bench.py --variants-only 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', {k:(round(x['kernel_ms'],4), round(x['frac'],3)) for k,x in d.items() if k in ('headline','official16_landmask','lonlat','skipna')})"
done
done
