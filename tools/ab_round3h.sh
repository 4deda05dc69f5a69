# Round 3, A/B 8 (K3): member pairs through v_pk_*_f32 (default) vs the scalar
# form (eold), each with non-temporal member loads (ent / eoldnt)
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab3h
mkdir -p $O
: > $O/summary.txt
V=$GRAFT_REPO_ROOT/build/variants
for rep in 1 2; do
  for n in eold default ent eoldnt; do
    lib=""; [ "$n" != default ] && lib=$V/libwb2hip_$n.so
    WB2HIP_LIB=$lib timeout 120 python bench.py --workload ensemble --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-8s ensemble step_ms=%.4f kernel_ms=%.4f frac=%.3f value=%.4g' % ('$n', d['ms_per_step'], r['kernel_ms'], r['frac'], d['value']))" | tee -a $O/summary.txt
  done
done
