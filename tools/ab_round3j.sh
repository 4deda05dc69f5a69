# Round 3, A/B 10 (K3): rows per chunk x waves per workgroup, non-temporal loads
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab3j
mkdir -p $O
: > $O/summary.txt
V=$GRAFT_REPO_ROOT/build/variants
run() {
  local n=$1; shift
  lib=""; [ "$n" != default ] && lib=$V/libwb2hip_$n.so
  WB2HIP_LIB=$lib timeout 100 python bench.py --workload ensemble --steps 100 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-8s %-22s step_ms=%.4f kernel_ms=%.4f frac=%.3f value=%.4g' % ('$n', '$*', d['ms_per_step'], r['kernel_ms'], r['frac'], d['value']))" | tee -a $O/summary.txt
}
for rep in 1 2; do
  for r in 5 6 7 8; do
    for n in default ewg2; do run $n --rows-per-chunk $r; done
  done
done
