# Round 3, A/B 11 (K4f, 16 units per launch): non-temporal loads / stores off,
# LATSEG with 1.5 / 2 / 3 rounds of tasks
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab3l
mkdir -p $O
: > $O/summary.txt
V=$GRAFT_REPO_ROOT/build/variants
run() {
  local n=$1 wl=$2; shift 2
  lib=""; [ "$n" != default ] && lib=$V/libwb2hip_$n.so
  WB2HIP_LIB=$lib timeout 100 python bench.py --workload $wl --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-8s %-22s %-8s step_ms=%.4f kernel_ms=%.4f frac=%.3f' % ('$n', '$wl', '$LATSEG', d['ms_per_step'], r['kernel_ms'], r['frac']))" | tee -a $O/summary.txt
}
for rep in 1 2; do
  for wl in spectrum spectrum_mean spectrum_materialized; do
    for n in default fnl fns; do run $n $wl; done
  done
  for r in 1.5 2 3 0.5; do
    export WB2HIP_LATSEG_ROUNDS=$r LATSEG=rounds$r; run default spectrum; unset WB2HIP_LATSEG_ROUNDS LATSEG
  done
done
