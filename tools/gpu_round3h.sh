# Round 3: workgroup width of K3 (ens2/ens1) and K4f (fft2/fft1)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h
mkdir -p $O
: > $O/summary.txt
V=$GRAFT_REPO_ROOT/build/variants
run() {
  local name=$1 wl=$2; shift 2
  local lib=""; [ "$name" != default ] && lib=$V/libwb2hip_$name.so
  WB2HIP_LIB=$lib timeout 120 python bench.py --workload $wl --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('$name', '$wl', 'step_ms=%.4f kernel_ms=%.4f value=%.4g frac=%.3f' % (d['ms_per_step'], r['kernel_ms'], d['value'], r['frac']))
" | tee -a $O/summary.txt
}
for n in default ens2 ens1 default ens2; do run $n ensemble; done
for wl in spectrum spectrum_mean spectrum_materialized; do
  for n in default fft2 fft1 default; do run $n $wl; done
done
