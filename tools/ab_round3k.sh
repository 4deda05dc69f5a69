# Round 3: the spectrum workloads at 8 / 16 / 32 units per launch (the headline
# runs 16 units per launch)
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab3k
mkdir -p $O
: > $O/summary.txt
for rep in 1 2; do
for u in 8 16 32; do
  for wl in spectrum spectrum_mean spectrum_materialized; do
    timeout 100 python bench.py --workload $wl --steps 100 --warmup 10 --spectrum-units $u 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('units=%-3d %-22s step_ms=%.4f kernel_ms=%.4f frac=%.3f value=%.4g' % ($u, '$wl', d['ms_per_step'], r['kernel_ms'], r['frac'], d['value']))" | tee -a $O/summary.txt
  done
done
done
