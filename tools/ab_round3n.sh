# Round 3, A/B 13: rows per chunk of K3 at the full_suite launch size (16 x 13 slabs)
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab3n
mkdir -p $O
: > $O/summary.txt
for rep in 1 2; do
for r in 5 6 8 10 16; do
  WB2HIP_ENS_ROWS_PER_CHUNK=$r timeout 200 python bench.py --no-secondary --no-pcie --no-api --no-cpu-baseline --no-pmc --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); fs=d['full_suite']
print('rows=$r full_suite value=%.4g ms_per_step=%.3f K3_ms=%.3f frac=%.3f' % (fs['value'], fs['ms_per_step'], fs['ensemble_kernel']['kernel_ms'], fs['ensemble_kernel']['frac']))" | tee -a $O/summary.txt
done
done
