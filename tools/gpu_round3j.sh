# Round 3: rows per chunk of the ensemble launch, re-swept with the cheaper fold
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3j
mkdir -p $O
for r in 8 12 16 24 8 16; do
  timeout 120 python bench.py --workload ensemble --steps 60 --warmup 10 --rows-per-chunk $r 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('rows $r step_ms=%.4f kernel_ms=%.4f frac=%.3f value=%.4g' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value']))" | tee -a $O/ens_rows.txt
done
