# Round 3: K2 with grouped lanes per cell (global-only region), the leaner
# runtime-M K3 and the gathered ensembles: full GPU suite + the gather bench +
# global-only / 13-region K2 timings
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3o
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest -x -q -m gpu tests > $O/pytest_full.txt 2>&1; grep -E "passed|failed|error" $O/pytest_full.txt | tail -3 | tee $O/pytest.txt
grep -E "^(FAILED|ERROR)" $O/pytest_full.txt | head -10
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python $GRAFT_REPO_ROOT/tools/ens_gather_bench.py > $O/bench.log 2>&1)
grep -v "rocprofv3\|^[WE]2026" $O/bench.log | tail -1 | tee $O/gather_bench.json
f=$(find $O/prof -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY' | tee $O/gather_kernels.txt
import csv, sys
for r in list(csv.reader(open(sys.argv[1])))[1:4]:
    print(r[0][:90], r[1], 'avg_us %.1f' % (float(r[3]) / 1e3))
PY
rm -rf $O/prof
timeout 200 python bench.py --workload ensemble --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('ensemble step_ms=%.4f kernel_ms=%.4f frac=%.3f value=%.4g' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value']))" | tee $O/ens.txt
