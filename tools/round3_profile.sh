# Round 3 profiles: rocprofv3 per-kernel stats of the default bench command and of
# each workload, HBM traffic counters of the secondary kernels (separate --pmc
# passes, --kernel-trace only), the bench lines themselves.  Output under
# gpurun_out/r03/ for copying into profiles/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
O=$GRAFT_REPO_ROOT/gpurun_out/r03
export TMPDIR=/tmp
# 1. the bench lines (default = what the driver runs)
( time timeout 600 python bench.py ) > $O/r03_bench_default_line.json 2> $O/bench_default.err
tail -4 $O/bench_default.err
for w in ensemble spectrum spectrum_materialized spectrum_mean; do timeout 200 python bench.py --workload $w 2>/dev/null | tail -1 > $O/r03_bench_$w.json; done
# 2. per-kernel stats: the default command (all legs) and each workload alone
prof() {  # name, bench args...
  local name=$1; shift
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o run -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-pmc "$@" > $O/prof_$name.log 2>&1)
  f=$(find $O/prof_$name -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && python - "$f" > $O/r03_${name}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
w = csv.writer(sys.stdout)
w.writerow(rows[0])
for r in rows[1:]:
    if 'wb2' in r[0]:          # our kernels only; names cut to a readable length
        w.writerow([r[0][:160]] + r[1:])
PY
  rm -rf $O/prof_$name
}
prof default
prof deterministic --no-secondary --no-pcie --no-api --no-full-suite
for w in ensemble spectrum spectrum_materialized spectrum_mean; do prof $w --workload $w; done
# 3. traffic of the secondary kernels
for w in ensemble spectrum spectrum_materialized spectrum_mean; do
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${w}_$c -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --ramp-ms 0 --workload $w > /dev/null 2>&1)
    f=$(find $O/pmc_${w}_$c -name '*counter_collection.csv' | head -1)
    python - "$f" $w $c <<'PY'
import csv, sys, collections
f, w, c = sys.argv[1:4]
acc = collections.defaultdict(list)
try:
  for row in csv.DictReader(open(f)):
    if row.get('Counter_Name') == c:
        acc[row['Kernel_Name'][:70]].append(float(row['Counter_Value']))
  for k, v in acc.items():
    if 'wb2' in k:
        print(w, c, '|', k, '| launches', len(v), 'mean', sum(v) / len(v))
except Exception as e:
  print(w, c, 'FAILED', e)
PY
    rm -rf $O/pmc_${w}_$c
  done
done 2>&1 | tee $O/r03_pmc_raw.txt
ls $O
timeout 1500 python -m pytest -x -q -m gpu tests > $O/pytest_full.txt 2>&1; grep -E "passed|failed" $O/pytest_full.txt | tail -3 | tee $O/pytest.txt
