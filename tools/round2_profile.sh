# Round 2 profiles: rocprofv3 per-kernel stats of the four bench workloads and the
# HBM traffic counters of the dominant kernels (separate --pmc passes,
# --kernel-trace only), written under gpurun_out/r02/ for copying into profiles/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
O=$GRAFT_REPO_ROOT/gpurun_out/r02
export TMPDIR=/tmp
for w in deterministic ensemble spectrum spectrum_materialized spectrum_mean; do
  extra="--no-full-suite --no-api"; [ $w != deterministic ] && extra="--workload $w"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -o run -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline $extra > $O/prof_$w.log 2>&1)
  f=$(find $O/prof_$w -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -12 "$f" > $O/r02_${w}_kernel_stats.csv
  tail -1 $O/prof_$w.log | cut -c1-200
  rm -rf $O/prof_$w
done
for w in deterministic ensemble spectrum spectrum_materialized spectrum_mean; do
  extra="--no-full-suite --no-api"; [ $w != deterministic ] && extra="--workload $w"
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${w}_$c -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --ramp-ms 0 --no-cpu-baseline $extra > /dev/null 2>&1)
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
done 2>&1 | tee $O/r02_pmc_raw.txt
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_det.json
for w in ensemble spectrum spectrum_materialized spectrum_mean; do timeout 200 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_$w.json; done
ls $O
# roctx ranges: marker + kernel trace of a short run (no counters in this pass)
(cd /tmp && timeout 300 rocprofv3 --marker-trace --kernel-trace --output-format csv -d $O/markers -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --ramp-ms 0 --no-cpu-baseline --no-full-suite --no-api > /dev/null 2>&1)
f=$(find $O/markers -name '*marker_api_trace.csv' | head -1); [ -n "$f" ] && (head -1 "$f"; grep wb2_ "$f" | head -12) > $O/r02_marker_trace_excerpt.csv
rm -rf $O/markers
cat $O/r02_marker_trace_excerpt.csv | cut -c1-200
