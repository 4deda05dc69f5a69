# FETCH_SIZE / WRITE_SIZE of the secondary kernels, separate passes, kernel-trace only
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for w in ensemble spectrum spectrum_mean; do
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${w}_$c -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload $w > /dev/null 2>&1)
    f=$(find gpurun_out/pmc_${w}_$c -name '*counter_collection.csv' | head -1)
    python - "$f" $w $c <<'PY'
import csv, sys, collections
f, w, c = sys.argv[1:4]
acc = collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    if row.get('Counter_Name') == c:
        acc[row['Kernel_Name'][:70]].append(float(row['Counter_Value']))
for k, v in acc.items():
    if 'wb2' in k:
        print(w, c, k, 'launches', len(v), 'mean', sum(v) / len(v))
PY
    rm -rf gpurun_out/pmc_${w}_$c
  done
done
