# SQ counters of the ensemble and spectrum kernels (one pass, 8 SQ slots)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for w in ensemble spectrum spectrum_mean; do
  extra=""; [ $w != deterministic ] && extra="--workload $w"
  (cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sq_$w -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --ramp-ms 0 $extra > /dev/null 2>&1)
  f=$(find gpurun_out/sq_$w -name '*counter_collection.csv' | head -1)
  python - "$f" $w <<'PY'
import csv, sys, collections
f, w = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f)):
    k = row['Kernel_Name'][:60]
    if 'wb2' in k:
        acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in acc.items():
    print(w, '|', k, '|', ' '.join(f'{c}={sum(v)/len(v):.4g}' for c, v in sorted(d.items())))
PY
  rm -rf gpurun_out/sq_$w
done
