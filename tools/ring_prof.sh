# rocprofv3 averages of the headline K1 launch: batch form against ring forms
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/ring
for cfg in "0 0" "4 0" "3 3" "0 0" "4 0" "3 0"; do
  set -- $cfg
  rm -rf /tmp/rp
  (cd /tmp && WB2HIP_K1_RING=$1 WB2HIP_K1_RING_WAVES=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp -o run -- python $GRAFT_REPO_ROOT/bench.py --no-pmc --no-cpu-baseline --no-secondary --no-api --no-pcie --no-full-suite > /tmp/rp.log 2>/dev/null)
  f=$(find /tmp/rp -name '*kernel_stats.csv' | head -1)
  python - "$f" /tmp/rp.log "$1 $2" <<'PY'
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'stream_partials' in r['Name']]
line = [l for l in open(sys.argv[2]) if l.startswith('{')][-1]
d = json.loads(line)
print('ring/waves', sys.argv[3], [(r['Calls'], round(float(r['AverageNs']) / 1e3, 1)) for r in rows[:2]],
      'bench', round(d['value'] / 1e9, 1), round(d['ms_per_step'], 4), round(d['roofline']['frac'], 4))
PY
done 2>&1 | tee gpurun_out/ring/prof.txt
