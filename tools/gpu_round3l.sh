# Round 3: K1 defaults (4 columns per lane for the heavy instantiations) and K3
# with packed member pairs: parity (full GPU suite), K3 A/B against the previous
# object (build/variants/libwb2hip_v4.so = same K1, old K3), the default line.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3l
mkdir -p $O
: > $O/summary.txt
timeout 1500 python -m pytest -x -q -m gpu tests > $O/pytest_full.txt 2>&1; grep -E "passed|failed|error" $O/pytest_full.txt | tail -3 | tee $O/pytest.txt
V=$GRAFT_REPO_ROOT/build/variants
for rep in 1 2; do
  for n in v4 default; do
    lib=""; [ "$n" != default ] && lib=$V/libwb2hip_$n.so
    WB2HIP_LIB=$lib timeout 120 python bench.py --workload ensemble --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$n ensemble step_ms=%.4f kernel_ms=%.4f frac=%.3f value=%.4g' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value']))" | tee -a $O/summary.txt
  done
done
timeout 400 python bench.py --no-cpu-baseline 2>$O/bench.err | tail -1 > $O/bench_line.json
python - <<'PY' | tee -a $GRAFT_REPO_ROOT/gpurun_out/r3l/summary.txt
import json
d = json.loads(open('gpurun_out/r3l/bench_line.json').read())
r = d['roofline']
print('headline value=%.4g ms/step=%.4f K1=%.4f frac=%.3f traffic_ratio=%s' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('traffic_over_algorithmic')))
for k, v in d['variants'].items():
    print('variant %-22s %.4f ms frac %.3f' % (k, v['kernel_ms'], v['frac']))
print('ensemble', d['ensemble']['roofline']['kernel_ms'], d['ensemble']['roofline']['frac'], d['ensemble']['value'])
print('full_suite', d.get('full_suite'))
PY
