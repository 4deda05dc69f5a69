# Round 3: K1 with packed elementwise pairs, |d| as an FMA source modifier,
# SGPR-base addressing and wide loads on rows of any length (lon-lat layout):
# parity (full GPU suite), the default line, and the old one-column rule for
# the lon-lat layout beside it.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3k
mkdir -p $O
timeout 1500 python -m pytest -x -q -m gpu tests > $O/pytest_full.txt 2>&1; grep -E "passed|failed|error" $O/pytest_full.txt | tail -3 | tee $O/pytest.txt
timeout 400 python bench.py --no-cpu-baseline 2>$O/bench.err | tail -1 > $O/bench_line.json
python - <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/r3k/summary.txt
import json
d = json.loads(open('gpurun_out/r3k/bench_line.json').read())
r = d['roofline']
print('headline value=%.4g ms/step=%.4f K1=%.4f frac=%.3f traffic_ratio=%s' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('traffic_over_algorithmic')))
for k, v in d['variants'].items():
    print('variant %-22s %.4f ms frac %.3f' % (k, v['kernel_ms'], v['frac']))
print('ensemble', d['ensemble']['roofline']['kernel_ms'], d['ensemble']['roofline']['frac'], d['ensemble']['value'])
PY
for rule in 0 1 0 1; do
WB2HIP_UNALIGNED_VEC=$rule timeout 200 python bench.py --variants-only 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('UNALIGNED_VEC=$rule ' + ' '.join('%s=%.4f(%.3f)' % (k, v['kernel_ms'], v['frac']) for k, v in d.items()))" | tee -a $O/summary.txt
done
