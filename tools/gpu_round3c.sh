# Round 3: the default bench line (what the driver runs) + the bench / traffic tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c
mkdir -p $O
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_default.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3c/bench_default.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', {k: d['roofline'][k] for k in ('frac', 'kernel_ms', 'traffic', 'traffic_over_algorithmic') if k in d['roofline']})
print('traffic err', d['roofline'].get('traffic_live_error'))
for k in ('ensemble', 'spectrum'):
    print(k, d[k].get('value'), d[k].get('roofline', d[k]).get('frac') if 'roofline' in d[k] else d[k])
for k in ('materialized', 'time_mean'):
    print(' ', k, d['spectrum'][k].get('roofline', d['spectrum'][k]))
print('variants')
for k, v in d['variants'].items():
    print('  ', k, v if 'error' in k else (round(v['kernel_ms'], 4), round(v['frac'], 3)))
print('pcie', d.get('pcie_inclusive'))
print('full_suite', d['full_suite']['value'], d['full_suite']['ensemble_kernel']['frac'])
print('api', d.get('api'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
timeout 900 python -m pytest -x -q -m gpu tests/test_bench_gpu.py tests/test_live_traffic_gpu.py 2>&1 | tail -8 | tee $O/pytest.txt
