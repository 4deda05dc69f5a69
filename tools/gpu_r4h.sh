cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
timeout 900 python tools/live_traffic.py --workload all > gpurun_out/r4h/live.json 2> gpurun_out/r4h/live.err; cat gpurun_out/r4h/live.json | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items(): print(k, round(v['ratio'],4), v['launches'], v['kernel'][:60])
"; tail -3 gpurun_out/r4h/live.err
timeout 900 python -m pytest -x -q -m gpu tests/test_live_traffic_gpu.py tests/test_bench_gpu.py > gpurun_out/r4h/pytest.txt 2>&1; tail -5 gpurun_out/r4h/pytest.txt
( time timeout 1500 python bench.py > gpurun_out/r4h/bench_default.json 2> gpurun_out/r4h/bench_default.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4h/bench_default.json').read().strip().splitlines()[-1])
print('value', d['value']/1e9, 'frac', d['roofline']['frac'], 'traffic ratio', d['roofline'].get('traffic_over_algorithmic'))
for k in ('ensemble',):
  r=d[k]['roofline']; print(k, r['frac'], r.get('frac_min'), r.get('frac_max'), r.get('traffic_over_algorithmic'))
for k,v in d['spectrum'].items():
  if isinstance(v,dict) and 'roofline' in v: r=v['roofline']; print('spectrum/'+k, r['frac'], r.get('frac_min'), r.get('frac_max'), r.get('traffic_over_algorithmic'))
r=d['spectrum']['roofline']; print('spectrum', r['frac'], r.get('frac_min'), r.get('frac_max'), r.get('traffic_over_algorithmic'))
for k,v in d['variants'].items(): print('k1', k, round(v['frac'],3), round(v['frac_min'],3), round(v['frac_max'],3))
a=d['api_official_chunk']; print('official', a.get('value',0)/1e9, {k:(round(v['value']/1e9,1)) for k,v in a.get('by_batch_chunks',{}).items()}, a.get('error'))
print('api', d['api'].get('value',0)/1e9)
print('full_suite', d['full_suite'].get('value',0)/1e9)
print('cpu', d['cpu_baseline'])
PY
tail -3 gpurun_out/r4h/bench_default.err
