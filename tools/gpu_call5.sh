# Round 2, GPU call 5: full parity suite after the cache / thread / boundary
# rework, the bench line (api leg), API host profile.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c5
O=gpurun_out/c5
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/pytest_gpu.txt; tail -30 $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python - <<'PY'
import json
d = json.load(open('gpurun_out/c5/bench_driver.json'))
print('value %.4g  ms/step %.4f  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))
print('api', d.get('api'))
print('full_suite', {k: v for k, v in d['full_suite'].items() if k != 'workload'})
print('cpu', d['cpu_baseline']['legs'], d['cpu_baseline']['cores'])
PY
tail -3 $O/bench_driver.err
timeout 300 python tools/api_profile.py 2>&1 | grep -v amdgpu | head -45 > $O/api_profile.txt; head -45 $O/api_profile.txt
for w in spectrum spectrum_mean ensemble; do timeout 200 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$w.json; python -c "
import json; d=json.load(open('$O/bench_$w.json')); r=d['roofline']; print('$w kernel_ms %.4f GB/s %.0f frac %.3f value %.4g' % (r['kernel_ms'], r['achieved'], r['frac'], d['value']))"; done
