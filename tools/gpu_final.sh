# Round-end check on a fresh box: the driver's three steps, then the profiles.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
O=gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python - <<'PY'
import json
d = json.load(open('gpurun_out/final/bench_driver.json'))
print('value %.4g  ms/step %.4f  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))
print('api', d['api']['ms_per_step'], 'full_suite', d['full_suite']['value'], d['full_suite']['ensemble_kernel']['frac'])
print('cpu', d['cpu_baseline']['legs'])
PY
# (profiles: tools/round2_profile.sh, run separately)
