# Round-end check, no -x (every failure listed), bench without the CPU leg.
cd $GRAFT_REPO_ROOT
O=gpurun_out/final2
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 | tee $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d = json.load(open('gpurun_out/final2/bench.json'))
print('value %.4g  ms/step %.4f  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))
print('api', d['api']['ms_per_step'], 'full_suite', d['full_suite']['value'])
PY
