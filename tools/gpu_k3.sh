cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/k3
timeout 900 python -m pytest tests/test_ens_gpu.py tests/test_bench_launch_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/k3/pytest.txt
for rep in 1 2; do
for v in default base; do
  if [ $v = default ]; then unset WB2HIP_LIB; else export WB2HIP_LIB=$PWD/build/variants/libwb2hip_$v.so; fi
  echo "== $v" >> gpurun_out/k3/summary.txt
  timeout 300 python bench.py --workload ensemble --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/k3/summary.txt
done
done
python - <<'PY'
import json
for l in open('gpurun_out/k3/summary.txt'):
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['value']/1e9, d['roofline']['kernel_ms'], d['roofline']['frac'])
    else: print(l.strip())
PY
