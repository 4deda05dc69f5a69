# Quick GPU check of the pieces touched last (bench paths, feeder, K2).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/chk
timeout 1200 python -m pytest tests/test_bench_gpu.py tests/test_eval_gpu.py tests/test_det_gpu.py tests/test_ens_gpu.py tests/test_bench_launch_gpu.py -m gpu -x -q 2>&1 | tail -12 | tee gpurun_out/chk/pytest.txt
timeout 200 python bench.py --workload ensemble --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ensemble kernel_ms %.4f frac %.3f value %.4g ms_per_step %.4f' % (r['kernel_ms'], r['frac'], d['value'], d['ms_per_step']))"
