cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c8
O=gpurun_out/c8
timeout 900 python -m pytest tests/test_spectrum_gpu.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_gpu.txt
for rep in 1 2; do
for w in spectrum spectrum_mean; do timeout 200 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_${w}_$rep.json; python -c "
import json; d=json.load(open('$O/bench_${w}_$rep.json')); r=d['roofline']; print('$w kernel_ms %.4f GB/s %.0f frac %.3f value %.4g ms_per_step %.4f' % (r['kernel_ms'], r['achieved'], r['frac'], d['value'], d['ms_per_step']))"; done
done
