cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c7
O=gpurun_out/c7
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_ens_gpu.py tests/test_bench_launch_gpu.py tests/test_fuzz_gpu.py tests/test_golden_fixtures.py tests/test_spatial_gpu.py -m gpu -q 2>&1 | tail -15 > $O/pytest_gpu.txt; tail -15 $O/pytest_gpu.txt
for rep in 1 2 3; do
for v in "" k3_sort2; do
  lib=""; [ -n "$v" ] && lib="$GRAFT_REPO_ROOT/build/variants/libwb2hip_$v.so"
  WB2HIP_LIB=$lib timeout 200 python bench.py --workload ensemble --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $O/ens_${v:-main}_$rep.json
  python -c "
import json; d=json.load(open('$O/ens_${v:-main}_$rep.json')); r=d['roofline']; print('ens ${v:-main} kernel_ms %.4f GB/s %.0f frac %.3f value %.4g' % (r['kernel_ms'], r['achieved'], r['frac'], d['value']))"
done
done
