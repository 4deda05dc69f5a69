# One-off GPU checks of a kernel change (edit freely; not part of the product):
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
timeout 1200 python -m pytest -x -q -m gpu tests/test_det_gpu.py tests/test_fuzz_gpu.py tests/test_chunk_batching_gpu.py tests/test_eval_gpu.py tests/test_edge_gpu.py tests/test_golden_fixtures.py tests/test_reference_vectors.py > gpurun_out/ab/pytest.txt 2>&1; tail -4 gpurun_out/ab/pytest.txt
for rep in 1 2; do
for v in 1 0; do
  WB2HIP_WF_COLLAPSE=$v timeout 300 python bench.py --variants-only 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('collapse=$v', {k:(round(x['kernel_ms'],4), round(x['frac'],3)) for k,x in d.items() if k in ('headline','official16_landmask','skipna')})"
done
done
WB2HIP_WF_COLLAPSE=1 timeout 300 python tools/official_chunk.py --chunks 256 --batch 32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('official', d['value']/1e9, d['roofline']['det_acc']['frac'], d['roofline']['wind']['frac'], d['wall_ms_per_chunk'])"
