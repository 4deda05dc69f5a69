cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_suite_step_gpu.py tests/test_staging.py tests/test_chunk_batching_gpu.py tests/test_eval_gpu.py -x -q -m gpu > gpurun_out/s1/pytest.txt 2>&1; tail -5 gpurun_out/s1/pytest.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s1/bench_line.txt 2> gpurun_out/s1/bench.err ) 2>&1 | grep real
tail -c 3000 gpurun_out/s1/bench_line.txt; wc -c gpurun_out/s1/bench_line.txt
cp bench_detail.json gpurun_out/s1/ 2>/dev/null
timeout 300 python tools/official_chunk.py --chunks 64 --batch 1 --sections > gpurun_out/s1/sections.txt 2>&1; tail -25 gpurun_out/s1/sections.txt
