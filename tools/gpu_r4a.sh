cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest -x -q -m gpu tests/test_chunk_batching_gpu.py tests/test_det_gpu.py tests/test_eval_gpu.py tests/test_evalall.py tests/test_threads_gpu.py tests/test_xarray_loop_gpu.py > gpurun_out/r4a/pytest.txt 2>&1; tail -15 gpurun_out/r4a/pytest.txt
WB2HIP_FUSE_VARIABLES=0 timeout 600 python tools/official_chunk.py --chunks 32 --batch 1 > gpurun_out/r4a/official_unfused.json 2> gpurun_out/r4a/official_unfused.err; tail -c 1500 gpurun_out/r4a/official_unfused.json
timeout 600 python tools/official_chunk.py --chunks 96 --batch 1,16,32 > gpurun_out/r4a/official.json 2> gpurun_out/r4a/official.err; tail -c 3000 gpurun_out/r4a/official.json; tail -5 gpurun_out/r4a/official.err
timeout 600 python tools/official_chunk.py --chunks 64 --batch 16 --profile > gpurun_out/r4a/profile16.txt 2>&1; head -70 gpurun_out/r4a/profile16.txt
