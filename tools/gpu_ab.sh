# One-off GPU checks of a kernel change (edit freely; not part of the product):
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
timeout 1200 python -m pytest -x -q -m gpu tests/test_chunk_batching_gpu.py > gpurun_out/ab/pytest.txt 2>&1; tail -30 gpurun_out/ab/pytest.txt
