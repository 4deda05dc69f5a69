cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/chk
timeout 1200 python -m pytest tests/test_rank_histogram_gpu.py tests/test_fuzz_gpu.py tests/test_tier2_gpu.py -m gpu -q 2>&1 | tail -30 | tee gpurun_out/chk/pytest.txt
