cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/s10
timeout 600 python -m pytest tests/test_chunk_program_gpu.py tests/test_chunk_batching_gpu.py tests/test_eval_gpu.py -x -q -m gpu 2>&1 | tail -8
timeout 300 python tools/official_chunk.py --chunks 512 --batch 1 --profile > gpurun_out/s10/profile.txt 2>&1; grep -A45 "cumulative" gpurun_out/s10/profile.txt | head -70
