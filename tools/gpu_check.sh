cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/chk
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | tee gpurun_out/chk/pytest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
