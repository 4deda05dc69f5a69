# The whole GPU suite on the current tree (what the driver runs at round end)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/suite
timeout 1200 python -m pytest -x -q -m gpu tests > gpurun_out/suite/pytest_full.txt 2>&1; grep -E "passed|failed|error" gpurun_out/suite/pytest_full.txt | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
