cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/s12
( time timeout 1500 python -m pytest -x -q -m gpu tests > gpurun_out/s12/pytest_full.txt 2>&1 ) 2>&1 | grep real; tail -8 gpurun_out/s12/pytest_full.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
