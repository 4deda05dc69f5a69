# Round 3, last run: the driver-shaped default line + the whole GPU suite
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03f
mkdir -p $O
( time timeout 600 python bench.py ) > $O/r03_bench_default_line.json 2> $O/bench_default.err
tail -4 $O/bench_default.err
timeout 1500 python -m pytest -x -q -m gpu tests > $O/pytest_full.txt 2>&1; grep -E "passed|failed|error" $O/pytest_full.txt | tail -3 | tee $O/pytest.txt
