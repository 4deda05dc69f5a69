cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p
mkdir -p $O
timeout 600 python -m pytest -x -q -m gpu tests/test_det_gpu.py -k "quarter_degree" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
