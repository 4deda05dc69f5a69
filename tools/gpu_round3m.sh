# Round 3: gathered ensembles (wb2_ens_partials_gather) and the leaner runtime-M
# kernels -- parity + timing (gather in place vs index_select + strided K3)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3m
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest -x -q -m gpu tests/test_ens_gpu.py tests/test_evalall.py tests/test_tier2_gpu.py tests/test_rank_histogram_gpu.py tests/test_reference_vectors.py tests/test_fuzz_gpu.py > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -3
grep -E "^(FAILED|ERROR)|Error" $O/pytest.txt | head -10
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python $GRAFT_REPO_ROOT/tools/ens_gather_bench.py > $O/bench.log 2>&1)
grep -v "rocprofv3\|^[WE]2026" $O/bench.log | tail -1 | tee $O/gather_bench.json
f=$(find $O/prof -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY' | tee $O/gather_kernels.txt
import csv, sys
for r in list(csv.reader(open(sys.argv[1])))[1:8]:
    print(r[0][:90], r[1], 'avg_us %.1f' % (float(r[3]) / 1e3))
PY
rm -rf $O/prof
