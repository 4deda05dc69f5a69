cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3n
mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python $GRAFT_REPO_ROOT/tools/ens_gather_bench.py > $O/bench.log 2>&1)
tail -1 $O/bench.log
f=$(find $O/prof -name '*kernel_stats.csv' | head -1)
cut -d, -f1-4 $f | sed 's/(wb2.*)"/"/' | cut -c1-150 | head -8
rm -rf $O/prof
