cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/pairprof
for leg in pairs separate; do
 for nf in "" "--no-field"; do
  tag=${leg}${nf:+_nofield}
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_$tag -o run -- python $GRAFT_REPO_ROOT/tools/pair_bench.py --legs $leg --reps 6 $nf > /tmp/pp_$tag.log 2>&1)
  f=$(find /tmp/pp_$tag -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && head -12 "$f" > gpurun_out/pairprof/${tag}_kernel_stats.csv
  tail -1 /tmp/pp_$tag.log
 done
done
