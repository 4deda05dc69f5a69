# The evidence of a round, collected on the GPU box into gpurun_out/profiles_rNN/
# (copy what is to be judged into profiles/):
#   bash tools/round_profile.sh r04 [quick]
# * the default bench line (what the driver runs), the official-chunk leg and the
#   K3 variants as JSON;
# * rocprofv3 --kernel-trace --stats summaries of the default command and of
#   every --workload (the average kernel durations the bench line's rooflines
#   must agree with);
# * the PMC traffic of every benched kernel (tools/live_traffic.py, separate
#   --pmc passes, --kernel-trace only).
R=${1:-r04}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/profiles_$R
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python bench.py > $OUT/${R}_bench_default_line.json 2> $OUT/bench.err ) 2>&1 | grep real
timeout 600 python tools/live_traffic.py --workload all > $OUT/${R}_live_traffic.json 2>> $OUT/bench.err
stats() {  # name, bench args...
  name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o run -- python $GRAFT_REPO_ROOT/bench.py "$@" > $GRAFT_REPO_ROOT/$OUT/prof_$name.log 2>&1)
  f=$(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && head -25 "$f" > $OUT/${R}_${name}_kernel_stats.csv
  rm -rf /tmp/prof_$name
}
stats default --no-pmc --no-cpu-baseline
if [ "$2" != quick ]; then
  stats deterministic --no-pmc --no-cpu-baseline --no-secondary --no-api --no-pcie --no-full-suite
  for w in ensemble spectrum spectrum_materialized spectrum_mean; do
    stats $w --workload $w --no-cpu-baseline
    timeout 300 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${R}_bench_$w.json
  done
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_oc -o run -- python $GRAFT_REPO_ROOT/tools/official_chunk.py --chunks 128 --batch 32 > $GRAFT_REPO_ROOT/$OUT/${R}_official_chunk_batch32.json 2>/dev/null)
  f=$(find /tmp/prof_oc -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -25 "$f" > $OUT/${R}_official_chunk_kernel_stats.csv; rm -rf /tmp/prof_oc
  timeout 600 python tools/official_chunk.py > $OUT/${R}_official_chunk.json 2>/dev/null
  timeout 600 python tools/k3_variants.py > $OUT/${R}_k3_variants.json 2>/dev/null
  timeout 600 python tools/tier2_variants.py > $OUT/${R}_tier2_variants.json 2>/dev/null
fi
ls -la $OUT
