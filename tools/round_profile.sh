# The evidence of a round, collected on the GPU box into gpurun_out/profiles_rNN/
# (copy what is to be judged into profiles/):
#   bash tools/round_profile.sh r05 [quick]
# * the contract line of the default command (what the driver runs) and its
#   bench_detail.json; the same with --detail (every variant set);
# * rocprofv3 --kernel-trace --stats summaries of the default command and of
#   every workload (the average kernel durations the rooflines must agree
#   with): the first 60 kernels of each, so that no benched kernel falls off;
# * the PMC traffic of every benched kernel (tools/live_traffic.py, separate
#   --pmc passes, --kernel-trace only).
R=${1:-r06}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/profiles_$R
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${R}_bench_default_line.json 2> $OUT/bench.err ) 2>&1 | grep real
cp bench_detail.json $OUT/${R}_bench_detail_default.json
( time timeout 1500 python bench.py --detail > $OUT/${R}_bench_detail_line.json 2>> $OUT/bench.err ) 2>&1 | grep real
cp bench_detail.json $OUT/${R}_bench_detail.json
timeout 600 python tools/live_traffic.py --workload all > $OUT/${R}_live_traffic.json 2>> $OUT/bench.err
stats() {  # name, command...
  name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o run -- "$@" > $GRAFT_REPO_ROOT/$OUT/prof_$name.log 2>&1)
  f=$(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && head -61 "$f" > $OUT/${R}_${name}_kernel_stats.csv
  rm -rf /tmp/prof_$name
}
B=$GRAFT_REPO_ROOT/bench.py
stats default python $B --no-pmc --no-cpu-baseline
if [ "$2" != quick ]; then
  stats deterministic python $B --no-pmc --no-cpu-baseline --no-secondary --no-api --no-pcie --no-full-suite
  for w in ensemble spectrum spectrum_materialized spectrum_mean spectrum_materialized_f64 spectrum_mean_f64; do
    stats $w python $B --workload $w --no-cpu-baseline
    timeout 300 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${R}_bench_$w.json
  done
  stats energy_score python $GRAFT_REPO_ROOT/tools/tier2_variants.py --only energy_score --reps 1
  stats k3_hosted python $GRAFT_REPO_ROOT/tools/k3_variants.py --reps 1 --only members44_hosted,members33_hosted,members77_hosted,members45,members51
  stats official_chunk python $GRAFT_REPO_ROOT/tools/official_chunk.py --batch default --chunks 480 --sections
  stats official_chunk_by_chunk python $GRAFT_REPO_ROOT/tools/official_chunk.py --batch 1 --chunks 256 --sections
  stats official_probabilistic python $GRAFT_REPO_ROOT/tools/official_probabilistic.py --chunks 1024 --windows default --only-windows
  stats official_spatial python $GRAFT_REPO_ROOT/tools/spatial_leg.py --window
  stats official_spatial_by_chunk python $GRAFT_REPO_ROOT/tools/spatial_leg.py --chunk-by-chunk
  timeout 900 python tools/official_chunk.py --batch 1,16,32,default --host-fed > $OUT/${R}_official_chunk.json 2>/dev/null
  timeout 300 python tools/pair_bench.py --window 8 --chunks 2 --reps 5 --json $OUT/${R}_pair_bench.jsonl > /dev/null 2>&1
  timeout 300 python tools/pair_bench.py --reps 5 --json $OUT/${R}_pair_bench.jsonl > /dev/null 2>&1
  timeout 300 python tools/pair_bench.py --no-field --window 8 --chunks 2 --reps 5 --json $OUT/${R}_pair_bench.jsonl > /dev/null 2>&1
  timeout 300 python tools/host_fed_leg.py 2>/dev/null | tail -1 > $OUT/${R}_host_fed_leg.json
  timeout 300 python tools/download_bench.py 2>/dev/null | tail -1 > $OUT/${R}_download.json
  for g in 240x121 64x32; do timeout 300 python tools/official_probabilistic.py --grid $g 2>/dev/null | tail -1; done > $OUT/${R}_official_probabilistic.json
  timeout 300 python tools/k3_grid.py --iters 200 2>/dev/null | tail -1 > $OUT/${R}_k3_grid.json
  timeout 300 python tools/map_accumulate_bench.py 2>/dev/null | tail -1 > $OUT/${R}_map_accumulate.json
  timeout 300 python tools/live_traffic.py --workload map_accumulate 2>/dev/null | tail -1 >> $OUT/${R}_map_accumulate.json
  timeout 600 python tools/k3_variants.py > $OUT/${R}_k3_variants.json 2>/dev/null
  timeout 600 python tools/tier2_variants.py > $OUT/${R}_tier2_variants.json 2>/dev/null
  ( for th in 4 8 16; do WB2HIP_COPY_THREADS=$th timeout 120 python tools/upload_sweep.py 2>/dev/null | tail -1; done
    for slots in 4 8; do WB2HIP_STAGE_SLOTS=$slots timeout 120 python tools/upload_sweep.py 2>/dev/null | tail -1; done
    for ds in 2 3; do WB2HIP_DMA_STREAMS=$ds timeout 120 python tools/upload_sweep.py 2>/dev/null | tail -1; done ) > $OUT/${R}_upload_sweep.txt
fi
python tools/kernel_table.py $R > $OUT/${R}_kernel_table.md 2>/dev/null || true
ls -la $OUT
